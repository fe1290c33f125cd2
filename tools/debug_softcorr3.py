import sys, os
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'dv-matcher_amd'))
from dvm import ops
from oracle import oracle as O
g=dict(np.load(os.path.join(ROOT,'tests/golden/softcorr_randn_1024x1024_s3.npz')))
f1,f2=g['feat1'][0][:128],g['feat2'][0][:64]
F1=torch.from_numpy(f1).cuda()[None].repeat(64,1,1).contiguous()
F2=torch.from_numpy(f2).cuda()[:,None,:].contiguous()
v,i,s,l=ops.softcorr(F1,F2,1.0,topk=1,variant=2)
dd=(-s).cpu().numpy().T   # [128,64] distances from MFMA kernel
acc=O.dot_chain(f1,f2); na=O.rownorm2(f1)[:,None]; nb=O.rownorm2(f2)[None,:]
f32=np.float32
def sq(x): return np.sqrt(np.maximum(x,f32(0))).astype(f32)
cands={'(acc+na)+nb':sq((acc+na)+nb),'(acc+nb)+na':sq((acc+nb)+na),'acc+(na+nb)':sq(acc+(na+nb)),
       'chain from na then +nb': None}
for k,vv in cands.items():
    if vv is not None: print(k,'mismatch',(vv!=dd).sum())
# is dd^2 close: compare squares
d2_est=(dd.astype(np.float64)**2)
ref=((acc+na)+nb).astype(np.float64)
print('rel err of d^2 vs expected d2: max',np.abs(d2_est-ref).max()/ref.mean())
bad=np.argwhere(sq((acc+na)+nb)!=dd)
print('sign of diff', np.sign((dd-sq((acc+na)+nb))[bad[:,0],bad[:,1]]).sum(), len(bad))
# sqrt candidates: maybe non-correctly-rounded sqrt
x=((acc+na)+nb)
exact=np.sqrt(x.astype(np.float64))
print('dd vs exact sqrt in ulps: max', np.abs((dd-exact)/np.spacing(dd)).max(), ' correctly rounded max 0.5')
