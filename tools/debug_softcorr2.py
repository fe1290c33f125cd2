import sys, os
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'dv-matcher_amd'))
from dvm import ops
from oracle import oracle as O
g=dict(np.load(os.path.join(ROOT,'tests/golden/softcorr_randn_1024x1024_s3.npz')))
f1,f2=g['feat1'][0][:128],g['feat2'][0][:64]
d=O.cdist(f1,f2)   # [128,64]
F1=torch.from_numpy(f1).cuda()[None].repeat(64,1,1).contiguous()
F2=torch.from_numpy(f2).cuda()[:,None,:].contiguous()   # [64,1,128]
for variant in (1,2):
    v,i,s,l=ops.softcorr(F1,F2,1.0,topk=1,variant=variant)
    dd=(-s).cpu().numpy().T   # [128,64]
    print('variant',variant,'M=1 mismatches',(dd!=d).sum(),'of',d.size, 'max ulp', np.abs((dd-d)/np.spacing(d)).max())
    if (dd!=d).any():
        bad=np.argwhere(dd!=d); print(' bad query rows mod32 hist',np.bincount(bad[:,0]%32,minlength=32)); print(' q//32',np.bincount(bad[:,0]//32))
# row norms check
n=ops.rownorm2(torch.from_numpy(f1).cuda()).cpu().numpy(); print('norm mismatch',(n!=O.rownorm2(f1)).sum())
