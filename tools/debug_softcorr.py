import sys, os
import numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'dv-matcher_amd'))
from dvm import ops
from oracle import oracle as O
g=dict(np.load(os.path.join(ROOT,'tests/golden/softcorr_randn_1024x1024_s3.npz')))
f1,f2=g['feat1'][0],g['feat2'][0]
a=float(g['alpha'])
v1,i1,s1,l1=[t.cpu().numpy()[0] for t in ops.softcorr(torch.from_numpy(f1).cuda()[None],torch.from_numpy(f2).cuda()[None],a,variant=1)]
v2,i2,s2,l2=[t.cpu().numpy()[0] for t in ops.softcorr(torch.from_numpy(f1).cuda()[None],torch.from_numpy(f2).cuda()[None],a,variant=2)]
ov,oi,os_,ol=O.softcorr(f1,f2,a)
print('v1 vs oracle smax mism',(s1!=os_).sum(),'v2 vs oracle',(s2!=os_).sum(),'idx v2==oracle',(i2==oi).all())
bad=np.where(s2!=os_)[0]
print('bad rows',bad[:40])
print('bad rows mod 32',np.bincount(bad%32,minlength=32))
print('bad rows (row//32)%4 (wave)',np.bincount((bad//32)%4,minlength=4))
j=oi[bad,0]
print('argmin key mod 64 hist',np.bincount(j%64,minlength=64))
print('key local: sub',np.bincount((j%64)//32),' h=((j%32)>>2)&1',np.bincount(((j%32)>>2)&1))
d=O.cdist(f1,f2)
dmin=d.min(1)
print('oracle smax == dmin*negalpha',(os_==dmin*np.float32(-a)).all())
# what d would produce s2?
na=np.float32(-a)
print('s2/na vs dmin ulps', ((s2[bad]/na - dmin[bad])/np.spacing(dmin[bad]))[:20])
