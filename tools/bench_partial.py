"""SCAPE-partial-shaped criterion step (config 3): N_src = 4995 full shape against N_tgt = 2200 partial view,
GraphDeformLoss_Neural_Partial forward + backward w.r.t. the features and the Deformer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from models.loss import GraphDeformLoss_Neural_Partial
from models.model import Deformer
B, N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 4995, 2200
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, N, 128, generator=g).to(dev).requires_grad_(True)
f2 = torch.randn(B, M, 128, generator=g).to(dev).requires_grad_(True)
v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, M, 3, generator=g).to(dev)
d1, d2 = torch.cdist(v1, v1), torch.cdist(v2, v2)
dfm = Deformer(k=10).to(dev)
crit = GraphDeformLoss_Neural_Partial(k_deform=10, w_dist=0.02, w_map=0.0, k_dist=500, N_dist=1000, partial=True, w_deform=0.5,
                                      w_img=0, w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01, save_name="bench")


def step():
    out = crit(f1, f2, d1, d2, v1, v2, 50.0, dfm)
    out[0].backward()
    return out


for _ in range(2): out = step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print("partial criterion fwd+bwd B=%d %dx%d: %.2f ms; losses %s; grad finite %s" %
      (B, N, M, dt * 1e3, [round(float(torch.as_tensor(o)), 4) for o in out], bool(torch.isfinite(f1.grad).all() and torch.isfinite(f2.grad).all())))
