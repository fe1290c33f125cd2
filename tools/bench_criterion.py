"""The criterion alone in training mode (forward + backward to the features and the Deformer), B pairs of N points: wall time per
call and the host's enqueue time — is this phase of the step bound by the host or by the GPU?  usage: bench_criterion.py [B N reps]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
import train_driver as td
from models.model import Deformer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0); random.seed(0)
cfg = td.FULL_CFG
crit = td.build_criterion(cfg, False, N)
dfm = Deformer(k=10).to(dev).train()
g = torch.Generator().manual_seed(1)
f1 = torch.randn(B, N, 128, generator=g).to(dev).requires_grad_(True)
f2 = torch.randn(B, N, 128, generator=g).to(dev).requires_grad_(True)
v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
d1, d2 = torch.cdist(v1, v1), torch.cdist(v2, v2)


def step():
    out = crit(f1, f2, d1, d2, v1, v2, 10.0, dfm)
    t1 = time.perf_counter()
    out[0].backward()
    f1.grad = f2.grad = None
    return t1


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter(); tf = 0.0
for _ in range(reps):
    ts = time.perf_counter(); t1 = step(); tf += t1 - ts
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("criterion fwd+bwd, B=%d N=%d: %.2f ms per call; host enqueue %.2f ms (forward %.2f, backward %.2f)" % (B, N, dt * 1e3, th / reps * 1e3, tf / reps * 1e3, (th - tf) / reps * 1e3))
