"""Driver for profiling the soft-correspondence kernel alone: B pairs, N=M=2048, d=128."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, 2048, 128, generator=g).cuda(); f2 = torch.randn(B, 2048, 128, generator=g).cuda()
for _ in range(2): ops.softcorr(f1, f2, 100.0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): ops.softcorr(f1, f2, 100.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print("softcorr B=%d: %.3f ms/launch-group, %.2f us per pair-direction, %.1f TF (actual 2NMd)" % (B, dt * 1e3, dt / B * 1e6, B * 2 * 2048 * 2048 * 128 / dt / 1e12))
