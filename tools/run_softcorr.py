"""Driver for profiling the soft-correspondence kernel alone: B pairs, N=M=2048, d=128.
usage: run_softcorr.py [B] [reps] [variant] [alpha] [scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
g = torch.Generator().manual_seed(0)
f1 = (torch.randn(B, 2048, 128, generator=g) * scale).cuda(); f2 = (torch.randn(B, 2048, 128, generator=g) * scale).cuda()
for _ in range(2): ops.softcorr(f1, f2, alpha, variant=variant)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): out = ops.softcorr(f1, f2, alpha, variant=variant)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print("softcorr variant=%d alpha=%g B=%d: %.3f ms/call, %.2f us per pair-direction, %.1f TF (2NMd)" % (variant, alpha, B, dt * 1e3, dt / B * 1e6, B * 2 * 2048 * 2048 * 128 / dt / 1e12))
if variant == 3:
    ref = ops.softcorr(f1, f2, alpha, variant=2)
    print("  idx equal:", bool((ref[1] == out[1]).all()), " smax equal:", bool((ref[2] == out[2]).all()),
          " max rel val diff: %.2e" % float(((ref[0] - out[0]).abs() / ref[0].clamp_min(1e-30)).max()),
          " max rel sum diff: %.2e" % float(((ref[3] - out[3]).abs() / ref[3]).max()))
# (rows that failed pass B's certification: run with DVM_DEBUG=2, printed per call on stderr)
