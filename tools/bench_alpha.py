"""The fused pair forward at the ends of the training schedule's alpha range (10 ... 100): below alpha = 32 the sweep
evaluates every softmax term (flat rows), above it the lean variant skips the provably negligible ones."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvm import ops
dev = torch.device("cuda", 0)
wl = ops.deformer_weight_list(dict(np.load(os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz"))), dev)
B, N = 256, 2048
ALPHAS = [float(x) for x in os.environ.get("ALPHAS", "10,31,33,100").split(",")]
ITERS = int(os.environ.get("ITERS", "5"))
g = torch.Generator().manual_seed(0)
v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
s1 = torch.zeros(B, dtype=torch.int32, device=dev)
# the two synthetic feature sets of SURVEY §8d: N(0,1) features and the "trained-like" set 0.3 * relu(N(0,1))
for kind in ("randn", "trained-like"):
    f1, f2 = torch.randn(B, N, 128, generator=g).to(dev), torch.randn(B, N, 128, generator=g).to(dev)
    if kind == "trained-like":
        f1, f2 = 0.3 * torch.relu(f1), 0.3 * torch.relu(f2)
    for alpha in ALPHAS:
        outs = None
        for _ in range(2): outs = ops.pair_forward(wl, f1, f2, v1, v2, alpha, s1, s1, out=outs)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(ITERS): outs = ops.pair_forward(wl, f1, f2, v1, v2, alpha, s1, s1, out=outs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / ITERS
        print("%-12s alpha %5.1f: %.2f ms per %d pairs (%.0f pairs/s)" % (kind, alpha, dt * 1e3, B, B / dt))
