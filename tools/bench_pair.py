import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvm import ops
dev = torch.device("cuda", 0)
wl = ops.deformer_weight_list(dict(np.load(os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz"))), dev)
for (B, N, M) in ((1, 4995, 4995), (2, 4995, 4995), (1, 4995, 2200), (8, 2048, 2048), (64, 2048, 2048)):
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(B, N, 128, generator=g).to(dev), torch.randn(B, M, 128, generator=g).to(dev)
    v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, M, 3, generator=g).to(dev)
    s1 = torch.zeros(B, dtype=torch.int32, device=dev); s2 = torch.zeros(B, dtype=torch.int32, device=dev)
    outs = None
    for _ in range(3): outs = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s1, s2, with_map=(N == M), out=outs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): outs = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s1, s2, with_map=(N == M), out=outs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("pair_forward B=%d %dx%d: %.3f ms (%.1f pairs/s)" % (B, N, M, dt * 1e3, B / dt))
