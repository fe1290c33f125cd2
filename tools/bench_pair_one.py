"""One shape of tools/bench_pair.py (for rocprofv3): B N M reps."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvm import ops
B, N, M, reps = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda", 0)
wl = ops.deformer_weight_list(dict(np.load(os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz"))), dev)
g = torch.Generator().manual_seed(0)
f1, f2 = torch.randn(B, N, 128, generator=g).to(dev), torch.randn(B, M, 128, generator=g).to(dev)
v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, M, 3, generator=g).to(dev)
s1 = torch.zeros(B, dtype=torch.int32, device=dev); s2 = torch.zeros(B, dtype=torch.int32, device=dev)
outs = None
for _ in range(reps): outs = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s1, s2, with_map=(N == M), out=outs)
torch.cuda.synchronize()
