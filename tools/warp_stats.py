import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch, numpy as np
import bench
from dvm import ops
wl = ops.deformer_weight_list(bench.load_weights(), torch.device("cuda"))
f1, f2, v1, v2, s1, s2 = bench.make_batch(16, 1000, torch.device("cuda"))
o12, o21 = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s1, s2)
w = o12["warped"]
print("warped min/max", w.min().item(), w.max().item(), "frac outside [0,1]^3:", ((w < 0) | (w > 1)).any(-1).float().mean().item())
print("verts12 min/max", o12["verts12"].min().item(), o12["verts12"].max().item())
d = (w - v1).norm(dim=-1); print("displacement mean/max", d.mean().item(), d.max().item())
print("losses", o12["losses"][0].tolist())
import time
def t(a, b, name):
    for _ in range(2): ops.chamfer(a, b, want_idx=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.chamfer(a, b, want_idx=False)
    torch.cuda.synchronize(); print("chamfer %-22s %.3f ms (16 pairs, both ways)" % (name, (time.perf_counter() - t0) / 5 * 1e3))
t(o12["warped"], v2, "warped12 <-> v2")
t(o12["verts12"], v2, "verts12 <-> v2")
t(v1, v2, "v1 <-> v2 (uniform)")
d1, d2, _, _ = ops.chamfer(o12["warped"], v2)
print("mean d(warped->v2) %.4g  mean d(v2->warped) %.4g" % (d1.mean().item(), d2.mean().item()))
u = torch.unique((o12["verts12"][0] * 1e6).round(), dim=0).shape[0]
print("distinct verts12 points in pair 0:", u, "of", o12["verts12"].shape[1])
