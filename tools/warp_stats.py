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
