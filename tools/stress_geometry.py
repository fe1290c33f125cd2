"""Grid-based searches (Chamfer, xyz kNN, graph build) on degenerate clouds: coincident points, a line, a plane,
huge / tiny extents, far-apart clouds — against the CPU oracle (bit-exact indices, distances to 1e-6 relative)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from dvm import ops
from oracle import oracle as O

dev = torch.device("cuda", 0)
rs = np.random.RandomState(0)
N = 700


def clouds():
    yield "coincident", np.tile(rs.rand(1, 3), (N, 1)), rs.rand(N, 3)
    yield "two_points", np.repeat(rs.rand(2, 3), N // 2, 0), rs.rand(N, 3)
    yield "line", np.outer(np.linspace(0, 1, N), [1, 2, 3]), rs.rand(N, 3)
    yield "plane", np.concatenate([rs.rand(N, 2), np.zeros((N, 1))], 1), rs.rand(N, 3)
    yield "huge", rs.rand(N, 3) * 1e4, rs.rand(N, 3) * 1e4
    yield "tiny", rs.rand(N, 3) * 1e-5, rs.rand(N, 3) * 1e-5
    yield "far_apart", rs.rand(N, 3), rs.rand(N, 3) + 50.0
    yield "one_outlier", np.concatenate([rs.rand(N - 1, 3), [[1e3, 1e3, 1e3]]]), rs.rand(N, 3)


bad = 0
for name, a, b in clouds():
    a, b = a.astype(np.float32), b.astype(np.float32)
    ta, tb = torch.from_numpy(a).to(dev)[None], torch.from_numpy(b).to(dev)[None]
    d1, d2, i1, i2 = ops.chamfer(ta, tb)
    od1, od2, oi1, oi2 = O.chamfer(a, b)
    ok_c = np.array_equal(i1[0].cpu().numpy(), oi1) and np.array_equal(i2[0].cpu().numpy(), oi2) and \
        np.allclose(d1[0].cpu().numpy(), od1, rtol=1e-6, atol=0) and np.allclose(d2[0].cpu().numpy(), od2, rtol=1e-6, atol=0)
    idx = ops.knn_cdist(ta, ta, 10)[0].cpu().numpy()
    ok_k = np.array_equal(idx, O.knn_cdist(a, a, 10))
    g = ops.dg_build(ta, torch.zeros(1, dtype=torch.int32, device=dev))
    og = O.dg_build(a, 0)
    ok_g = all(np.array_equal(g[k][0].cpu().numpy(), og[k]) for k in ("nodes_idx", "one_ring", "infl_idx")) and \
        np.allclose(g["weights"][0].cpu().numpy(), og["weights"], rtol=1e-5, atol=1e-7, equal_nan=True)
    print("%-12s chamfer %s  knn %s  graph %s" % (name, ok_c, ok_k, ok_g))
    bad += (not ok_c) + (not ok_k) + (not ok_g)
print("mismatches:", bad)
