"""Every C-ABI operator alone at the two shapes the drivers use (one 4995-point pair; 8 pairs of 2048 points):
a quick way to spot an operator whose launch geometry does not fit a shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import numpy as np
import torch
from dvm import ops

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3


for B, N in ((1, 4995), (8, 2048)):
    g = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(B, N, 128, generator=g).to(dev), torch.randn(B, N, 128, generator=g).to(dev)
    v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, N, 3, generator=g).to(dev)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    wl = ops.deformer_weight_list(dict(np.load(os.path.join(ROOT, "tests", "golden", "deformer_scape_r_weights.npz"))), dev)
    rows = []
    rows.append(("softcorr (alpha 100, top-10)", timed(lambda: ops.softcorr(f1, f2, 100.0))))
    pval, pidx, _, _ = ops.softcorr(f1, f2, 100.0)
    rows.append(("argmin_exact (screened)", timed(lambda: ops.argmin_exact(f1, f2))))
    rows.append(("apply Pi@verts", timed(lambda: ops.apply(pval, pidx, v2))))
    rows.append(("knn_cdist xyz k=10", timed(lambda: ops.knn_cdist(v1, v1, 10))))
    idx11, idx22 = ops.knn_cdist(v1, v1, 10), ops.knn_cdist(v2, v2, 10)
    rows.append(("fps N/2", timed(lambda: ops.fps(v1, N // 2, start))))
    rows.append(("dg_build", timed(lambda: ops.dg_build(v1, start))))
    g1 = ops.dg_build(v1, start)
    v12 = ops.apply(pval, pidx, v2)
    rows.append(("deformer", timed(lambda: ops.deformer(wl, f1, f2, v1, v12, idx11, idx22, pval, pidx, g1["nodes_idx"]))))
    d9 = ops.deformer(wl, f1, f2, v1, v12, idx11, idx22, pval, pidx, g1["nodes_idx"])
    R = ops.rot6d(d9[..., 3:].contiguous() + torch.tensor([1, 0, 0, 0, 1, 0.], device=dev))
    rows.append(("dg_warp_arap", timed(lambda: ops.dg_warp_arap(v1, g1, R, d9[..., :3].contiguous()))))
    warped = ops.dg_warp_arap(v1, g1, R, d9[..., :3].contiguous())[0]
    rows.append(("chamfer (warped, verts2)", timed(lambda: ops.chamfer(warped, v2, want_idx=False))))
    rows.append(("chamfer (verts1, verts2)", timed(lambda: ops.chamfer(v1, v2, want_idx=False))))
    rows.append(("map_term", timed(lambda: ops.map_term(v12, v2, idx11, idx22, pval, pidx))))
    rows.append(("knn_neg C=128 k=40", timed(lambda: ops.knn_neg(f1, f1, 40))))
    x64 = torch.randn(B, N, 64, generator=g).to(dev)
    rows.append(("knn_neg C=64 k=40", timed(lambda: ops.knn_neg(x64, x64, 40))))
    rows.append(("pos_encoding", timed(lambda: ops.pos_encoding(v1.transpose(1, 2).contiguous()))))
    p16, v64 = torch.randn(B, N, 16, generator=g).to(dev) * 0.3, x64
    rows.append(("sa_attention_pm", timed(lambda: ops.sa_attention_pm(p16, v64))))
    qkv = torch.randn(B, N, 384, generator=g).to(dev)
    idx40 = ops.knn_neg(f1, f1, 40)
    rows.append(("n2p_core_fwd C=128", timed(lambda: ops.n2p_core_fwd(qkv, idx40))))
    k_d = min(500, N // 2)
    anchors = torch.randperm(N, generator=g)[:min(1000, N // 2)].to(dev).int()
    dist = torch.cdist(v1, v1)
    rows.append(("dist_loss (N_dist 1000, k 500)", timed(lambda: ops.dist_loss(f1, dist, anchors, k_d))))
    print("---- B = %d, N = M = %d" % (B, N))
    for name, ms in rows:
        print("%-34s %8.3f ms" % (name, ms))
