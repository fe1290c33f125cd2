#!/usr/bin/env python3
"""Deformation driver reproducing the reference's point-cloud branch (deform.py:219-262) on the MI355X path.

    graph(verts1) -> Pi_12 = topk_pi(knnsearch_t_grad(feat1, feat2, alpha=100)) -> verts12 = Pi_12 @ verts2
    -> Deformer(feat1[idx11], feat2[idx22], verts1, verts12, Pi_12, nodes) -> R (6D + identity), T -> dg(verts1, R, T)
    -> <out>/deform_<a>_<b>.off

Default: the fused C-ABI path (dvm_pair_direction_fwd_f32, nothing N x M in HBM).  --reference-sequence runs the
same steps one by one through the reference-named module API (models.loss / models.model / lib.*), dense Pi
included; both must give the same points (tests/test_gpu_backbone.py::test_deform_driver).
Features come from --pairs (.npz with verts1, verts2, feat1, feat2, name1, name2) or are synthetic.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from dvm import ops  # noqa: E402
import models.loss as ml  # noqa: E402
from models.model import Deformer  # noqa: E402


def deform_fused(deformer, feat1, feat2, verts1, verts2, alpha, start):
    out = ops.pair_direction(deformer.weight_list(feat1.device), feat1, feat2, verts1, verts2, alpha, start, with_map=False)
    return out["warped"]


def deform_reference_sequence(deformer, feat1, feat2, verts1, verts2, alpha, start, k_deform=10):
    """deform.py:232-257 call by call."""
    crit = ml.GraphDeformLoss_Neural(save_name="deform", dump=True)
    num_nodes_all1, dg_list1, _ = crit.deformation_graph_node(verts1, start)
    Pi_12 = crit.topk_pi(ml.knnsearch_t_grad(feat1, feat2, alpha=alpha))
    idx11, idx22 = ml.knn_grad(verts1, verts1, k_deform), ml.knn_grad(verts2, verts2, k_deform)
    feat2_conv, feat1_conv = ml.index_points(feat2, idx22), ml.index_points(feat1, idx11)
    verts12 = torch.matmul(Pi_12, verts2)
    deformations = deformer(feat1_conv, feat2_conv, verts1, verts12, Pi_12, num_nodes_all1.long())
    iden = torch.tensor([1, 0, 0, 0, 1, 0], dtype=torch.float32, device=verts1.device).view(1, 1, 6)
    R1 = ml.rotation_6d_to_matrix(deformations[:, :, 3:] + iden)
    T1 = deformations[:, :, :3]
    pts = [dg(verts1[i], R1[i].unsqueeze(0), T1[i].unsqueeze(0).contiguous())[0] for i, dg in enumerate(dg_list1)]
    return torch.cat(pts, dim=0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", nargs="*", default=None)
    ap.add_argument("--synthetic", type=int, default=1)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--alpha", type=float, default=100.0)
    ap.add_argument("--deformer-weights", default=os.path.join(HERE, "..", "tests", "golden", "deformer_scape_r_weights.npz"),
                    help="npz of the Deformer state_dict ('.' -> '__') or a .pth state_dict")
    ap.add_argument("--reference-sequence", action="store_true")
    ap.add_argument("--out", default="result/deform_amd")
    args = ap.parse_args(argv)
    assert torch.cuda.is_available(), "the deformation path needs a HIP device"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    deformer = Deformer(10)
    if args.deformer_weights.endswith(".npz"):
        deformer.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in np.load(args.deformer_weights).items()})
    else:
        deformer.load_state_dict(torch.load(args.deformer_weights, map_location="cpu"))
    deformer = deformer.to(dev).eval()
    os.makedirs(args.out, exist_ok=True)
    g = torch.Generator().manual_seed(args.seed)
    items = []
    if args.pairs:
        for path in args.pairs:
            d = np.load(path, allow_pickle=False)
            items.append((str(d["name1"]), str(d["name2"]), *[torch.from_numpy(d[k]).float() for k in ("verts1", "verts2", "feat1", "feat2")]))
    else:
        for p in range(args.synthetic):
            v1, v2 = torch.rand(args.points, 3, generator=g), torch.rand(args.points, 3, generator=g)
            f1, f2 = torch.randn(args.points, 128, generator=g), torch.randn(args.points, 128, generator=g)
            items.append(("s%03da" % p, "s%03db" % p, v1, v2, f1, f2))
    files = []
    with torch.no_grad():
        for name1, name2, v1, v2, f1, f2 in items:
            v1, v2, f1, f2 = (t.to(dev)[None] for t in (v1, v2, f1, f2))
            start = torch.randint(0, v1.shape[1], (1,), generator=g)   # deform.py draws it inside FPS
            fn = deform_reference_sequence if args.reference_sequence else deform_fused
            warped = fn(deformer, f1, f2, v1, v2, args.alpha, start.to(dev))
            path = os.path.join(args.out, "deform_%s_%s.off" % (name1, name2))
            ml.save_off_file(path, warped[0].cpu().numpy())
            files.append(path)
    print(json.dumps({"files": files}))


if __name__ == "__main__":
    main()
