"""Feature-space kNN with the dist-loss's k = 500 at the shipped column counts (block top-k kernel, up to 8192 columns) against
the oracle, and the dist-loss term at (N, N_dist, k_dist) = (4995 | 2200, 1000, 500) against an fp64 torch evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dvm import ops
from oracle import oracle as O
g = torch.Generator().manual_seed(1)
for (N, M, C, k) in ((50, 4995, 128, 500), (40, 2200, 128, 500), (30, 4097, 64, 300), (20, 8192, 128, 512), (33, 2049, 128, 65)):
    a, b = torch.randn(1, N, C, generator=g), torch.randn(1, M, C, generator=g)
    idx = ops.knn_neg(a.cuda(), b.cuda(), k)[0].cpu().numpy()
    ref = O.knn_neg(a[0].numpy(), b[0].numpy(), k)
    print("knn", N, M, C, k, "equal:", np.array_equal(idx, ref))
# dist loss at the shipped sizes against a torch fp64 evaluation
from oracle import torch_ref as TR
for (N, nA, k) in ((4995, 1000, 500), (2200, 1000, 500)):
    feat = torch.randn(1, N, 128, generator=g)
    v = torch.rand(1, N, 3, generator=g)
    dist = torch.cdist(v, v)
    anchors = torch.randperm(N, generator=g)[:nA].int()
    out = ops.dist_loss(feat.cuda(), dist.cuda(), anchors.cuda(), k)
    ref = TR.dist_loss_term(feat, dist, anchors.long(), k)
    print("dist_loss", N, nA, k, "rel_err %.1e" % (abs(float(out[0]) - float(ref)) / abs(float(ref))))
