"""Differentiable / module-level operators built on dvm.ops (the C ABI).

Each function here is what one of the reference-named modules calls; it owns the
`torch.autograd.Function` (where a backward exists) and the tensor-layout glue.
"""
import torch

from . import ops
from ._lib import DvmError


def deformer_sparse(module, feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1):
    """Deformer forward on raw features + kNN indices + sparse Pi -> (B,Nn,9)."""
    wl = module.weight_list(feat1.device)
    return ops.deformer(wl, feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1)


def mlp(module, z):
    """Deformer's decoder MLP 262->512->256->128->9 (ELU) on z (B,Nn,262)."""
    wl = module.weight_list(z.device)
    return ops.deformer_mlp(wl, z)


def pos_encoding(coor):
    return ops.pos_encoding(coor)


def sa_attention(x, w_qk, w_v, b_v):
    return ops.sa_attention(x, w_qk, w_v, b_v)


def n2p_attention(x, K, wq, wk, wv, heads):
    return ops.n2p_attention(x, K, wq, wk, wv, heads)
