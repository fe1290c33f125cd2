"""Deterministic, name-keyed (re)initialisation of a module's parameters and BN buffers.

Used on the reference's modules when the golden vectors are generated and on this repo's
mirror modules in the tests, so both carry bit-identical weights without shipping a 7 MB
state_dict.  Depends only on torch's CPU generator (same torch build in both places).

`gain` scales every weight matrix.  gain = 1 (unit-variance layers under residual connections) puts LG-Net in a CHAOTIC
regime: one flipped feature-space neighbour changes a point by ~2e-3, which flips neighbours of its neighbours in the
next attention layer, and so on — the reference evaluated with 1 and with 8 CPU threads then disagrees on half of the
points of a 1024-point SCAPE shape (tests/golden/bb_noise_summary.npz).  gain = 0.5 damps the cascade (a flip stays
local), which is the regime a trained network must be in for its output to be reproducible at all; whole-network
free-running comparisons use it.
"""
import zlib

import torch


def _gen(name, salt):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31))


@torch.no_grad()
def reinit(module, salt=0, gain=1.0):
    seen = set()
    for name, p in sorted(module.state_dict().items()):
        if p.data_ptr() in seen:  # tied / doubly-registered tensors (bnX == convX.1, q_conv == k_conv)
            continue
        seen.add(p.data_ptr())
        g = _gen(name, salt)
        if name.endswith("num_batches_tracked"):
            continue
        if name.endswith("running_var"):
            p.copy_(0.5 + torch.rand(p.shape, generator=g))
        elif name.endswith("running_mean"):
            p.copy_(0.1 * torch.randn(p.shape, generator=g))
        elif p.dim() == 1 and ("bn" in name or "norm" in name or name.split(".")[-2].isdigit() and name.endswith("weight") and p.dim() == 1):
            if name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        elif p.dim() == 1:  # biases
            p.copy_(0.05 * torch.randn(p.shape, generator=g))
        else:
            fan_in = p[0].numel()
            p.copy_(gain * torch.randn(p.shape, generator=g) / fan_in ** 0.5)
    return module


def dino_from_seed(seed, B, N):
    """(B,N,1152) stand-in visual features, exactly representable in fp16: a pure function of the seed, so fixtures
    store the seed instead of 2.3 MB per 1024 points (generator and tests call this same function)."""
    return torch.randn(B, N, 1152, generator=torch.Generator().manual_seed(int(seed))).half().float()
