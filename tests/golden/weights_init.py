"""Deterministic, name-keyed (re)initialisation of a module's parameters and BN buffers.

Used on the reference's modules when the golden vectors are generated and on this repo's
mirror modules in the tests, so both carry bit-identical weights without shipping a 7 MB
state_dict.  Depends only on torch's CPU generator (same torch build in both places).
"""
import zlib

import torch


def _gen(name, salt):
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31))


@torch.no_grad()
def reinit(module, salt=0):
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
            p.copy_(torch.randn(p.shape, generator=g) / fan_in ** 0.5)
    return module
