"""One-process-per-GPU data parallelism for the pair workload.

Pairs are independent units: the forward benchmark shards them with no collective.  Training
has exactly one exchange step, the sum of the flat fp32 gradient (2.12 M floats, 8.5 MB) — one
bucket, one all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """Contiguous, balanced partition of `n_units` pairs: rank r owns [lo, hi)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradBucket:
    """All parameters' gradients viewed through ONE flat fp32 buffer so that a step needs a single
    all-reduce.  Parameters that received no gradient (the reference has 12 such)
    contribute zeros, so every rank reduces the same layout."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:  # tied parameters appear once
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)

    def pack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        return self.flat

    def unpack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def all_reduce_mean(self, group=None):
        """grad <- mean over ranks of grad (what DDP computes), with one collective."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pack()
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)
        self.unpack()


def global_minmax(x, group=None):
    """Batch-global min/max across ranks: pos_encoding_sin_wave normalises with the min/max of the
    WHOLE batch tensor (models/model.py:548), so a sharded batch needs this to match a
    single-process run."""
    mn, mx = x.min().reshape(1), x.max().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return mn, mx
