"""One-process-per-GPU data parallelism for the pair workload.

Pairs are independent units: the forward benchmark shards them with no collective.  Training
has exactly one exchange step, the sum of the flat fp32 gradient (2.12 M floats, 8.5 MB) — one
bucket, one all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
"""
import ctypes

import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """Contiguous, balanced partition of `n_units` pairs: rank r owns [lo, hi)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradBucket:
    """All parameters' gradients live in ONE flat fp32 buffer, so a step needs a single all-reduce and no
    gather/scatter copies: `attach()` makes every `p.grad` a view into the buffer (autograd then accumulates
    in place), `zero()` clears all of them with one fill.  Parameters that receive no gradient (the reference
    has 12 such) simply stay zero, so every rank reduces the same layout.

    Gradients that were produced before `attach()` (or re-created by `zero_grad(set_to_none=True)`) are still
    honoured: `all_reduce_mean()` copies any `p.grad` that is not the bucket's own view into place first."""

    def __init__(self, params, attach=False):
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
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if attach:
            self.attach()

    def attach(self):
        """Point every parameter's .grad at its slice of the flat buffer (existing gradients are kept)."""
        for p, v in zip(self.params, self.views):
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def zero(self):
        """zero_grad for attached parameters: one fill instead of one per tensor."""
        self.flat.zero_()

    def _gather_foreign(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def all_reduce_mean(self, group=None):
        """grad <- mean over ranks of grad (what DDP computes for a loss that is a mean over the batch), one collective."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._gather_foreign()
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)

    def all_reduce_sum(self, group=None, async_op=False, always=False):
        """grad <- sum over ranks of grad, one collective.  With every rank back-propagating
        `criterion.data_parallel_loss(B_shard / B_global)` this is the single-process gradient of the global batch
        (the criterion mixes per-batch sums and means, so a plain mean would shrink its sum-type terms by 1/world).
        async_op=True returns the work handle (wait() before the optimizer step)."""
        self._gather_foreign()
        if dist.is_initialized() and (always or dist.get_world_size(group) > 1):   # always: also a one-rank group (the collective runs)
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None


def global_minmax(x, group=None):
    """Batch-global min/max across ranks: pos_encoding_sin_wave normalises with the min/max of the
    WHOLE batch tensor (models/model.py:548), so a sharded batch needs this to match a
    single-process run."""
    mn, mx = x.min().reshape(1), x.max().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return mn, mx


# ---------------------------------------------------------------------------------------------------------------------
# The caller's collective of the library's *_sync_f32 entry points (include/dvm.h: dvm_collective), on torch.distributed.
_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)


class _CollectiveStruct(ctypes.Structure):
    _fields_ = [("allreduce", _ALLREDUCE_FN), ("user", ctypes.c_void_p)]


class TorchCollective:
    """Cross-rank statistics for the native training node (dvm_uni3fc_train_{fwd,bwd}_sync_f32): the library calls `allreduce`
    on the host while it enqueues its launches; the statistics buffers lie inside the node's arena, so the call is an in-place
    torch.distributed.all_reduce of a VIEW of that arena tensor, issued under the HIP stream the library names (RCCL with the
    "nccl" backend; gloo copies through the host — the two-ranks-on-one-GPU tests).  `calls` counts the collectives."""

    def __init__(self, group=None):
        self.group, self.arena, self.calls, self.error = group, None, 0, None
        self._fn = _ALLREDUCE_FN(self._allreduce)            # (kept alive with the object: the library holds a raw pointer)
        self.struct = _CollectiveStruct(self._fn, None)

    def bind(self, arena):
        """The uint8 tensor the next native call works in (the buffers handed to `allreduce` are slices of it)."""
        self.arena = arena
        return ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

    def _allreduce(self, user, buf, count, dtype, op, stream):
        try:
            a = self.arena
            off, size = int(buf) - a.data_ptr(), 8 if dtype == 1 else 4
            if off < 0 or off + count * size > a.numel():
                raise RuntimeError("collective buffer outside the bound arena")
            t = a[off:off + count * size].view(torch.float64 if dtype == 1 else torch.float32)
            rop = (dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX)[op]
            st = torch.cuda.ExternalStream(int(stream), device=a.device) if stream else torch.cuda.default_stream(a.device)
            with torch.cuda.stream(st):
                dist.all_reduce(t, op=rop, group=self.group)
            self.calls += 1
            return 0
        except Exception as e:  # noqa: BLE001  (no exception may cross the C boundary: reported by the caller of the native call)
            self.error = e
            return -1
