"""Differentiable / module-level operators built on dvm.ops (the C ABI).

Each function here is what one of the reference-named modules calls; it owns the
`torch.autograd.Function` (where a backward exists) and the tensor-layout glue.
"""
import threading

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


class _Conv1x1(torch.autograd.Function):
    """y[b] = W x[b] (+ bias) over (B,Cin,N): forward on dvm_linear_f32 (the reference's fp32 chain), backward as the
    two library GEMMs dX = W^T dY, dW = sum_b dY x^T."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.linear(x, w, bias=bias, channel_major=True)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        B = gy.shape[0]
        w2 = w.reshape(w.shape[0], -1)
        gx = torch.bmm(w2.t().unsqueeze(0).expand(B, -1, -1), gy) if ctx.needs_input_grad[0] else None
        gw = torch.bmm(gy, x.transpose(1, 2)).sum(0).view_as(w) if ctx.needs_input_grad[1] else None
        gb = gy.sum(dim=(0, 2)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def conv1x1(x, w, bias=None):
    """nn.Conv1d(kernel_size=1) over x (B,Cin,N) with weight (Cout,Cin[,1]) -> (B,Cout,N), autograd included."""
    ops._need_gpu(x, w)
    if not _needs_grad(x, w, bias):
        return ops.linear(x, w, bias=bias, channel_major=True)
    return _Conv1x1.apply(x.contiguous(), w, bias)


# ------------------------------------------------------------------------------------------
# Point-major training path: activations stay (B,N,C) between the operators — the layout of the kNN / attention cores and of
# dvm_linear_f32's inference mode — so no transposes (6 % of the round-1 step) and no ATen glue sit between them.
# Gradient accumulation fusion: when a leaf parameter already holds a .grad buffer (a zeroed flat bucket, or the
# gradient of an earlier call in the same step — the criterion calls the network once per shape), the backward kernels add
# into that buffer and hand autograd `None`: same sums as autograd's own `p.grad += g`, without a zero-fill and an add
# launch per parameter and call.  It must stay off for torch.autograd.grad() (which promises not to touch .grad), hence
# opt-in: the training driver switches it on.
_FUSE_GRAD_ACCUMULATION = False


def fuse_grad_accumulation(on=True):
    global _FUSE_GRAD_ACCUMULATION
    prev, _FUSE_GRAD_ACCUMULATION = _FUSE_GRAD_ACCUMULATION, bool(on)
    return prev


def _grad_buffer(p):
    g = p.grad if (_FUSE_GRAD_ACCUMULATION and p.is_leaf) else None
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == p.device) else None


class _LinearPM(torch.autograd.Function):
    """y = act(x W^T + bias) on point-major x (..., K): forward = the reference's fp32 chain (dvm_linear_f32), backward on the
    library's own kernels too: dX = (dY act') W through dvm_linear_f32 with the operands' roles swapped, dW through
    dvm_linear_wgrad_f32 (reduction over the rows, operands read as they lie)."""

    @staticmethod
    def forward(ctx, x, w, bias, slope):
        y = ops.linear(x, w, bias=bias, slope=slope)
        ctx.save_for_backward(x, w, y if slope != 1.0 else None)
        ctx.slope, ctx.has_bias = slope, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.slope != 1.0:
            gy = torch.ops.aten.leaky_relu_backward(gy, y, ctx.slope, True)     # y is the activation's OUTPUT
        Co = w.shape[0]
        w2 = w.reshape(Co, -1)
        K = w2.shape[1]
        g2, x2 = gy.reshape(-1, Co), x.reshape(-1, K)
        gx = ops.linear(w2.reshape(1, Co, K), g2, channel_major=True).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            buf = _grad_buffer(w)
            if buf is not None:
                ops.linear_wgrad(g2, x2, out=buf)
            else:
                gw = ops.linear_wgrad(g2, x2).view_as(w)
        gb = g2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


def linear_pm(x, w, bias=None, slope=1.0):
    """Point-major 1x1 conv / linear layer with autograd: x (..., K), w (Co,K[,1]) -> (..., Co)."""
    if not _needs_grad(x, w, bias):
        return ops.linear(x, w, bias=bias, slope=slope)
    return _LinearPM.apply(x.contiguous(), w, bias, slope)


class _BNActPM(torch.autograd.Function):
    """y = act(BatchNorm_train(x + res)) over the rows of point-major x (..., C), forward and backward on the fused kernels."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, momentum, eps, slope):
        y, mean, invstd = ops.bn_act_train_fwd_pm(x, res, gamma, beta, eps, slope, momentum, running_mean, running_var)
        ctx.save_for_backward(x, res, y, gamma, beta, mean, invstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, y, gamma, beta, mean, invstd = ctx.saved_tensors
        bufs = (_grad_buffer(gamma), _grad_buffer(beta))
        fused = bufs[0] is not None and bufs[1] is not None
        dx, dgamma, dbeta = ops.bn_act_train_bwd_pm(dy.contiguous(), y, x, res, gamma, mean, invstd, ctx.slope, grads=bufs if fused else None)
        if fused:
            dgamma = dbeta = None
        return dx, (dx if res is not None else None), dgamma, dbeta, None, None, None, None, None


class _CounterSink(threading.local):
    """Per thread: while `pending` is a list, fused BatchNorm calls append their `num_batches_tracked` to it instead of
    bumping it, and the owner of the list bumps them all in ONE launch (Uni3FC's point-major training forward: 52 -> 1)."""
    pending = None


_counter_sink = _CounterSink()


class batched_counter_updates:
    """Context manager around a forward pass made of bn_act_pm calls."""

    def __enter__(self):
        self._outer, _counter_sink.pending = _counter_sink.pending, []
        return self

    def __exit__(self, *exc):
        mine, _counter_sink.pending = _counter_sink.pending, self._outer
        if mine:
            with torch.no_grad():
                torch._foreach_add_(mine, 1)
        return False


def bn_act_pm(bn, x, res=None, slope=1.0):
    """act(bn(x + res)) for an nn.BatchNorm1d `bn` applied to point-major x (B,N,C) (the module's channel axis is the LAST
    one here).  Training mode with plain batch statistics: fused kernels; otherwise (eval mode, SyncBatchNorm, ...) the
    module itself on a transposed view."""
    fused = (type(bn) is torch.nn.BatchNorm1d and bn.training and bn.track_running_stats and bn.affine
             and bn.momentum is not None and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024)
    if not fused:
        z = x if res is None else x + res
        y = bn(z.transpose(1, 2)).transpose(1, 2)
        return y if slope == 1.0 else torch.nn.functional.leaky_relu(y, slope) if slope != 0.0 else torch.relu(y)
    if _counter_sink.pending is not None:
        _counter_sink.pending.append(bn.num_batches_tracked)
    else:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    return _BNActPM.apply(x.contiguous(), None if res is None else res.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.momentum, bn.eps, slope)


class _Uni3FCTrain(torch.autograd.Function):
    """LG-Net's whole training-mode forward / backward as ONE autograd node over two native calls (dvm_uni3fc_train_{fwd,bwd}_f32:
    the launches of Uni3FC._forward_train_pm and of the graph autograd records for it, enqueued without Python in between).
    Inputs after (x, dino): the trainable tensors of the parameter table, in table order, so that autograd routes their
    gradients; `meta` = (table tensors, positions of the trainable ones, k, eps, momentum)."""

    @staticmethod
    def forward(ctx, meta, x, dino, *trainable):
        table, where, k, eps, momentum = meta[:5]
        defer = len(meta) > 5 and meta[5] is not None     # meta[5]: a list that receives the arena (deferred running statistics)
        groups = meta[6] if len(meta) > 6 else 1          # meta[6]: network calls merged into this one
        coll = meta[7] if len(meta) > 7 else None         # meta[7]: dvm.dist.TorchCollective — batch statistics over all ranks
        feat, tmp, arena = ops.uni3fc_train_forward(table, x, dino, k, eps, momentum, defer_stats=defer, groups=groups, coll=coll)
        if defer:
            meta[5].append(arena)
        ctx.set_materialize_grads(False)
        ctx.meta, ctx.arena, ctx.dino = meta, arena, dino
        ctx.trainable = trainable
        ctx.save_for_backward(feat, tmp)
        return feat, tmp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_feat, g_tmp):
        if ctx.arena is None:
            # the activations live in a caller-side arena that the first backward releases (1.1 GB at 8 x 2048), and in the fused
            # mode the gradients were ADDED into the parameters' buffers: a second pass through this node cannot be replayed
            raise RuntimeError("dvm Uni3FC training node: backward ran already (its activation arena is released after the first pass; "
                               "retain_graph / a second backward through the network is not supported — set Uni3FC.native_train = False for that)")
        table, where, k = ctx.meta[:3]
        feat, tmp = ctx.saved_tensors
        trainable = ctx.trainable
        if g_feat is None:
            g_feat = torch.zeros_like(feat)
        bufs = [_grad_buffer(p) if ctx.needs_input_grad[3 + i] else None for i, p in enumerate(trainable)]
        fused = all(b is not None for b, need in zip(bufs, ctx.needs_input_grad[3:]) if need) and all(ctx.needs_input_grad[3:])
        grads = [None] * len(table)
        if fused:       # the kernels add into the parameters' existing .grad buffers; autograd gets None
            for pos, b in zip(where, bufs):
                grads[pos] = b
            out = [None] * len(trainable)
        else:           # one zeroed flat buffer; autograd accumulates its views into .grad
            sizes = [p.numel() for p in trainable]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=feat.device)
            out, off = [], 0
            for pos, p, n in zip(where, trainable, sizes):
                grads[pos] = flat[off:off + n]
                out.append(flat[off:off + n].view_as(p))
                off += n
            out = [g if need else None for g, need in zip(out, ctx.needs_input_grad[3:])]
        ops.uni3fc_train_backward(table, grads, ctx.dino, feat, tmp, ctx.arena, g_feat.contiguous(), None if g_tmp is None else g_tmp.contiguous(), k,
                                  groups=ctx.meta[6] if len(ctx.meta) > 6 else 1, coll=ctx.meta[7] if len(ctx.meta) > 7 else None)
        ctx.arena = None
        return (None, None, None) + tuple(out)


class _CriterionTrain(torch.autograd.Function):
    """The deformation part of the training criterion — both directions of deform() for the B pairs of a step, as 2B directional
    pairs — as ONE autograd node over two native calls (dvm_criterion_train_{fwd,bwd}_f32): the ~300 launches the autograd path
    enqueued through ~40 nodes, without Python in between.  meta = (verts (2B,N,3), graph dict, knn_idx (2B,N,k), alpha, topk,
    with_map, dist) with dist = None or (dist1, dist2, anchors1, anchors2, k_dist) for the dist term; inputs: feat (2B,N,128) and the
    Deformer's 10 tensors in ops.DEFORMER_KEYS order.  -> terms (2B,7)."""

    @staticmethod
    def forward(ctx, meta, feat, *trainable):
        verts, g, knn_idx, alpha, topk, with_map, dist = meta
        terms, arena = ops.criterion_train_forward([p.detach() for p in trainable], feat, verts, g, knn_idx, alpha, topk, with_map, dist)
        ctx.meta, ctx.arena, ctx.trainable = meta, arena, trainable
        ctx.save_for_backward(feat)
        return terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_terms):
        if ctx.arena is None:
            raise RuntimeError("dvm criterion training node: backward ran already (its arena is released after the first pass)")
        verts, g, knn_idx, alpha, topk, with_map, dist = ctx.meta
        (feat,) = ctx.saved_tensors
        trainable = ctx.trainable
        grads, out = _param_grad_targets(trainable, ctx.needs_input_grad[2:], feat.device)
        d_feat = ops.criterion_train_backward([p.detach() for p in trainable], grads, g_terms.contiguous(), feat, verts, g, knn_idx, alpha,
                                              ctx.arena, topk, with_map, dist)
        ctx.arena = None
        return (None, d_feat if ctx.needs_input_grad[1] else None) + tuple(out)


def criterion_train(meta, feat, trainable):
    return _CriterionTrain.apply(meta, feat, *trainable)


def _param_grad_targets(trainable, need, device):
    """Where a native node adds its parameter gradients: the parameters' existing .grad buffers (-> autograd gets None), else views of one
    zeroed flat buffer (-> autograd accumulates them).  -> (buffers for the node, what backward returns)."""
    bufs = [_grad_buffer(p) if n else None for p, n in zip(trainable, need)]
    if all(need) and all(b is not None for b in bufs):
        return bufs, [None] * len(trainable)
    sizes = [p.numel() for p in trainable]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
    grads, off = [], 0
    for n in sizes:
        grads.append(flat[off:off + n])
        off += n
    return grads, [gr.view_as(p) if nd else None for gr, p, nd in zip(grads, trainable, need)]


class _CriterionDirTrain(torch.autograd.Function):
    """ONE direction of deform() for P pairs with N source and M target points as one native node (dvm_criterion_dir_train_{fwd,bwd}_f32):
    the partial-shape configs, whose two directions have different shapes.  meta = (verts_s, verts_t, source graph dict, knn_s, knn_t,
    alpha, topk, with_map); inputs feat_s (P,N,128), feat_t (P,M,128), the Deformer's 10 tensors.  -> terms (P,7)."""

    @staticmethod
    def forward(ctx, meta, feat_s, feat_t, *trainable):
        verts_s, verts_t, g, knn_s, knn_t, alpha, topk, with_map = meta
        terms, arena = ops.criterion_dir_train_forward([p.detach() for p in trainable], feat_s, feat_t, verts_s, verts_t, g, knn_s, knn_t, alpha, topk,
                                                       with_map)
        ctx.meta, ctx.arena, ctx.trainable = meta, arena, trainable
        ctx.save_for_backward(feat_s, feat_t)
        return terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_terms):
        if ctx.arena is None:
            raise RuntimeError("dvm criterion training node: backward ran already (its arena is released after the first pass)")
        verts_s, verts_t, g, knn_s, knn_t, alpha, topk, with_map = ctx.meta
        feat_s, feat_t = ctx.saved_tensors
        grads, out = _param_grad_targets(ctx.trainable, ctx.needs_input_grad[3:], feat_s.device)
        d_s, d_t = ops.criterion_dir_train_backward([p.detach() for p in ctx.trainable], grads, g_terms.contiguous(), feat_s.contiguous(),
                                                    feat_t.contiguous(), verts_s, verts_t, g, knn_s, knn_t, alpha, ctx.arena, topk, with_map)
        ctx.arena = None
        return (None, d_s if ctx.needs_input_grad[1] else None, d_t if ctx.needs_input_grad[2] else None) + tuple(out)


def criterion_dir_train(meta, feat_s, feat_t, trainable):
    return _CriterionDirTrain.apply(meta, feat_s, feat_t, *trainable)


def uni3fc_train(meta, x, dino, trainable):
    return _Uni3FCTrain.apply(meta, x, dino, *trainable)


def uni3fc_train_merged(meta, x1, dino1, x2, dino2, trainable):
    """The criterion's two network calls (train.py:100-101) as ONE native node with two groups (same point count): every row-wise
    kernel runs over both calls' rows at once, BatchNorm statistics / position-encoding ranges / running-statistics updates per
    call, in call order — the results of two separate calls (bit-identical forward) for half the launches.
    -> ((feat1, tmp1), (feat2, tmp2))."""
    B = x1.shape[0]
    meta = tuple(meta)
    feat, tmp = _Uni3FCTrain.apply(meta[:5] + (None, 2) + meta[7:8], torch.cat([x1, x2], 0), torch.cat([dino1, dino2], 0), *trainable)
    return (feat[:B], tmp[:B]), (feat[B:], tmp[B:])


def sa_attention_pm(xt, w_qk, w_v, b_v):
    """SA_Layer's x_r on point-major xt (B,N,64) with autograd: ONE projection GEMM (q/k and v stacked), the fused attention
    core both ways."""
    nq = w_qk.shape[0]
    C = xt.shape[-1]
    w = torch.cat([w_qk.reshape(nq, C), w_v.reshape(-1, C)], 0)
    pv = linear_pm(xt, w, torch.cat([b_v.new_zeros(nq), b_v]))
    return _SACore.apply(pv[..., :nq].contiguous(), pv[..., nq:].contiguous())


def n2p_attention_pm(xt, K, wq, wk, wv, heads):
    """N2PAttention's attention output on point-major xt (B,N,C) with autograd."""
    C = xt.shape[-1]
    idx = ops.knn_neg(xt, xt, K)
    w = torch.cat([wq.reshape(C, C), wk.reshape(C, C), wv.reshape(C, C)], 0)
    return _N2PCore.apply(linear_pm(xt, w), idx, heads)


def pos_encoding(coor, group=None, sync=False):
    """sync=True: normalise with the min/max over every rank's shard of the batch (two scalar all-reduces), so that a
    sharded batch is encoded exactly like the same batch in one process."""
    import torch.distributed as dist
    if sync and dist.is_initialized() and dist.get_world_size(group) > 1:
        from .dist import global_minmax
        mn, mx = global_minmax(coor, group)
        return ops.pos_encoding(coor, torch.cat([mn, mx]))
    return ops.pos_encoding(coor)


class _BNAct(torch.autograd.Function):
    """y = act(BatchNorm_train(x + res)) on the fused HIP kernels, forward and backward."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, momentum, eps, slope):
        y, mean, invstd = ops.bn_act_train_fwd(x, res, gamma, beta, eps, slope, momentum, running_mean, running_var)
        ctx.save_for_backward(x, res, y, gamma, mean, invstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, y, gamma, mean, invstd = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_act_train_bwd(dy.contiguous(), y, x, res, gamma, mean, invstd, ctx.slope)
        return dx, (dx if res is not None else None), (dgamma if gamma is not None else None), \
            (dbeta if gamma is not None else None), None, None, None, None, None


def bn_act(bn, x, res=None, slope=1.0):
    """act(bn(x + res)) for an nn.BatchNorm1d `bn` over (B,C,N); slope 1: no activation, 0: ReLU, 0.2: LeakyReLU.
    Training mode with plain batch statistics goes through the fused HIP kernels (the module's running statistics and
    batch counter are updated as nn.BatchNorm1d does); anything else — eval mode, SyncBatchNorm, no running stats —
    uses the module itself."""
    fused = (type(bn) is torch.nn.BatchNorm1d and bn.training and bn.track_running_stats and bn.affine
             and bn.momentum is not None and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32)
    if not fused:
        z = x if res is None else x + res
        y = bn(z)
        return y if slope == 1.0 else torch.nn.functional.leaky_relu(y, slope) if slope != 0.0 else torch.relu(y)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    return _BNAct.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, slope)


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class _SACore(torch.autograd.Function):
    """SA_Layer attention core on point-major projections, forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, p, v):
        xr, stats, cinv = ops.sa_attention_train_fwd(p, v)
        ctx.save_for_backward(p.detach(), v.detach(), xr, stats, cinv)
        return xr

    @staticmethod
    def backward(ctx, gxr):
        p, v, xr, stats, cinv = ctx.saved_tensors
        return ops.sa_attention_bwd(p, v, xr, stats, cinv, gxr.contiguous())


def sa_attention(x, w_qk, w_v, b_v):
    """SA_Layer's x_r (B,64,N).  The 1x1 projections are GEMMs under autograd; the attention core (energy,
    row softmax, column renormalisation, weighted sum) runs on the fused HIP kernels both ways — no N x N
    tensor in HBM."""
    if not _needs_grad(x, w_qk, w_v, b_v):
        return ops.sa_attention(x, w_qk, w_v, b_v)
    ops._need_gpu(x, w_qk, w_v)
    # both projections as ONE channel-major batched GEMM (K-contiguous operands forward and backward: the point-major
    # F.linear form makes the weight gradient a K-strided GEMM that runs ~10x slower), then a small transpose
    B, C, N = x.shape
    nq = w_qk.shape[0]
    w = torch.cat([w_qk.reshape(nq, C), w_v.reshape(-1, C)], 0)
    pv = conv1x1(x, w, torch.cat([b_v.new_zeros(nq), b_v])).transpose(1, 2)
    return _SACore.apply(pv[..., :nq].contiguous(), pv[..., nq:].contiguous()).transpose(1, 2)


class _N2PCore(torch.autograd.Function):
    """Attention of every point over its K gathered neighbours, forward and backward on the HIP kernels
    (dvm_n2p_core_{fwd,bwd}_f32); only the (B,N,K,heads) attention weights are kept for the backward."""

    @staticmethod
    def forward(ctx, qkv, idx, heads):
        out, attn = ops.n2p_core_fwd(qkv, idx, heads)
        ctx.save_for_backward(qkv.detach(), idx, attn)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, idx, attn = ctx.saved_tensors
        return ops.n2p_core_bwd(qkv, idx, attn, gout.contiguous(), ctx.heads), None, None


def n2p_attention(x, K, wq, wk, wv, heads):
    """N2PAttention's attention output.  The kNN indices come from the HIP kernel (no gradient, as in the
    reference); training projects q/k/v with one GEMM (k(x_j - x_i) = kp_j - kp_i by linearity) under
    autograd and runs the gather-attention core and its backward on the HIP kernels."""
    if not _needs_grad(x, wq, wk, wv):
        return ops.n2p_attention(x, K, wq, wk, wv, heads)
    B, C, N = x.shape
    xt = x.transpose(1, 2).contiguous()
    idx = ops.knn_neg(xt, xt, K)
    w = torch.cat([wq.reshape(C, C), wk.reshape(C, C), wv.reshape(C, C)], 0)
    qkv = conv1x1(x, w).transpose(1, 2).contiguous()   # (B,N,3C); channel-major GEMM, see sa_attention
    return _N2PCore.apply(qkv, idx, heads).transpose(1, 2)


# ------------------------------------------------------------------------------------------
# Differentiable pieces of the criterion (training path).
# Index searches and the fused soft correspondence run on the HIP kernels; the light, differentiable
# algebra around them (gathers, weighted sums, the Deformer's GEMMs, the warp) is expressed with
# torch ops on the device so that autograd provides its backward.  The soft correspondence has its own
# fused HIP backward (dvm_softcorr_bwd_f32).
# ------------------------------------------------------------------------------------------
class _SoftCorrTopK(torch.autograd.Function):
    """Top-k soft correspondence with its fused HIP backward (dvm_softcorr_bwd_f32: the dense N x M softmax
    term is recomputed tile by tile from row_smax / row_sum, never stored)."""

    @staticmethod
    def forward(ctx, f1, f2, alpha, topk):
        val, idx, smax, ssum = ops.softcorr(f1, f2, alpha, topk=topk)
        ctx.save_for_backward(f1.detach(), f2.detach(), val, idx, smax, ssum)
        ctx.alpha = alpha
        ctx.mark_non_differentiable(idx)
        return val, idx

    @staticmethod
    def backward(ctx, gval, _gidx):
        f1, f2, val, idx, smax, ssum = ctx.saved_tensors
        df1, df2 = ops.softcorr_bwd(f1, f2, ctx.alpha, val, idx, smax, ssum, gval.contiguous())
        return df1, df2, None, None


def softcorr_topk(f1, f2, alpha, topk=10):
    """Differentiable (w.r.t. f1, f2) top-k soft correspondence: (val (B,N,k), idx (B,N,k) int32)."""
    return _SoftCorrTopK.apply(f1, f2, alpha, topk)


class _SparseApply(torch.autograd.Function):
    """out[i] = sum_t val[i,t] V[idx[i,t]] on the HIP kernels, both ways (no (B,N,k,C) gather in HBM)."""

    @staticmethod
    def forward(ctx, val, idx, V):
        ctx.save_for_backward(val.detach(), idx, V.detach())
        return ops.apply(val, idx, V)

    @staticmethod
    def backward(ctx, gout):
        val, idx, V = ctx.saved_tensors
        dval, dV = ops.apply_bwd(val, idx, V, gout)
        return dval, None, dV


class _DistLoss(torch.autograd.Function):
    """dist-loss term (models/loss.py:1351-1396) per batch element: fused HIP forward; the backward builds the
    sparse (anchor, point) weights with a HIP kernel and finishes with two library GEMMs per batch."""

    @staticmethod
    def forward(ctx, feat, dist, anchors, k):
        out, idx = ops.dist_loss(feat, dist, anchors, k, want_idx=True)
        ctx.save_for_backward(feat.detach(), dist, anchors, idx)
        return out

    @staticmethod
    def backward(ctx, gout):
        feat, dist, anchors, idx = ctx.saved_tensors
        W = ops.dist_loss_bwd_weights(feat, dist, anchors, idx, gout.contiguous())   # (B,nA,N)
        a = anchors.long()
        fa = feat.index_select(1, a)    # (feat[:, a] runs ATen's general advanced-indexing kernel: 216 us for these 4 MB; index_select: ~10)
        dfeat = W.sum(1).unsqueeze(-1) * feat - torch.bmm(W.transpose(1, 2), fa)
        dfa = W.sum(2).unsqueeze(-1) * fa - torch.bmm(W, feat)
        dfeat.index_add_(1, a, dfa)
        return dfeat, None, None, None


def dist_loss(feat, dist, anchors, k):
    """(B,) sum over anchors of 1 - |cos(x, y)| with autograd w.r.t. feat."""
    return _DistLoss.apply(feat, dist, anchors, k)


def sparse_apply(val, idx, V):
    """Pi~ @ V with autograd: val (B,N,k), idx (B,N,k), V (B,M,C) -> (B,N,C)."""
    return _SparseApply.apply(val, idx, V)


def pool_rows(x, idx, w, bias):
    """Deformer's k -> 1 pooling conv: out[i] = sum_s w[s] x[idx[i,s]] + bias (models/model.py:466-470) — the
    same weighted gather-sum with row-independent weights (autograd sums the expanded weights' gradient)."""
    B, N, k = idx.shape
    return _SparseApply.apply(w.reshape(1, 1, k).expand(B, N, k), idx, x) + bias


def gather_rows(x, idx):
    """x (B,P,C), idx (B,N,k) -> (B,N,k,C)."""
    B, N, k = idx.shape
    return torch.gather(x, 1, idx.long().reshape(B, N * k, 1).expand(-1, -1, x.shape[-1])).view(B, N, k, -1)


class _ChamferNN(torch.autograd.Function):
    """Squared nearest-neighbour distances both ways; gradients flow to both clouds through the
    (constant) arg-min indices, like the ChamferDistancePytorch extension's backward."""

    @staticmethod
    def forward(ctx, a, b):
        d1, d2, i1, i2 = ops.chamfer(a, b)
        ctx.save_for_backward(a.detach(), b.detach(), i1, i2)
        return d1, d2

    @staticmethod
    def backward(ctx, g1, g2):
        a, b, i1, i2 = ctx.saved_tensors
        return ops.chamfer_bwd(a, b, i1, i2, g1.contiguous(), g2.contiguous())


def chamfer_nn(a, b):
    return _ChamferNN.apply(a, b)


class _Rot6D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d6):
        ctx.save_for_backward(d6.detach())
        return ops.rot6d(d6)

    @staticmethod
    def backward(ctx, gR):
        (d6,) = ctx.saved_tensors
        return ops.rot6d_bwd(d6, gR.contiguous())


def rot6d(d6):
    """rotation_6d_to_matrix with autograd, both ways on the HIP kernels: (...,6) -> (...,3,3)."""
    return _Rot6D.apply(d6)


class _WarpArap(torch.autograd.Function):
    """Embedded-deformation warp + ARAP (lib/deformation_graph_point.py:233-261), differentiable w.r.t. the
    node rotations / translations; forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, verts, R, T, nodes_idx, one_ring, infl_idx, weights):
        g = dict(nodes_idx=nodes_idx, one_ring=one_ring, infl_idx=infl_idx, weights=weights)
        warped, arap, _ = ops.dg_warp_arap(verts, g, R, T)
        ctx.save_for_backward(verts, R.detach(), T.detach(), nodes_idx, one_ring, infl_idx, weights)
        return warped, arap

    @staticmethod
    def backward(ctx, gw, ga):
        verts, R, T, nodes_idx, one_ring, infl_idx, weights = ctx.saved_tensors
        g = dict(nodes_idx=nodes_idx, one_ring=one_ring, infl_idx=infl_idx, weights=weights)
        dR, dT = ops.dg_warp_arap_bwd(verts, g, R, T, gw.contiguous(), ga.contiguous())
        return None, dR, dT, None, None, None, None


def dg_warp_arap(verts, g, R, T):
    """verts (B,N,3), batched graph dict, R (B,Nn,3,3), T (B,Nn,3) -> warped (B,N,3), arap (B,), with autograd."""
    return _WarpArap.apply(verts, R, T.contiguous(), g["nodes_idx"], g["one_ring"], g["infl_idx"], g["weights"])
