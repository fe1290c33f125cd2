"""Tensor-level wrappers over the C ABI (include/dvm.h).

torch is used for device memory and streams only: every function takes
contiguous fp32 / int32 tensors that live on a HIP device, allocates the outputs
and the scratch there, and enqueues the kernels on torch's current stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import DvmError, check

_ws_cache = {}
_pair_ctx = set()


def _need_gpu(*ts):
    """Every wrapper starts here: all tensors on ONE HIP device, which becomes the current device (the C ABI launches on
    `torch.cuda.current_stream()` and keeps its per-device state — kernel attributes, helper streams — by current device).
    (This and the two casts below run ~1500 times per training step on the host: they stay on attribute reads.)"""
    dev = -1
    for t in ts:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise DvmError("dvm ops need tensors on a HIP device (got %s); there is no CPU fallback"
                           % (t.device if isinstance(t, torch.Tensor) else type(t)))
        d = t.get_device()
        if dev < 0:
            dev = d
        elif d != dev:
            raise DvmError("dvm ops need all tensors on one device (got cuda:%d and cuda:%d)" % (dev, d))
    if dev >= 0 and dev != torch.cuda.current_device():
        torch.cuda.set_device(dev)


def _f(t):
    """A contiguous fp32 tensor without autograd history (the same object when it already is one)."""
    if t.dtype is torch.float32 and not t.requires_grad and t.is_contiguous():
        return t
    return t.detach().contiguous().float()


def _i(t):
    if t.dtype is torch.int32 and t.is_contiguous():
        return t
    return t.detach().contiguous().to(torch.int32)


def _p(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current stream on the current device as the raw hipStream_t (no Stream object: this runs per launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, current stream, tag): calls on one stream are ordered and may share it,
    calls on different streams (or devices) never do."""
    dev_index = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
    key = (dev_index, _raw_stream(dev_index) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def neg_alpha_f32(alpha):
    """`-alpha * distance`: ATen casts the python/numpy scalar to fp32 (models/loss.py:112)."""
    return float(torch.tensor(-float(alpha), dtype=torch.float32).item())


def rownorm2(x):
    _need_gpu(x)
    x = _f(x)
    rows, K = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    check(_lib.load().dvm_rownorm2_f32(_p(x), rows, K, _p(out), _stream()), "dvm_rownorm2_f32")
    return out


def linear(x, w, bias=None, res=None, bn=None, slope=1.0, channel_major=False, out=None, prefix=None, post=None):
    """1x1 conv / linear layer with its fused epilogue (dvm_linear_f32): act(bn(x W^T + bias + res)).
    point-major (default): x (..., K) -> (..., Co); channel_major: x (B,K,N) -> (B,Co,N) (nn.Conv1d's layout).
    w (Co,K[,1]); bn = (alpha, beta) of the eval-mode BatchNorm (models.model._bn_affine); slope 1 = no activation,
    0 = ReLU, else LeakyReLU.  The contraction is the reference's single-thread fp32 chain, bit for bit.
    post = (scale, r): the JBU "fixup" form r + scale * (x W^T + bias) in the same launch (dvm_linear_scaled_residual_f32;
    excludes res / bn / slope / prefix)."""
    _need_gpu(x, w, bias, res)
    x, w = _f(x), _f(w)
    Co = w.shape[0]
    w = w.reshape(Co, -1)
    K = w.shape[1]
    Cg = 0
    if channel_major:
        B, Kx, N = x.shape
        shape = (B, Co, N)
    elif prefix is not None:   # rows [prefix[b] | x[b, n]]: a conv over cat((g broadcast over the points, x), channels)
        _need_gpu(prefix)
        B, N, Kx = x.shape
        prefix = _f(prefix).reshape(B, -1)
        Cg = prefix.shape[1]
        Kx += Cg
        shape = (B, N, Co)
    else:
        Kx = x.shape[-1]
        B, N = 1, x.numel() // Kx
        shape = x.shape[:-1] + (Co,)
    if Kx != K:
        raise DvmError("linear: x has %d input channels, w expects %d" % (Kx, K))
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != tuple(shape) or not out.is_contiguous() or out.dtype != torch.float32:
        raise DvmError("linear: bad `out`")
    bias = None if bias is None else _f(bias)
    res = None if res is None else _f(res)
    if res is not None and tuple(res.shape) != tuple(shape):
        raise DvmError("linear: residual shape %s != output shape %s" % (tuple(res.shape), tuple(shape)))
    al, be = (None, None) if bn is None else (_f(bn[0]), _f(bn[1]))
    if post is not None:
        if res is not None or bn is not None or slope != 1.0 or Cg:
            raise DvmError("linear: `post` excludes res / bn / slope / prefix")
        scale, r = post
        _need_gpu(r)
        r = _f(r)
        if tuple(r.shape) != tuple(shape):
            raise DvmError("linear: post residual shape %s != output shape %s" % (tuple(r.shape), tuple(shape)))
        check(_lib.load().dvm_linear_scaled_residual_f32(_p(x), _p(w), B, N, K, Co, 1 if channel_major else 0, _p(bias), float(scale),
                                                         _p(r), _p(out), _stream()), "dvm_linear_scaled_residual_f32")
        return out
    if Cg:
        check(_lib.load().dvm_linear_prefix_f32(_p(prefix), Cg, _p(x), _p(w), B, N, K, Co, _p(bias), _p(res), _p(al), _p(be),
                                                float(slope), _p(out), _stream()), "dvm_linear_prefix_f32")
        return out
    check(_lib.load().dvm_linear_f32(_p(x), _p(w), B, N, K, Co, 1 if channel_major else 0, _p(bias), _p(res), _p(al), _p(be),
                                     float(slope), _p(out), _stream()), "dvm_linear_f32")
    return out


def set_deterministic(on=True):
    """Fixed summation order in LG-Net's backward (dvm_set_deterministic: ordered weight-gradient partials, SA backward without its
    split, sorted in-edge lists): parameter gradients are bit-reproducible from run to run.  Returns the previous setting."""
    return bool(_lib.load().dvm_set_deterministic(1 if on else 0))


def is_deterministic():
    """The library's current setting (dvm_get_deterministic: the flag lives in the library — DVM_DETERMINISTIC, dvm_set_deterministic
    from any caller — and is only READ here)."""
    return bool(_lib.load().dvm_get_deterministic())


def linear_wgrad(gy, x, out=None):
    """dW = gy^T x over the rows: gy (R,Co), x (R,K) -> (Co,K) (dvm_linear_wgrad_f32; fp32 atomics over row chunks).
    With `out` (Co,K) the product is ADDED to it in place."""
    _need_gpu(gy, x, out)
    gy, x = _f(gy), _f(x)
    R, Co = gy.shape
    K = x.shape[1]
    if out is None:
        dW = torch.zeros(Co, K, dtype=torch.float32, device=gy.device)
    else:
        dW = out
        if dW.dtype != torch.float32 or not dW.is_contiguous() or dW.numel() != Co * K:
            raise ValueError("linear_wgrad: out must be a contiguous float32 tensor of %d x %d elements" % (Co, K))
    lib = _lib.load()
    if lib.dvm_get_deterministic():   # per-chunk partial tiles added in chunk order (needs scratch)
        nb = lib.dvm_linear_wgrad_workspace_bytes(R, Co, K)
        ws = workspace(nb, gy.device, "wgrad")
        check(lib.dvm_linear_wgrad_ws_f32(_p(gy), _p(x), R, Co, K, _p(dW), _p(ws), nb, _stream()), "dvm_linear_wgrad_ws_f32")
    else:
        check(lib.dvm_linear_wgrad_f32(_p(gy), _p(x), R, Co, K, _p(dW), _stream()), "dvm_linear_wgrad_f32")
    return dW


_bn_pm_sizes = {}


def _bn_pm_bytes(lib, R, C):
    nb = _bn_pm_sizes.get((R, C))
    if nb is None:
        nb = _bn_pm_sizes[(R, C)] = lib.dvm_bn_pm_workspace_bytes(R, C)
    return nb


def bn_act_train_fwd_pm(x, res, gamma, beta, eps, slope, momentum, running_mean=None, running_var=None):
    """Fused training-mode BatchNorm on point-major x (..., C): y = act(bn(x + res)); returns (y, mean, invstd)."""
    _need_gpu(x, res, gamma, beta)
    x = _f(x)
    C = x.shape[-1]
    R = x.numel() // C
    res = None if res is None else _f(res)
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    lib = _lib.load()
    nb = _bn_pm_bytes(lib, R, C)
    ws = workspace(nb, x.device, "bn_pm")
    check(lib.dvm_bn_act_train_fwd_pm_f32(_p(x), _p(res), _p(gamma), _p(beta), R, C, float(eps), float(slope), float(momentum), _p(y), _p(mean),
                                          _p(invstd), _p(running_mean), _p(running_var), _p(ws), nb, _stream()), "dvm_bn_act_train_fwd_pm_f32")
    return y, mean, invstd


def bn_act_train_bwd_pm(dy, y, x, res, gamma, mean, invstd, slope, grads=None):
    """-> (dx, dgamma, dbeta); with grads = (dgamma, dbeta) given, the parameter gradients are ADDED to those tensors."""
    _need_gpu(dy, y, x)
    dy, x = _f(dy), _f(x)
    C = x.shape[-1]
    R = x.numel() // C
    dx = torch.empty_like(x)
    if grads is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
    else:
        dgamma, dbeta = grads
        for t in grads:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C:
                raise ValueError("bn_act_train_bwd_pm: grads must be contiguous float32 tensors of %d elements" % C)
    lib = _lib.load()
    nb = _bn_pm_bytes(lib, R, C)
    ws = workspace(nb, x.device, "bn_pm")
    check(lib.dvm_bn_act_train_bwd_pm_f32(_p(dy), _p(y), _p(x), _p(res), _p(gamma), _p(mean), _p(invstd), R, C, float(slope), _p(dx), _p(dgamma),
                                          _p(dbeta), int(grads is not None), _p(ws), nb, _stream()), "dvm_bn_act_train_bwd_pm_f32")
    return dx, dgamma, dbeta


def softcorr(f1, f2, alpha, topk=10, variant=0, stats=True):
    """f1 (B,N,d), f2 (B,M,d) -> pi_val (B,N,topk), pi_idx (B,N,topk) int32, row_smax (B,N), row_sum (B,N)."""
    _need_gpu(f1, f2)
    f1, f2 = _f(f1), _f(f2)
    B, N, d = f1.shape
    M = f2.shape[1]
    lib = _lib.load()
    val = torch.empty(B, N, topk, dtype=torch.float32, device=f1.device)
    idx = torch.empty(B, N, topk, dtype=torch.int32, device=f1.device)
    smax = torch.empty(B, N, dtype=torch.float32, device=f1.device) if stats else None
    ssum = torch.empty(B, N, dtype=torch.float32, device=f1.device) if stats else None
    nb = lib.dvm_softcorr_workspace_bytes(B, N, M, d)
    ws = workspace(nb, f1.device, "softcorr")
    check(lib.dvm_softcorr_fwd_f32(_p(f1), _p(f2), B, N, M, d, neg_alpha_f32(alpha), topk, _p(val), _p(idx), _p(smax),
                                   _p(ssum), variant, _p(ws), nb, _stream()), "dvm_softcorr_fwd_f32")
    return val, idx, smax, ssum


def softcorr_bwd(f1, f2, alpha, val, idx, smax, ssum, gval, variant=0):
    """Backward of softcorr: gval (B,N,topk) -> (d_f1 (B,N,d), d_f2 (B,M,d))."""
    _need_gpu(f1, f2, gval)
    f1, f2, gval, val = _f(f1), _f(f2), _f(gval), _f(val)
    B, N, d = f1.shape
    M = f2.shape[1]
    topk = val.shape[-1]
    lib = _lib.load()
    df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
    nb = lib.dvm_softcorr_bwd_workspace_bytes(B, N, M, d)
    ws = workspace(nb, f1.device, "softcorr_bwd")
    check(lib.dvm_softcorr_bwd_f32(_p(f1), _p(f2), B, N, M, d, neg_alpha_f32(alpha), topk, _p(val), _p(idx.contiguous()),
                                   _p(smax), _p(ssum), _p(gval), _p(df1), _p(df2), variant, _p(ws), nb, _stream()),
          "dvm_softcorr_bwd_f32")
    return df1, df2


def n2p_core_fwd(qkv, idx, heads=4):
    """qkv (B,N,3C), idx (B,N,K) int32 -> out (B,N,C), attn (B,N,K,heads)."""
    _need_gpu(qkv, idx)
    qkv = _f(qkv)
    B, N, C3 = qkv.shape
    C, K = C3 // 3, idx.shape[-1]
    out = torch.empty(B, N, C, dtype=torch.float32, device=qkv.device)
    attn = torch.empty(B, N, K, heads, dtype=torch.float32, device=qkv.device)
    check(_lib.load().dvm_n2p_core_fwd_f32(_p(qkv), _p(idx.contiguous()), B, N, C, K, heads, _p(out), _p(attn), _stream()),
          "dvm_n2p_core_fwd_f32")
    return out, attn


def n2p_core_bwd(qkv, idx, attn, gout, heads=4):
    """-> d_qkv (B,N,3C)."""
    _need_gpu(qkv, idx, attn, gout)
    qkv, gout = _f(qkv), _f(gout)
    B, N, C3 = qkv.shape
    C, K = C3 // 3, idx.shape[-1]
    dqkv = torch.empty_like(qkv)
    lib = _lib.load()
    nb = lib.dvm_n2p_core_bwd_workspace_bytes(B, N, K)
    ws = workspace(nb, qkv.device, "n2p_bwd")
    check(lib.dvm_n2p_core_bwd_f32(_p(qkv), _p(idx.contiguous()), _p(attn), _p(gout), B, N, C, K, heads, _p(dqkv), _p(ws), nb,
                                   _stream()), "dvm_n2p_core_bwd_f32")
    return dqkv


def argmin_exact(f1, f2, want_dist=False, screen=True):
    """Hard map T[b,i] = argmin_j of the exact-difference distance (ties -> lowest j).  screen=False evaluates every
    column in that form; the default screens the columns on the matrix cores first (same result)."""
    _need_gpu(f1, f2)
    f1, f2 = _f(f1), _f(f2)
    B, N, d = f1.shape
    M = f2.shape[1]
    lib = _lib.load()
    T = torch.empty(B, N, dtype=torch.int32, device=f1.device)
    dm = torch.empty(B, N, dtype=torch.float32, device=f1.device) if want_dist else None
    nb = lib.dvm_argmin_workspace_bytes(B, N, M, d, 0) if screen else 0
    ws = workspace(nb, f1.device, "argmin") if nb else None
    check(lib.dvm_argmin_exact_f32(_p(f1), _p(f2), B, N, M, d, _p(T), _p(dm), _p(ws) if nb else None, nb, _stream()),
          "dvm_argmin_exact_f32")
    return (T, dm) if want_dist else T


def argmin_pair(f1, f2):
    """Both hard maps of a pair from one sweep: (T12 (B,N) into f2, T21 (B,M) into f1)."""
    _need_gpu(f1, f2)
    f1, f2 = _f(f1), _f(f2)
    B, N, d = f1.shape
    M = f2.shape[1]
    lib = _lib.load()
    T12 = torch.empty(B, N, dtype=torch.int32, device=f1.device)
    T21 = torch.empty(B, M, dtype=torch.int32, device=f1.device)
    nb = lib.dvm_argmin_workspace_bytes(B, N, M, d, 1)
    ws = workspace(nb, f1.device, "argmin") if nb else None
    check(lib.dvm_argmin_pair_f32(_p(f1), _p(f2), B, N, M, d, _p(T12), _p(T21), None, None, _p(ws) if nb else None, nb, _stream()),
          "dvm_argmin_pair_f32")
    return T12, T21


def knn_cdist(x, y, k):
    _need_gpu(x, y)
    x, y = _f(x), _f(y)
    B, N, C = x.shape
    M = y.shape[1]
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    lib = _lib.load()
    nb = lib.dvm_knn_cdist_workspace_bytes(B, N, M, C)
    ws = workspace(nb, x.device, "knn_cdist")
    check(lib.dvm_knn_cdist_f32(_p(x), _p(y), B, N, M, C, k, _p(idx), _p(ws), nb, _stream()), "dvm_knn_cdist_f32")
    return idx


def apply(pi_val, pi_idx, V):
    _need_gpu(pi_val, pi_idx, V)
    pi_val, pi_idx, V = _f(pi_val), _i(pi_idx), _f(V)
    B, N, topk = pi_val.shape
    M, C = V.shape[1], V.shape[2]
    out = torch.empty(B, N, C, dtype=torch.float32, device=V.device)
    check(_lib.load().dvm_softcorr_apply_f32(_p(pi_val), _p(pi_idx), _p(V), B, N, M, topk, C, _p(out), _stream()),
          "dvm_softcorr_apply_f32")
    return out


def apply_bwd(pi_val, pi_idx, V, gout, atomics=False):
    """Backward of apply: -> (d_val (B,N,topk), d_V (B,M,C)).  atomics=True: the scatter-add variant (no workspace)."""
    _need_gpu(pi_val, pi_idx, V, gout)
    pi_val, pi_idx, V, gout = _f(pi_val), _i(pi_idx), _f(V), _f(gout)
    B, N, topk = pi_val.shape
    M, C = V.shape[1], V.shape[2]
    dval = torch.empty_like(pi_val)
    dV = torch.empty_like(V)
    lib = _lib.load()
    nb = 0 if atomics else lib.dvm_softcorr_apply_bwd_workspace_bytes(B, N, M, topk)
    ws = workspace(nb, V.device, "apply_bwd") if nb else None
    check(lib.dvm_softcorr_apply_bwd_f32(_p(pi_val), _p(pi_idx), _p(V), _p(gout), B, N, M, topk, C, _p(dval), _p(dV),
                                         _p(ws) if nb else None, nb, _stream()), "dvm_softcorr_apply_bwd_f32")
    return dval, dV


def fps(xyz, npoint, start):
    _need_gpu(xyz, start)
    xyz, start = _f(xyz), _i(start)
    B, N, _ = xyz.shape
    out = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    check(_lib.load().dvm_fps_f32(_p(xyz), B, N, npoint, _p(start), _p(out), _stream()), "dvm_fps_f32")
    return out


def dg_build(xyz, start):
    """xyz (B,N,3), start (B,) -> dict(nodes_idx, one_ring, infl_idx, dists, weights, sigma)."""
    _need_gpu(xyz, start)
    xyz, start = _f(xyz), _i(start)
    B, N, _ = xyz.shape
    Nn = N // 2
    dev = xyz.device
    lib = _lib.load()
    out = dict(nodes_idx=torch.empty(B, Nn, dtype=torch.int32, device=dev),
               one_ring=torch.empty(B, Nn, 9, dtype=torch.int32, device=dev),
               infl_idx=torch.empty(B, N, 3, dtype=torch.int32, device=dev),
               dists=torch.empty(B, N, 3, dtype=torch.float32, device=dev),
               weights=torch.empty(B, N, 3, dtype=torch.float32, device=dev),
               sigma=torch.empty(B, dtype=torch.float64, device=dev))
    nb = lib.dvm_dg_build_workspace_bytes(B, N)
    ws = workspace(nb, dev, "dg_build")
    check(lib.dvm_dg_build_f32(_p(xyz), B, N, _p(start), _p(out["nodes_idx"]), _p(out["one_ring"]), _p(out["infl_idx"]),
                               _p(out["dists"]), _p(out["weights"]), _p(out["sigma"]), _p(ws), nb, _stream()),
          "dvm_dg_build_f32")
    return out


def rot6d(d6):
    """rotation_6d_to_matrix: d6 (...,6) -> (...,3,3)."""
    _need_gpu(d6)
    d6 = _f(d6)
    rows = d6.numel() // 6
    R = torch.empty(*d6.shape[:-1], 3, 3, dtype=torch.float32, device=d6.device)
    check(_lib.load().dvm_rot6d_f32(_p(d6), rows, _p(R), _stream()), "dvm_rot6d_f32")
    return R


def rot6d_bwd(d6, gR):
    _need_gpu(d6, gR)
    d6, gR = _f(d6), _f(gR)
    out = torch.empty_like(d6)
    check(_lib.load().dvm_rot6d_bwd_f32(_p(d6), _p(gR), d6.numel() // 6, _p(out), _stream()), "dvm_rot6d_bwd_f32")
    return out


def dg_warp_arap_bwd(xyz, g, R, T, gw, ga):
    """-> (d_R (B,Nn,3,3), d_T (B,Nn,3))."""
    _need_gpu(xyz, R, T, gw, ga)
    xyz, R, T, gw, ga = _f(xyz), _f(R), _f(T), _f(gw), _f(ga)
    B, N, _ = xyz.shape
    dR, dT = torch.empty_like(R), torch.empty_like(T)
    check(_lib.load().dvm_dg_warp_arap_bwd_f32(_p(xyz), B, N, _p(_i(g["nodes_idx"])), _p(_i(g["one_ring"])),
                                               _p(_i(g["infl_idx"])), _p(_f(g["weights"])), _p(R), _p(T), _p(gw), _p(ga),
                                               _p(dR), _p(dT), _stream()), "dvm_dg_warp_arap_bwd_f32")
    return dR, dT


def chamfer_bwd(a, b, i1, i2, g1, g2):
    _need_gpu(a, b, g1, g2)
    a, b, g1, g2 = _f(a), _f(b), _f(g1), _f(g2)
    B, N, _ = a.shape
    M = b.shape[1]
    da, db = torch.empty_like(a), torch.empty_like(b)
    check(_lib.load().dvm_chamfer_bwd_f32(_p(a), _p(b), _p(_i(i1)), _p(_i(i2)), _p(g1), _p(g2), B, N, M, _p(da), _p(db),
                                          _stream()), "dvm_chamfer_bwd_f32")
    return da, db


def dg_warp_arap(xyz, g, R, T):
    """xyz (B,N,3), graph dict, R (B,Nn,3,3), T (B,Nn,3) -> warped (B,N,3), arap (B,), sr (B,)."""
    _need_gpu(xyz, R, T)
    xyz, R, T = _f(xyz), _f(R), _f(T)
    B, N, _ = xyz.shape
    dev = xyz.device
    warped = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
    arap = torch.empty(B, dtype=torch.float32, device=dev)
    sr = torch.empty(B, dtype=torch.float32, device=dev)
    check(_lib.load().dvm_dg_warp_arap_fwd_f32(_p(xyz), B, N, _p(_i(g["nodes_idx"])), _p(_i(g["one_ring"])),
                                               _p(_i(g["infl_idx"])), _p(_f(g["weights"])), _p(R), _p(T), _p(warped),
                                               _p(arap), _p(sr), _stream()), "dvm_dg_warp_arap_fwd_f32")
    return warped, arap, sr


def dg_warp_arap_graph(xyz, g, R, T):
    """As dg_warp_arap for a graph whose node count and ring width are whatever g holds (the mesh-mode graph:
    nodes_idx (B,Nn), one_ring (B,Nn,K), infl_idx / weights (B,N,3))."""
    _need_gpu(xyz, R, T)
    xyz, R, T = _f(xyz), _f(R), _f(T)
    B, N, _ = xyz.shape
    nodes, ring = _i(g["nodes_idx"]), _i(g["one_ring"])
    Nn, K = nodes.shape[1], ring.shape[2]
    dev = xyz.device
    warped = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
    arap = torch.empty(B, dtype=torch.float32, device=dev)
    sr = torch.empty(B, dtype=torch.float32, device=dev)
    check(_lib.load().dvm_dg_warp_arap_graph_f32(_p(xyz), B, N, Nn, K, _p(nodes), _p(ring), _p(_i(g["infl_idx"])),
                                                 _p(_f(g["weights"])), _p(R), _p(T), _p(warped), _p(arap), _p(sr), _stream()),
          "dvm_dg_warp_arap_graph_f32")
    return warped, arap, sr


def chamfer(a, b, want_idx=True):
    _need_gpu(a, b)
    a, b = _f(a), _f(b)
    B, N, _ = a.shape
    M = b.shape[1]
    dev = a.device
    d1 = torch.empty(B, N, dtype=torch.float32, device=dev)
    d2 = torch.empty(B, M, dtype=torch.float32, device=dev)
    i1 = torch.empty(B, N, dtype=torch.int32, device=dev) if want_idx else None
    i2 = torch.empty(B, M, dtype=torch.int32, device=dev) if want_idx else None
    lib = _lib.load()
    nb = lib.dvm_chamfer_workspace_bytes(B, N, M)
    ws = workspace(nb, dev, "chamfer")
    check(lib.dvm_chamfer_fwd_f32(_p(a), _p(b), B, N, M, _p(d1), _p(d2), _p(i1), _p(i2), _p(ws), nb, _stream()),
          "dvm_chamfer_fwd_f32")
    return d1, d2, i1, i2


DEFORMER_KEYS = ["conv_layer.weight", "conv_layer.bias"] + [
    "deformation_decoder_layer.linear.%d.%s" % (li, wb) for li in (0, 2, 4, 6) for wb in ("weight", "bias")]


def deformer_weight_list(weights, device):
    """state_dict (reference key names; '.' may be '__') -> the 10 contiguous fp32 device tensors of the ABI."""
    out = []
    for k in DEFORMER_KEYS:
        v = weights[k] if k in weights else weights[k.replace(".", "__")]
        v = torch.as_tensor(v)
        out.append(v.detach().to(device=device, dtype=torch.float32).contiguous().reshape(-1) if "conv_layer" in k
                   else v.detach().to(device=device, dtype=torch.float32).contiguous())
    return out


def deformer(wl, feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1, variant=0):
    _need_gpu(feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1, *wl)
    feat1, feat2, verts1, verts12, pi_val = _f(feat1), _f(feat2), _f(verts1), _f(verts12), _f(pi_val)
    idx11, idx22, pi_idx, fps1 = _i(idx11), _i(idx22), _i(pi_idx), _i(fps1)
    B, N, _ = feat1.shape
    M = feat2.shape[1]
    Nn, k, topk = fps1.shape[1], idx11.shape[2], pi_val.shape[2]
    lib = _lib.load()
    out = torch.empty(B, Nn, 9, dtype=torch.float32, device=feat1.device)
    nb = lib.dvm_deformer_workspace_bytes(B, N, M, Nn)
    ws = workspace(nb, feat1.device, "deformer")
    check(lib.dvm_deformer_fwd_f32(_p(feat1), _p(feat2), _p(verts1), _p(verts12), _p(idx11), _p(idx22), _p(pi_val),
                                   _p(pi_idx), _p(fps1), B, N, M, Nn, k, topk, *[_p(w) for w in wl], _p(out), variant,
                                   _p(ws), nb, _stream()), "dvm_deformer_fwd_f32")
    return out


def map_term(verts12, verts2, idx11, idx22, pi_val, pi_idx):
    _need_gpu(verts12, verts2, idx11, idx22, pi_val, pi_idx)
    verts12, verts2, pi_val = _f(verts12), _f(verts2), _f(pi_val)
    idx11, idx22, pi_idx = _i(idx11), _i(idx22), _i(pi_idx)
    B, N, _ = verts12.shape
    M, k, topk = verts2.shape[1], idx11.shape[2], pi_val.shape[2]
    lib = _lib.load()
    out = torch.empty(B, dtype=torch.float32, device=verts12.device)
    nb = lib.dvm_map_term_workspace_bytes(B, N)
    ws = workspace(nb, verts12.device, "map")
    check(lib.dvm_map_term_f32(_p(verts12), _p(verts2), _p(idx11), _p(idx22), _p(pi_val), _p(pi_idx), B, N, M, k, topk,
                               _p(out), _p(ws), nb, _stream()), "dvm_map_term_f32")
    return out


def pair_direction(wl, feat1, feat2, verts1, verts2, alpha, fps_start, with_map=True, out=None):
    """Config-2 path for B pairs, one direction. Returns dict(warped, verts12, T12, losses[B,6])."""
    _need_gpu(feat1, feat2, verts1, verts2, fps_start, *wl)
    feat1, feat2, verts1, verts2, fps_start = _f(feat1), _f(feat2), _f(verts1), _f(verts2), _i(fps_start)
    B, N, _ = feat1.shape
    M = feat2.shape[1]
    dev = feat1.device
    lib = _lib.load()
    if out is None:
        out = dict(warped=torch.empty(B, N, 3, dtype=torch.float32, device=dev),
                   verts12=torch.empty(B, N, 3, dtype=torch.float32, device=dev),
                   T12=torch.empty(B, N, dtype=torch.int32, device=dev),
                   losses=torch.empty(B, 6, dtype=torch.float32, device=dev))
    nb = lib.dvm_pair_direction_workspace_bytes(B, N, M)
    ws = workspace(nb, dev, "pair")
    check(lib.dvm_pair_direction_fwd_f32(_p(feat1), _p(feat2), _p(verts1), _p(verts2), B, N, M, neg_alpha_f32(alpha),
                                         _p(fps_start), *[_p(w) for w in wl], int(with_map), _p(out["warped"]),
                                         _p(out["verts12"]), _p(out["T12"]), _p(out["losses"]), _p(ws), nb, _stream()),
          "dvm_pair_direction_fwd_f32")
    return out


def knn_neg(a, b, k):
    """knn_new / knn: a (B,N,C), b (B,M,C) -> idx (B,N,k) int32, nearest (largest score) first."""
    _need_gpu(a, b)
    a, b = _f(a), _f(b)
    B, N, C = a.shape
    M = b.shape[1]
    lib = _lib.load()
    idx = torch.empty(B, N, k, dtype=torch.int32, device=a.device)
    nb = lib.dvm_knn_neg_workspace_bytes(B, N, M, C, k)
    ws = workspace(nb, a.device, "knn_neg")
    check(lib.dvm_knn_neg_f32(_p(a), _p(b), B, N, M, C, k, _p(idx), _p(ws), nb, _stream()), "dvm_knn_neg_f32")
    return idx


def softcorr_dense(f1, f2, alpha):
    _need_gpu(f1, f2)
    f1, f2 = _f(f1), _f(f2)
    B, N, d = f1.shape
    M = f2.shape[1]
    lib = _lib.load()
    P = torch.empty(B, N, M, dtype=torch.float32, device=f1.device)
    nb = lib.dvm_softcorr_dense_workspace_bytes(B, N, M, d)
    ws = workspace(nb, f1.device, "softcorr_dense")
    check(lib.dvm_softcorr_dense_f32(_p(f1), _p(f2), B, N, M, d, neg_alpha_f32(alpha), _p(P), _p(ws), nb, _stream()),
          "dvm_softcorr_dense_f32")
    return P


def deformer_mlp(wl, z):
    """z (..., 262) -> (..., 9) with the Deformer's decoder weights wl (deformer_weight_list)."""
    _need_gpu(z, *wl)
    z = _f(z)
    rows = z.numel() // 262
    lib = _lib.load()
    out = torch.empty(*z.shape[:-1], 9, dtype=torch.float32, device=z.device)
    nb = lib.dvm_deformer_mlp_workspace_bytes(rows)
    ws = workspace(nb, z.device, "mlp")
    check(lib.dvm_deformer_mlp_fwd_f32(_p(z), rows, *[_p(w) for w in wl[2:]], _p(out), _p(ws), nb, _stream()),
          "dvm_deformer_mlp_fwd_f32")
    return out


def pos_encoding(x, minmax=None):
    """x (B,3,N) -> (B,384,N).  minmax: optional device tensor (min, max) replacing the tensor's own range."""
    _need_gpu(x)
    x = _f(x)
    B, _, N = x.shape
    lib = _lib.load()
    out = torch.empty(B, 384, N, dtype=torch.float32, device=x.device)
    if minmax is not None:
        _need_gpu(minmax)
        mm = _f(minmax).reshape(2)
        check(lib.dvm_pos_encoding_minmax_f32(_p(x), _p(mm), B, N, _p(out), _stream()), "dvm_pos_encoding_minmax_f32")
        return out
    nb = lib.dvm_pos_encoding_workspace_bytes()
    ws = workspace(nb, x.device, "posenc")
    check(lib.dvm_pos_encoding_f32(_p(x), B, N, _p(out), _p(ws), nb, _stream()), "dvm_pos_encoding_f32")
    return out


def bn_act_train_fwd(x, res, gamma, beta, eps, slope, momentum, running_mean=None, running_var=None):
    """Training-mode BatchNorm over (B,C,N) of x (+ res) with the activation fused -> (y, save_mean, save_invstd);
    running_mean / running_var (device tensors) are updated in place."""
    _need_gpu(x)
    x = _f(x)
    res = None if res is None else _f(res)
    B, C, N = x.shape
    lib = _lib.load()
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = lib.dvm_bn_workspace_bytes(B, C, N)
    ws = workspace(nb, x.device, "bn")
    opt = lambda t: None if t is None else _p(t)  # noqa: E731
    if running_mean is not None:
        assert running_mean.is_contiguous() and running_var.is_contiguous() and running_mean.dtype == torch.float32
    check(lib.dvm_bn_act_train_fwd_f32(_p(x), opt(res), opt(None if gamma is None else _f(gamma)), opt(None if beta is None else _f(beta)),
                                       B, C, N, float(eps), float(slope), float(momentum), _p(y), _p(mean), _p(invstd),
                                       opt(running_mean), opt(running_var), _p(ws), nb, _stream()), "dvm_bn_act_train_fwd_f32")
    return y, mean, invstd


def bn_act_train_bwd(dy, y, x, res, gamma, mean, invstd, slope):
    """-> (dx (also the gradient of res), dgamma, dbeta)."""
    _need_gpu(dy, y, x)
    dy, y, x = _f(dy), _f(y), _f(x)
    res = None if res is None else _f(res)
    B, C, N = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = lib.dvm_bn_workspace_bytes(B, C, N)
    ws = workspace(nb, x.device, "bn")
    check(lib.dvm_bn_act_train_bwd_f32(_p(dy), _p(y), _p(x), None if res is None else _p(res), None if gamma is None else _p(_f(gamma)),
                                       _p(mean), _p(invstd), B, C, N, float(slope), _p(dx), _p(dgamma), _p(dbeta), _p(ws), nb,
                                       _stream()), "dvm_bn_act_train_bwd_f32")
    return dx, dgamma, dbeta


def graph_geodesics(nbr, w):
    """nbr (N,K) int32 neighbour lists (-1 padded), w (N,K) float64 edge lengths, on the device -> (N,N) float64."""
    _need_gpu(nbr, w)
    nbr, w = nbr.contiguous().int(), w.contiguous().double()
    N, K = nbr.shape
    D = torch.empty(N, N, dtype=torch.float64, device=nbr.device)
    check(_lib.load().dvm_graph_geodesics_f64(_p(nbr), _p(w), N, K, _p(D), _stream()), "dvm_graph_geodesics_f64")
    return D


def proj2img(pts):
    """One view's point cloud (B,N,3) -> (img (B,3,224,224), pc_min (B,2), grid_size (B,), offsets (B,2))."""
    _need_gpu(pts)
    pts = _f(pts)
    B, N, _ = pts.shape
    lib = _lib.load()
    dev = pts.device
    img = torch.empty(B, 3, 224, 224, dtype=torch.float32, device=dev)
    pc_min, grid, off = (torch.empty(B, 2, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev),
                         torch.empty(B, 2, dtype=torch.float32, device=dev))
    nb = lib.dvm_proj2img_workspace_bytes(B)
    ws = workspace(nb, dev, "proj2img")
    check(lib.dvm_proj2img_f32(_p(pts), B, N, _p(img), _p(pc_min), _p(grid), _p(off), _p(ws), nb, _stream()), "dvm_proj2img_f32")
    return img, pc_min, grid, off


def i2p(pts, f, pc_min, grid, offsets, normalize=False, out=None, col=0):
    """Image features f (B,C,H,W) sampled at every point's pixel: (B,N,C), or columns [col, col+C) of `out` (B,N,ldo)."""
    _need_gpu(pts, f, pc_min, grid, offsets)
    pts, f, pc_min, grid, offsets = _f(pts), _f(f), _f(pc_min), _f(grid), _f(offsets)
    B, N, _ = pts.shape
    C, H, W = f.shape[1:]
    if out is None:
        out, col = torch.empty(B, N, C, dtype=torch.float32, device=pts.device), 0
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[:2] == (B, N) and col + C <= out.shape[2]
    check(_lib.load().dvm_i2p_f32(_p(pts), _p(f), _p(pc_min), _p(grid), _p(offsets), B, N, C, H, W, int(bool(normalize)),
                                  ctypes.c_void_p(out.data_ptr() + 4 * col), out.shape[2], _stream()), "dvm_i2p_f32")
    return out


def adaptive_conv(x_padded, kernel, tap_major=False):
    """FeatUp's AdaptiveConv: x_padded (B,C,H+d-1,W+d-1), kernel (B,H,W,d,d) — or (B,d*d,H,W) with tap_major — -> (B,C,H,W)."""
    _need_gpu(x_padded, kernel)
    x_padded, kernel = _f(x_padded), _f(kernel)
    B, C, Hp, Wp = x_padded.shape
    if tap_major:
        _, d2, H, W = kernel.shape
        d = int(round(d2 ** 0.5))
        ok = d * d == d2
    else:
        _, H, W, d, dd = kernel.shape
        ok = d == dd
    if not ok or Hp != H + d - 1 or Wp != W + d - 1 or kernel.shape[0] != B:
        raise DvmError("adaptive_conv: shapes %s and %s do not match" % (tuple(x_padded.shape), tuple(kernel.shape)))
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x_padded.device)
    check(_lib.load().dvm_adaptive_conv_f32(_p(x_padded), _p(kernel), B, C, H, W, d, 1 if tap_major else 0, _p(out), _stream()),
          "dvm_adaptive_conv_f32")
    return out


def bicubic_resize_pad(x, size, pad=0):
    """F.pad(F.interpolate(x, size, mode='bicubic', align_corners=False), [pad]*4, mode='reflect') in one kernel:
    x (B,C,Hi,Wi) -> (B,C,Ho+2pad,Wo+2pad)."""
    _need_gpu(x)
    x = _f(x)
    B, C, Hi, Wi = x.shape
    Ho, Wo = size
    out = torch.empty(B, C, Ho + 2 * pad, Wo + 2 * pad, dtype=torch.float32, device=x.device)
    check(_lib.load().dvm_bicubic_resize_pad_f32(_p(x), B * C, Hi, Wi, Ho, Wo, pad, _p(out), _stream()), "dvm_bicubic_resize_pad_f32")
    return out


def jbu_kernel(proj, range_temp, sigma_spatial, d=7):
    """Normalised range x spatial kernel of a joint-bilateral stage: proj (B,32,H,W) -> (B,d*d,H,W) (dvm_jbu_kernel_f32)."""
    _need_gpu(proj, range_temp, sigma_spatial)
    proj = _f(proj)
    B, Kd, H, W = proj.shape
    out = torch.empty(B, d * d, H, W, dtype=torch.float32, device=proj.device)
    check(_lib.load().dvm_jbu_kernel_f32(_p(proj), _p(_f(range_temp)), _p(_f(sigma_spatial)), B, Kd, H, W, d, _p(out), _stream()),
          "dvm_jbu_kernel_f32")
    return out


def sa_attention_pm(p, v):
    """point-major p (B,N,16), v (B,N,64) -> x_r (B,N,64)."""
    _need_gpu(p, v)
    p, v = _f(p), _f(v)
    B, N, _ = p.shape
    lib = _lib.load()
    xr = torch.empty(B, N, 64, dtype=torch.float32, device=p.device)
    nb = lib.dvm_sa_attention_workspace_bytes(B, N)
    ws = workspace(nb, p.device, "sa")
    check(lib.dvm_sa_attention_fwd_f32(_p(p), _p(v), B, N, _p(xr), _p(ws), nb, _stream()), "dvm_sa_attention_fwd_f32")
    return xr


def sa_attention_train_fwd(p, v):
    """-> xr (B,N,64), stats (B,N,2), cinv (B,N)."""
    _need_gpu(p, v)
    p, v = _f(p), _f(v)
    B, N, _ = p.shape
    xr = torch.empty(B, N, 64, dtype=torch.float32, device=p.device)
    stats = torch.empty(B, N, 2, dtype=torch.float32, device=p.device)
    cinv = torch.empty(B, N, dtype=torch.float32, device=p.device)
    lib = _lib.load()
    nb = lib.dvm_sa_attention_train_fwd_workspace_bytes(B, N)
    ws = workspace(nb, p.device, "sa_train") if nb else None
    check(lib.dvm_sa_attention_train_fwd_f32(_p(p), _p(v), B, N, _p(xr), _p(stats), _p(cinv), _p(ws) if nb else None, nb, _stream()),
          "dvm_sa_attention_train_fwd_f32")
    return xr, stats, cinv


def sa_attention_bwd(p, v, xr, stats, cinv, gxr):
    """-> d_p (B,N,16), d_v (B,N,64)."""
    _need_gpu(p, v, gxr)
    p, v, gxr = _f(p), _f(v), _f(gxr)
    B, N, _ = p.shape
    dp, dv = torch.empty_like(p), torch.empty_like(v)
    lib = _lib.load()
    nb = lib.dvm_sa_attention_bwd_workspace_bytes(B, N)
    ws = workspace(nb, p.device, "sa_bwd")
    check(lib.dvm_sa_attention_bwd_f32(_p(p), _p(v), _p(xr), _p(stats), _p(cinv), _p(gxr), B, N, _p(dp), _p(dv), _p(ws), nb,
                                       _stream()), "dvm_sa_attention_bwd_f32")
    return dp, dv


def sa_attention(x, w_qk, w_v, b_v):
    """x (B,64,N) channel-major like the reference; returns x_r (B,64,N)."""
    xt = x.transpose(1, 2).contiguous()
    return sa_attention_pm(linear(xt, w_qk), linear(xt, w_v, bias=b_v)).transpose(1, 2)


def n2p_attention_pm(q, kp, vp, idx, heads=4):
    _need_gpu(q, kp, vp, idx)
    q, kp, vp, idx = _f(q), _f(kp), _f(vp), _i(idx)
    B, N, C = q.shape
    K = idx.shape[2]
    out = torch.empty(B, N, C, dtype=torch.float32, device=q.device)
    check(_lib.load().dvm_n2p_attention_fwd_f32(_p(q), _p(kp), _p(vp), _p(idx), B, N, C, K, heads, _p(out), _stream()),
          "dvm_n2p_attention_fwd_f32")
    return out


def n2p_attention(x, K, wq, wk, wv, heads=4):
    """x (B,C,N) channel-major like the reference; returns the attention output (B,C,N)."""
    C = x.shape[1]
    xt = x.transpose(1, 2).contiguous()
    idx = knn_neg(xt, xt, K)
    w = torch.cat([wq.reshape(C, C), wk.reshape(C, C), wv.reshape(C, C)], 0)
    qkv = linear(xt, w)
    out = n2p_attention_pm(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], idx, heads)
    return out.transpose(1, 2)


def dist_loss(feat, dist, anchors, k, want_idx=False):
    """feat (B,N,C), dist (B,N,N), anchors (nA,) -> (B,) sum over anchors of 1-|cos|."""
    _need_gpu(feat, dist, anchors)
    feat, dist, anchors = _f(feat), _f(dist), _i(anchors)
    B, N, C = feat.shape
    nA = anchors.shape[0]
    lib = _lib.load()
    out = torch.empty(B, dtype=torch.float32, device=feat.device)
    idx = torch.empty(B, nA, k, dtype=torch.int32, device=feat.device) if want_idx else None
    nb = lib.dvm_dist_loss_workspace_bytes(B, N, C, nA, k)
    ws = workspace(nb, feat.device, "dist")
    check(lib.dvm_dist_loss_fwd_f32(_p(feat), _p(dist), _p(anchors), B, N, C, nA, k, _p(out), _p(idx), _p(ws), nb,
                                    _stream()), "dvm_dist_loss_fwd_f32")
    return (out, idx) if want_idx else out


def dist_loss_bwd_weights(feat, dist, anchors, idx, gout):
    """-> W (B,nA,N), see include/dvm.h."""
    _need_gpu(feat, dist, anchors, idx, gout)
    feat, dist, anchors, idx, gout = _f(feat), _f(dist), _i(anchors), _i(idx), _f(gout)
    B, N, C = feat.shape
    nA, k = idx.shape[1], idx.shape[2]
    W = torch.empty(B, nA, N, dtype=torch.float32, device=feat.device)
    check(_lib.load().dvm_dist_loss_bwd_weights_f32(_p(feat), _p(dist), _p(anchors), _p(idx), _p(gout), B, N, C, nA, k, _p(W),
                                                    _stream()), "dvm_dist_loss_bwd_weights_f32")
    return W


U3_NWEIGHTS = 101


def uni3fc_weight_table(tensors):
    """The host array of DVM_U3_NWEIGHTS device pointers dvm_uni3fc_fwd_f32 takes (order: include/dvm.h) from a list of
    fp32 CUDA tensors; returns (kept-alive tensors, ctypes array)."""
    import ctypes
    ts = [_f(t) for t in tensors]
    if len(ts) != U3_NWEIGHTS:
        raise DvmError("uni3fc_weight_table: %d tensors, expected %d" % (len(ts), U3_NWEIGHTS))
    _need_gpu(*ts)
    return ts, (ctypes.c_void_p * U3_NWEIGHTS)(*[t.data_ptr() for t in ts])


def uni3fc_forward(table, x, dino, k=40):
    """LG-Net's eval-mode forward in ONE native call (dvm_uni3fc_fwd_f32): x (B,3,N), dino (B,N,1152) -> feat (B,N,128),
    tmp (B,N,64).  `table` = uni3fc_weight_table(...)."""
    import ctypes
    _need_gpu(x, dino)
    x, dino = _f(x), _f(dino)
    B, _, N = x.shape
    if tuple(dino.shape) != (B, N, 1152):
        raise DvmError("uni3fc_forward: dino features %s, expected %s" % (tuple(dino.shape), (B, N, 1152)))
    lib = _lib.load()
    dev = x.device
    feat = torch.empty(B, N, 128, dtype=torch.float32, device=dev)
    tmp = torch.empty(B, N, 64, dtype=torch.float32, device=dev)
    nb = lib.dvm_uni3fc_fwd_workspace_bytes(B, N, int(k))
    ws = workspace(nb, dev, "uni3fc")
    ctx = (dev.index, _stream())
    if ctx not in _pair_ctx:   # helper stream / events for this (device, stream): made once, outside the compute call
        check(lib.dvm_pair_init(_stream()), "dvm_pair_init")
        _pair_ctx.add(ctx)
    check(lib.dvm_uni3fc_fwd_f32(_p(x), _p(dino), B, N, ctypes.cast(table[1], ctypes.c_void_p), U3_NWEIGHTS, int(k), _p(feat), _p(tmp),
                                 _p(ws), nb, _stream()), "dvm_uni3fc_fwd_f32")
    return feat, tmp


U3_TRAIN_NPARAMS = 167   # DVM_U3_TRAIN_NPARAMS (include/dvm.h)


def _ptr_table(tensors, n):
    if len(tensors) != n:
        raise DvmError("pointer table: %d tensors, expected %d" % (len(tensors), n))
    return (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in tensors])


def _ensure_pair_ctx(dev):
    ctx = (dev.index, _stream())
    if ctx not in _pair_ctx:   # helper stream / events for this (device, stream): made once, outside the compute call
        check(_lib.load().dvm_pair_init(_stream()), "dvm_pair_init")
        _pair_ctx.add(ctx)


def knn_tap():
    """Test handle on the feature-space kNN layers.  tests/test_gpu_network.py replaces `ops.knn_neg` by a callable object with
    the attributes `forced` (None, or per-layer index arrays (B,N,k) to use INSTEAD of our own sets: teacher forcing with the
    reference's neighbours), `log` (list that receives our own sets, layer by layer) and `i` (cursor into `forced`).  The
    layer-by-layer Python path goes through that object call by call; a native whole-network call has no Python between its
    layers, so it hands the same object's `forced` entries to the library and appends the library's own sets to `log`
    (dvm_uni3fc_train_fwd_f32's knn_forced / knn_log).  Returns the object, or None when knn_neg is the plain function."""
    import inspect
    fn = globals()["knn_neg"]
    return None if inspect.isfunction(fn) else (fn if hasattr(fn, "log") and hasattr(fn, "forced") else None)


def _coll_check(coll, rc, what):
    if coll is not None and coll.error is not None:
        err, coll.error = coll.error, None
        raise DvmError("%s: the collective failed: %r" % (what, err))
    check(rc, what)


def uni3fc_train_forward(params, x, dino, k, eps, momentum, defer_stats=False, groups=1, coll=None):
    """LG-Net's training-mode forward in ONE native call (dvm_uni3fc_train_fwd_f32).  params: the U3_TRAIN_NPARAMS tensors of
    include/dvm.h's table (raw parameters + BatchNorm running statistics, updated in place); x (B,3,N), dino (B,N,1152)
    -> feat (B,N,128), tmp (B,N,64), arena (uint8 tensor holding what dvm_uni3fc_train_bwd_f32 needs)."""
    _need_gpu(x, dino)
    x, dino = _f(x), _f(dino)
    B, _, N = x.shape
    if tuple(dino.shape) != (B, N, 1152):
        raise DvmError("uni3fc_train_forward: dino features %s, expected %s" % (tuple(dino.shape), (B, N, 1152)))
    lib = _lib.load()
    dev = x.device
    feat = torch.empty(B, N, 128, dtype=torch.float32, device=dev)
    tmp = torch.empty(B, N, 64, dtype=torch.float32, device=dev)
    nb = lib.dvm_uni3fc_train_workspace_bytes(B, N, int(k))
    arena = torch.empty(nb, dtype=torch.uint8, device=dev)
    _ensure_pair_ctx(dev)
    table = _ptr_table(params, U3_TRAIN_NPARAMS)
    tap = knn_tap()
    forced = logs = ftab = ltab = None
    if tap is not None:
        logs = [torch.empty(B, N, int(k), dtype=torch.int32, device=dev) for _ in range(7)]
        ltab = ctypes.cast(_ptr_table(logs, 7), ctypes.c_void_p)
        if tap.forced is not None:
            import numpy as np
            forced = [torch.from_numpy(np.ascontiguousarray(tap.forced[tap.i + l]).astype(np.int32)).to(dev) for l in range(7)]
            if any(tuple(f.shape) != (B, N, int(k)) for f in forced):
                raise DvmError("uni3fc_train_forward: forced neighbour sets must be (B, N, k)")
            ftab = ctypes.cast(_ptr_table(forced, 7), ctypes.c_void_p)
    # coll (dvm.dist.TorchCollective): BatchNorm statistics and the position-encoding range over ALL ranks of a data-parallel step
    _coll_check(coll, lib.dvm_uni3fc_train_fwd_sync_f32(_p(x), _p(dino), B, N, ctypes.cast(table, ctypes.c_void_p), U3_TRAIN_NPARAMS, int(k), float(eps),
                                                        float(momentum), int(groups), 1 if defer_stats else 0, ftab, ltab, _p(feat), _p(tmp), _p(arena), nb,
                                                        coll.bind(arena) if coll is not None else None, _stream()), "dvm_uni3fc_train_fwd_f32")
    if tap is not None:
        tap.log.extend(logs)
        if forced is not None:
            tap.i += 7
    return feat, tmp, arena


CRIT_TRAIN_NPARAMS = 10


def _graph_c(g):
    """The batched graph dict as the C ABI wants it: int32 / fp32, contiguous (the same tensors when they already are — dg_build's
    always are; a caller-supplied `geometry=` with int64 or strided indices is converted instead of being reinterpreted)."""
    return {"nodes_idx": _i(g["nodes_idx"]), "one_ring": _i(g["one_ring"]), "infl_idx": _i(g["infl_idx"]), "weights": _f(g["weights"])}


def criterion_train_forward(params, feat, verts, g, knn_idx, alpha, topk=10, with_map=True, dist=None):
    """The training criterion for B pairs as ONE native call (dvm_criterion_train_fwd_f32).  feat (2B,N,128), verts (2B,N,3): the B first
    shapes followed by the B second shapes; g = their batched graph dict (dg_build), knn_idx (2B,N,k) their xyz-kNN; params: the
    Deformer's 10 tensors (conv weight, bias, then the decoder's weight / bias pairs); dist = None or (dist1 (B,N,N), dist2 (B,N,N),
    anchors1 (nA,) int32, anchors2 (nA,) int32, k_dist): the dist term of all 2B shapes.
    -> terms (2B,7) [map numerator, Chamfer side means of warped (2) and verts12 (2), ARAP, dist term], arena (uint8 tensor for the backward)."""
    _need_gpu(feat, verts, knn_idx, *[g[k_] for k_ in ("nodes_idx", "one_ring", "infl_idx", "weights")])
    feat, verts, knn_idx, g = _f(feat), _f(verts), _i(knn_idx), _graph_c(g)
    P, N, C = feat.shape
    k = knn_idx.shape[-1]
    lib = _lib.load()
    terms = torch.empty(P, 7, dtype=torch.float32, device=feat.device)
    d1, d2, a1, a2, kd = dist if dist is not None else (None, None, None, None, 0)
    nA = 0 if dist is None else int(a1.numel())
    if dist is not None and not (d1.dtype is torch.float32 and d1.is_contiguous() and d2.dtype is torch.float32 and d2.is_contiguous()
                                 and a1.dtype is torch.int32 and a2.dtype is torch.int32 and tuple(d1.shape) == (P // 2, N, N)
                                 and tuple(d2.shape) == (P // 2, N, N) and a2.numel() == nA):
        raise DvmError("criterion_train_forward: dist term inputs must be contiguous fp32 (B,N,N) matrices and int32 anchor lists of one length")
    nb = lib.dvm_criterion_train_workspace_bytes(P // 2, N, k, topk, nA, int(kd))
    arena = torch.empty(nb, dtype=torch.uint8, device=feat.device)
    if nA:
        _ensure_pair_ctx(feat.device)
    table = _ptr_table(params, CRIT_TRAIN_NPARAMS)
    check(lib.dvm_criterion_train_fwd_f32(_p(feat), _p(verts), _p(g["nodes_idx"]), _p(g["one_ring"]), _p(g["infl_idx"]), _p(g["weights"]), _p(knn_idx),
                                          P // 2, N, C, k, topk, neg_alpha_f32(alpha), ctypes.cast(table, ctypes.c_void_p), CRIT_TRAIN_NPARAMS,
                                          1 if with_map else 0, _p(d1), _p(d2), _p(a1), _p(a2), nA, int(kd), _p(terms), _p(arena), nb, _stream()),
          "dvm_criterion_train_fwd_f32")
    return terms, arena


def criterion_train_backward(params, grads, g_terms, feat, verts, g, knn_idx, alpha, arena, topk=10, with_map=True, dist=None):
    """dvm_criterion_train_bwd_f32: -> d_feat (2B,N,128); the Deformer's parameter gradients are ADDED into `grads`."""
    _need_gpu(feat, g_terms, knn_idx)
    feat, verts, knn_idx, g = _f(feat), _f(verts), _i(knn_idx), _graph_c(g)
    P, N, C = feat.shape
    k = knn_idx.shape[-1]
    lib = _lib.load()
    d_feat = torch.empty_like(feat)
    _, _, a1, a2, kd = dist if dist is not None else (None, None, None, None, 0)
    nA = 0 if dist is None else int(a1.numel())
    if nA:
        _ensure_pair_ctx(feat.device)
    ptab, gtab = _ptr_table(params, CRIT_TRAIN_NPARAMS), _ptr_table(grads, CRIT_TRAIN_NPARAMS)
    check(lib.dvm_criterion_train_bwd_f32(_p(_f(g_terms)), _p(feat), _p(verts), _p(g["nodes_idx"]), _p(g["one_ring"]), _p(g["infl_idx"]), _p(g["weights"]),
                                          _p(knn_idx), P // 2, N, C, k, topk, neg_alpha_f32(alpha), ctypes.cast(ptab, ctypes.c_void_p),
                                          ctypes.cast(gtab, ctypes.c_void_p), CRIT_TRAIN_NPARAMS, 1 if with_map else 0, _p(a1), _p(a2), nA, int(kd),
                                          _p(d_feat), _p(arena), arena.numel(), _stream()), "dvm_criterion_train_bwd_f32")
    return d_feat


def criterion_dir_train_forward(params, feat_s, feat_t, verts_s, verts_t, g, knn_s, knn_t, alpha, topk=10, with_map=False):
    """ONE direction of the training criterion's deformation part for P pairs with N source and M target points
    (dvm_criterion_dir_train_fwd_f32): g = the SOURCES' graph dict, knn_s (P,N,k) / knn_t (P,M,k) the xyz-kNN of both sides.
    -> terms (P,7), arena."""
    _need_gpu(feat_s, feat_t, verts_s, verts_t, knn_s, knn_t)
    feat_s, feat_t, verts_s, verts_t = _f(feat_s), _f(feat_t), _f(verts_s), _f(verts_t)
    knn_s, knn_t, g = _i(knn_s), _i(knn_t), _graph_c(g)
    P, N, C = feat_s.shape
    M, k = feat_t.shape[1], knn_s.shape[-1]
    lib = _lib.load()
    terms = torch.empty(P, 7, dtype=torch.float32, device=feat_s.device)
    nb = lib.dvm_criterion_dir_train_workspace_bytes(P, N, M, k, topk)
    arena = torch.empty(nb, dtype=torch.uint8, device=feat_s.device)
    table = _ptr_table(params, CRIT_TRAIN_NPARAMS)
    check(lib.dvm_criterion_dir_train_fwd_f32(_p(feat_s), _p(feat_t), _p(verts_s), _p(verts_t), _p(g["nodes_idx"]), _p(g["one_ring"]), _p(g["infl_idx"]),
                                              _p(g["weights"]), _p(knn_s), _p(knn_t), P, N, M, C, k, topk, neg_alpha_f32(alpha),
                                              ctypes.cast(table, ctypes.c_void_p), CRIT_TRAIN_NPARAMS, 1 if with_map else 0, _p(terms), _p(arena), nb,
                                              _stream()), "dvm_criterion_dir_train_fwd_f32")
    return terms, arena


def criterion_dir_train_backward(params, grads, g_terms, feat_s, feat_t, verts_s, verts_t, g, knn_s, knn_t, alpha, arena, topk=10, with_map=False):
    """dvm_criterion_dir_train_bwd_f32: -> (d_feat_s (P,N,128), d_feat_t (P,M,128)); parameter gradients ADDED into `grads`."""
    _need_gpu(feat_s, g_terms, knn_s, knn_t)
    feat_s, feat_t, verts_s, verts_t = _f(feat_s), _f(feat_t), _f(verts_s), _f(verts_t)
    knn_s, knn_t, g = _i(knn_s), _i(knn_t), _graph_c(g)
    P, N, C = feat_s.shape
    M, k = feat_t.shape[1], knn_s.shape[-1]
    lib = _lib.load()
    d_s, d_t = torch.empty_like(feat_s), torch.empty_like(feat_t)
    ptab, gtab = _ptr_table(params, CRIT_TRAIN_NPARAMS), _ptr_table(grads, CRIT_TRAIN_NPARAMS)
    check(lib.dvm_criterion_dir_train_bwd_f32(_p(_f(g_terms)), _p(feat_s), _p(feat_t), _p(verts_s), _p(verts_t), _p(g["nodes_idx"]), _p(g["one_ring"]),
                                              _p(g["infl_idx"]), _p(g["weights"]), _p(knn_s), _p(knn_t), P, N, M, C, k, topk, neg_alpha_f32(alpha),
                                              ctypes.cast(ptab, ctypes.c_void_p), ctypes.cast(gtab, ctypes.c_void_p), CRIT_TRAIN_NPARAMS,
                                              1 if with_map else 0, _p(d_s), _p(d_t), _p(arena), arena.numel(), _stream()),
          "dvm_criterion_dir_train_bwd_f32")
    return d_s, d_t


def uni3fc_train_running_stats(params, arena, B, N, k, momentum, groups=1):
    """The 26 running-statistics updates of a forward that ran with defer_stats=True (dvm_uni3fc_train_running_stats_f32), on the
    current stream."""
    lib = _lib.load()
    table = _ptr_table(params, U3_TRAIN_NPARAMS)
    check(lib.dvm_uni3fc_train_running_stats_f32(ctypes.cast(table, ctypes.c_void_p), U3_TRAIN_NPARAMS, int(B), int(N), int(k), int(groups), float(momentum),
                                                 _p(arena), arena.numel(), _stream()), "dvm_uni3fc_train_running_stats_f32")


def uni3fc_train_backward(params, grads, dino, feat, tmp, arena, g_feat, g_tmp, k, groups=1, coll=None):
    """dvm_uni3fc_train_bwd_f32: ADDS the parameter gradients into `grads` (tensors aligned with `params`; None for the
    running statistics)."""
    _need_gpu(g_feat, dino)
    B, N, _ = feat.shape
    lib = _lib.load()
    _ensure_pair_ctx(feat.device)
    g_feat = _f(g_feat)
    g_tmp = None if g_tmp is None else _f(g_tmp)
    ptab, gtab = _ptr_table(params, U3_TRAIN_NPARAMS), _ptr_table(grads, U3_TRAIN_NPARAMS)
    _coll_check(coll, lib.dvm_uni3fc_train_bwd_sync_f32(_p(g_feat), _p(g_tmp), _p(dino), _p(feat), _p(tmp), B, N, ctypes.cast(ptab, ctypes.c_void_p),
                                                        ctypes.cast(gtab, ctypes.c_void_p), U3_TRAIN_NPARAMS, int(k), int(groups), _p(arena), arena.numel(),
                                                        coll.bind(arena) if coll is not None else None, _stream()), "dvm_uni3fc_train_bwd_f32")


class GeometryCache:
    """Opt-in per-shape graph cache of the pair forward (the reference rebuilds both deformation graphs on every
    call, models/loss.py:1325-1337).  An entry is a DEDICATED workspace in which a dvm_pair_fwd_cached_f32 call left the
    coordinate-only products of a batch (graphs, grids, xyz kNN); it is keyed by whatever identifies the batch's geometry —
    the caller's shape ids AND FPS start indices (a different start gives a different graph) — plus the sizes.  A hit skips the
    geometry chain; outputs are bit-identical.  Least-recently-used entries are dropped beyond `max_entries`."""

    def __init__(self, max_entries=16):
        import collections
        self.max_entries, self.entries, self.hits, self.misses = max_entries, collections.OrderedDict(), 0, 0

    def lookup(self, key, nbytes, device):
        """-> (workspace, reuse flag)"""
        ent = self.entries.get(key)
        if ent is not None and ent.numel() >= nbytes and ent.device == device:
            self.entries.move_to_end(key)
            self.hits += 1
            return ent, 1
        ent = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self.entries[key] = ent
        self.misses += 1
        while len(self.entries) > self.max_entries:
            self.entries.popitem(last=False)
        return ent, 0

    def clear(self):
        self.entries.clear()


def pair_forward(wl, feat1, feat2, verts1, verts2, alpha, start1, start2, with_map=True, out=None, cache=None, key=None):
    """Config-2 path for B pairs, BOTH directions in one call.
    Returns (out12, out21), each dict(warped, verts12, T12, losses[B,6]).
    cache / key: a GeometryCache and the hashable identity of this batch's geometry (shape ids + FPS starts): on a hit the
    graphs / grids / xyz kNN of the cached call are reused (same bits, no geometry chain)."""
    _need_gpu(feat1, feat2, verts1, verts2, start1, start2, *wl)
    feat1, feat2, verts1, verts2 = _f(feat1), _f(feat2), _f(verts1), _f(verts2)
    start1, start2 = _i(start1), _i(start2)
    B, N, _ = feat1.shape
    M = feat2.shape[1]
    dev = feat1.device
    lib = _lib.load()

    def alloc(n):
        return dict(warped=torch.empty(B, n, 3, dtype=torch.float32, device=dev),
                    verts12=torch.empty(B, n, 3, dtype=torch.float32, device=dev),
                    T12=torch.empty(B, n, dtype=torch.int32, device=dev),
                    losses=torch.empty(B, 6, dtype=torch.float32, device=dev))
    o12, o21 = out if out is not None else (alloc(N), alloc(M))
    nb = lib.dvm_pair_workspace_bytes(B, N, M)
    ctx = (dev.index, _stream())
    if ctx not in _pair_ctx:   # helper streams / events for this (device, stream): made once, outside the compute call
        check(lib.dvm_pair_init(_stream()), "dvm_pair_init")
        _pair_ctx.add(ctx)
    if cache is not None:
        if key is None:
            raise DvmError("pair_forward: a GeometryCache needs the key of this batch's geometry (shape ids + FPS starts)")
        ws, reuse = cache.lookup((key, B, N, M, bool(with_map), ctx), nb, dev)
        check(lib.dvm_pair_fwd_cached_f32(_p(feat1), _p(feat2), _p(verts1), _p(verts2), B, N, M, neg_alpha_f32(alpha), _p(start1),
                                          _p(start2), *[_p(w) for w in wl], int(with_map), _p(o12["warped"]), _p(o12["verts12"]),
                                          _p(o12["T12"]), _p(o12["losses"]), _p(o21["warped"]), _p(o21["verts12"]), _p(o21["T12"]),
                                          _p(o21["losses"]), _p(ws), nb, reuse, _stream()), "dvm_pair_fwd_cached_f32")
        return o12, o21
    ws = workspace(nb, dev, "pair2")
    check(lib.dvm_pair_fwd_f32(_p(feat1), _p(feat2), _p(verts1), _p(verts2), B, N, M, neg_alpha_f32(alpha), _p(start1),
                               _p(start2), *[_p(w) for w in wl], int(with_map), _p(o12["warped"]), _p(o12["verts12"]),
                               _p(o12["T12"]), _p(o12["losses"]), _p(o21["warped"]), _p(o21["verts12"]), _p(o21["T12"]),
                               _p(o21["losses"]), _p(ws), nb, _stream()), "dvm_pair_fwd_f32")
    return o12, o21


def _cu_count():
    """Compute units of the current device."""
    return torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count


class PairPipeline:
    """The pair forward as a two-stage software pipeline over a stream of batches: stage 1 = the coordinate-only geometry of a
    batch (dvm_pair_geometry_f32: FPS -> node grid -> ring -> influence -> skinning, vertex grid, xyz kNN) on a stream of its own,
    stage 2 = everything that needs the features (dvm_pair_fwd_cached_f32 with reuse_geometry = 1) on torch's current stream.
    While stage 2 of batch t runs, stage 1 of batch t + 1 does: the N / 2 dependent FPS steps (0.65 us each, a latency chain that
    no batch size hides) leave the critical path.  NOTHING is cached — every batch's graphs are built exactly once, as in the
    reference (models/loss.py:1325-1337), only one stage earlier; outputs are bit-identical to pair_forward.

        pipe = PairPipeline(wl, B, N, M)
        tk = pipe.prefetch(v1, v2, s1, s2, ready=ev)          # batch 0
        for each batch:  (o12, o21), tk = pipe.step(tk, f1, f2, alpha, next_coords=(v1', v2', s1', s2'), ready=ev')

    step() enqueues the two stages in the order (`schedule`) that measured faster for the batch size (bench.py --pairs 32 ... 512,
    profiles/r6_pipeline_order.txt): "geometry first" while its FPS workgroups (one per cloud) fit the compute units once — the
    dependent FPS chain is then the long pole and must start at once —, "features first" beyond (the chip is throughput-bound and
    the sweep, enqueued first, is disturbed less).  `depth` workspaces (default 2) rotate; a ticket must be consumed by forward()
    before `depth` further prefetches."""

    def __init__(self, wl, B, N, M, with_map=True, depth=2, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.wl, self.B, self.N, self.M, self.with_map, self.dev = wl, int(B), int(N), int(M), bool(with_map), dev
        lib = _lib.load()
        self.nb = lib.dvm_pair_workspace_bytes(self.B, self.N, self.M)
        self.ws = [torch.empty(max(int(self.nb), 256), dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.free = [None] * depth           # event: the stage-2 call that last used the workspace has finished
        self.geo = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self.geo):
            check(lib.dvm_pair_init(_stream()), "dvm_pair_init")      # the geometry stream's own helper stream / events
        self.n = 0
        self.slot_gen = [0] * depth          # which prefetch last wrote the workspace (a ticket is valid while it is the one)
        self.schedule = "geometry first" if 2 * self.B <= _cu_count() else "features first"

    def close(self):
        """The current stream waits for whatever the geometry stream still has in flight (a prefetch that no forward() consumed writes
        into a workspace this object owns: its memory must not return to the allocator before that)."""
        if getattr(self, "geo", None) is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.geo)
            self.geo = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def prefetch(self, verts1, verts2, start1, start2, ready=None):
        """Enqueue stage 1 for a batch; -> ticket for forward().  ready: a torch.cuda.Event behind which the coordinates and starts are
        valid (e.g. recorded when the batch was loaded); None = behind everything enqueued on the current stream so far — which, called
        after a forward(), is the END of that forward: no overlap with it (pass `ready` to pipeline)."""
        _need_gpu(verts1, verts2, start1, start2)
        verts1, verts2, start1, start2 = _f(verts1), _f(verts2), _i(start1), _i(start2)
        if tuple(verts1.shape) != (self.B, self.N, 3) or tuple(verts2.shape) != (self.B, self.M, 3):
            raise DvmError("PairPipeline.prefetch: coordinates %s / %s, built for B=%d N=%d M=%d"
                           % (tuple(verts1.shape), tuple(verts2.shape), self.B, self.N, self.M))
        if self.geo is None:
            raise DvmError("PairPipeline.prefetch: the pipeline was closed")
        slot = self.n % len(self.ws)
        self.n += 1
        self.slot_gen[slot] = self.n
        cur = torch.cuda.current_stream(self.dev)
        if ready is None:
            self.geo.wait_stream(cur)                   # the coordinates were produced on the caller's stream
        else:
            self.geo.wait_event(ready)
        if self.free[slot] is not None:
            self.geo.wait_event(self.free[slot])        # the workspace's previous consumer is done
        with torch.cuda.stream(self.geo):
            check(_lib.load().dvm_pair_geometry_f32(_p(verts1), _p(verts2), self.B, self.N, self.M, _p(start1), _p(start2),
                                                    int(self.with_map), _p(self.ws[slot]), self.nb, _stream()), "dvm_pair_geometry_f32")
            ready = torch.cuda.Event()
            ready.record()
        return dict(slot=slot, gen=self.n, ready=ready, verts1=verts1, verts2=verts2, start1=start1, start2=start2)

    def forward(self, ticket, feat1, feat2, alpha, out=None):
        """Stage 2 of the ticket's batch on the current stream -> (out12, out21) as pair_forward."""
        _need_gpu(feat1, feat2, *self.wl)
        feat1, feat2 = _f(feat1), _f(feat2)
        B, N, M, dev = self.B, self.N, self.M, self.dev
        if tuple(feat1.shape) != (B, N, 128) or tuple(feat2.shape) != (B, M, 128):
            raise DvmError("PairPipeline.forward: features %s / %s, built for B=%d N=%d M=%d" % (tuple(feat1.shape), tuple(feat2.shape), B, N, M))
        lib = _lib.load()

        def alloc(n):
            return dict(warped=torch.empty(B, n, 3, dtype=torch.float32, device=dev),
                        verts12=torch.empty(B, n, 3, dtype=torch.float32, device=dev),
                        T12=torch.empty(B, n, dtype=torch.int32, device=dev),
                        losses=torch.empty(B, 6, dtype=torch.float32, device=dev))
        slot = ticket["slot"]
        if self.slot_gen[slot] != ticket["gen"]:
            raise DvmError("PairPipeline.forward: this ticket's workspace has been rewritten by a later prefetch (%d workspaces rotate: "
                           "consume a ticket before %d further prefetches)" % (len(self.ws), len(self.ws)))
        o12, o21 = out if out is not None else (alloc(N), alloc(M))
        _ensure_pair_ctx(dev)
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(ticket["ready"])
        check(lib.dvm_pair_fwd_cached_f32(_p(feat1), _p(feat2), _p(ticket["verts1"]), _p(ticket["verts2"]), B, N, M, neg_alpha_f32(alpha),
                                          _p(ticket["start1"]), _p(ticket["start2"]), *[_p(w) for w in self.wl], int(self.with_map),
                                          _p(o12["warped"]), _p(o12["verts12"]), _p(o12["T12"]), _p(o12["losses"]), _p(o21["warped"]),
                                          _p(o21["verts12"]), _p(o21["T12"]), _p(o21["losses"]), _p(self.ws[slot]), self.nb, 1, _stream()),
              "dvm_pair_fwd_cached_f32")
        ev = torch.cuda.Event()
        ev.record(cur)
        self.free[slot] = ev
        return o12, o21

    def step(self, ticket, feat1, feat2, alpha, next_coords=None, ready=None, out=None):
        """forward(ticket, ...) and prefetch(*next_coords, ready=ready) in the order that suits the batch size
        -> ((out12, out21), next ticket or None)."""
        if next_coords is None:
            return self.forward(ticket, feat1, feat2, alpha, out=out), None
        if self.schedule == "geometry first":
            nxt = self.prefetch(*next_coords, ready=ready)
            return self.forward(ticket, feat1, feat2, alpha, out=out), nxt
        outs = self.forward(ticket, feat1, feat2, alpha, out=out)
        return outs, self.prefetch(*next_coords, ready=ready)
