"""ctypes/numpy front-end of the CPU oracle (oracle/dvm_oracle.c).

TEST INFRASTRUCTURE ONLY: the parity checker for the HIP path and the
`cpu_baseline` ("port") leg of bench.py.  The product package never imports
this module.  Parity status: pinned against the reference's own outputs
(tests/golden/*.npz), chamfer excepted (third-party op, parity unpinned).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdvm_oracle.so")
_lib = None

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "dvm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.dvo_aten_sum.restype = ctypes.c_float
        _lib.dvo_map_term.restype = ctypes.c_float
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.c_void_p)


def neg_alpha_f32(alpha):
    """`-alpha * distance` in the reference multiplies an fp32 tensor by a python/numpy
    double: ATen casts the scalar to fp32 first (models/loss.py:112)."""
    return np.float32(-float(alpha))


def rownorm2(x):
    x = _f(x)
    out = np.empty(x.shape[0], np.float32)
    lib().dvo_rownorm2(_p(x), x.shape[0], x.shape[1], _p(out))
    return out


def cdist(a, b, exact=False):
    a, b = _f(a), _f(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().dvo_cdist(_p(a), _p(b), a.shape[0], b.shape[0], a.shape[1], int(exact), _p(out))
    return out


def argmin_exact(f1, f2):
    f1, f2 = _f(f1), _f(f2)
    T = np.empty(f1.shape[0], np.int32)
    dm = np.empty(f1.shape[0], np.float32)
    lib().dvo_argmin_exact(_p(f1), _p(f2), f1.shape[0], f2.shape[0], f1.shape[1], _p(T), _p(dm))
    return T, dm


def softcorr(f1, f2, alpha, topk=10):
    f1, f2 = _f(f1), _f(f2)
    N, M = f1.shape[0], f2.shape[0]
    val = np.empty((N, topk), np.float32)
    idx = np.empty((N, topk), np.int32)
    smax = np.empty(N, np.float32)
    ssum = np.empty(N, np.float32)
    lib().dvo_softcorr(_p(f1), _p(f2), N, M, f1.shape[1], ctypes.c_float(neg_alpha_f32(alpha)), topk, _p(val), _p(idx),
                       _p(smax), _p(ssum))
    return val, idx, smax, ssum


def knn_cdist(x, y, k):
    x, y = _f(x), _f(y)
    idx = np.empty((x.shape[0], k), np.int32)
    lib().dvo_knn_cdist(_p(x), _p(y), x.shape[0], y.shape[0], x.shape[1], k, _p(idx))
    return idx


def knn_neg(a, b, k):
    a, b = _f(a), _f(b)
    idx = np.empty((a.shape[0], k), np.int32)
    lib().dvo_knn_neg(_p(a), _p(b), a.shape[0], b.shape[0], a.shape[1], k, _p(idx))
    return idx


def apply(val, idx, V):
    val, idx, V = _f(val), _i(idx), _f(V)
    out = np.empty((val.shape[0], V.shape[1]), np.float32)
    lib().dvo_apply(_p(val), _p(idx), _p(V), val.shape[0], val.shape[1], V.shape[1], _p(out))
    return out


def fps(xyz, npoint, start):
    xyz = _f(xyz)
    out = np.empty(npoint, np.int32)
    lib().dvo_fps(_p(xyz), xyz.shape[0], npoint, int(start), _p(out))
    return out


def dg_build(xyz, start):
    xyz = _f(xyz)
    N = xyz.shape[0]
    Nn = N // 2
    nodes = np.empty(Nn, np.int32)
    ring = np.empty((Nn, 9), np.int32)
    infl = np.empty((N, 3), np.int32)
    dists = np.empty((N, 3), np.float32)
    w = np.empty((N, 3), np.float32)
    sigma = ctypes.c_double()
    lib().dvo_dg_build(_p(xyz), N, int(start), _p(nodes), _p(ring), _p(infl), _p(dists), _p(w), ctypes.byref(sigma))
    return dict(nodes_idx=nodes, one_ring=ring, infl_idx=infl, dists=dists, weights=w, sigma=sigma.value)


def rot6d(def9):
    def9 = _f(def9)
    Nn = def9.shape[0]
    R = np.empty((Nn, 3, 3), np.float32)
    T = np.empty((Nn, 3), np.float32)
    lib().dvo_rot6d(_p(def9), Nn, _p(R), _p(T))
    return R, T


def dg_warp_arap(xyz, g, R, T):
    xyz, R, T = _f(xyz), _f(R), _f(T)
    N = xyz.shape[0]
    warped = np.empty((N, 3), np.float32)
    arap, sr = ctypes.c_float(), ctypes.c_float()
    lib().dvo_dg_warp_arap(_p(xyz), N, _p(_i(g["nodes_idx"])), _p(_i(g["one_ring"])), _p(_i(g["infl_idx"])),
                           _p(_f(g["weights"])), _p(R), _p(T), _p(warped), ctypes.byref(arap), ctypes.byref(sr))
    return warped, arap.value, sr.value


def chamfer(a, b):
    a, b = _f(a), _f(b)
    N, M = a.shape[0], b.shape[0]
    d1, d2 = np.empty(N, np.float32), np.empty(M, np.float32)
    i1, i2 = np.empty(N, np.int32), np.empty(M, np.int32)
    lib().dvo_chamfer(_p(a), _p(b), N, M, _p(d1), _p(d2), _p(i1), _p(i2))
    return d1, d2, i1, i2


def _mlp_args(w):
    """w: dict with the reference state_dict keys (dots or double underscores)."""
    g = lambda k: _f(w[k] if k in w else w[k.replace(".", "__")])  # noqa: E731
    cw = g("conv_layer.weight").reshape(-1)
    cb = float(g("conv_layer.bias").reshape(-1)[0])
    mats = []
    for li in (0, 2, 4, 6):
        mats.append(g("deformation_decoder_layer.linear.%d.weight" % li))
        mats.append(g("deformation_decoder_layer.linear.%d.bias" % li))
    return cw, cb, mats


def deformer(weights, feat1, feat2, verts1, verts12, idx11, idx22, pval, pidx, fps1):
    cw, cb, mats = _mlp_args(weights)
    feat1, feat2, verts1, verts12 = _f(feat1), _f(feat2), _f(verts1), _f(verts12)
    idx11, idx22, pval, pidx, fps1 = _i(idx11), _i(idx22), _f(pval), _i(pidx), _i(fps1)
    N, M, Nn = feat1.shape[0], feat2.shape[0], fps1.shape[0]
    out = np.empty((Nn, 9), np.float32)
    lib().dvo_deformer(_p(feat1), _p(feat2), _p(verts1), _p(verts12), _p(idx11), _p(idx22), _p(pval), _p(pidx), _p(fps1),
                       N, M, Nn, idx11.shape[1], pval.shape[1], _p(cw), ctypes.c_float(cb), *[_p(m) for m in mats],
                       _p(out))
    return out


def map_term(verts12, verts2, idx11, idx22, pval, pidx):
    verts12, verts2, idx11, idx22, pval, pidx = _f(verts12), _f(verts2), _i(idx11), _i(idx22), _f(pval), _i(pidx)
    return float(lib().dvo_map_term(_p(verts12), _p(verts2), _p(idx11), _p(idx22), _p(pval), _p(pidx), verts12.shape[0],
                                    idx11.shape[1], pval.shape[1]))


def pair_direction(weights, feat1, feat2, verts1, verts2, alpha, fps_start, with_map=True):
    cw, cb, mats = _mlp_args(weights)
    feat1, feat2, verts1, verts2 = _f(feat1), _f(feat2), _f(verts1), _f(verts2)
    N, M = feat1.shape[0], feat2.shape[0]
    warped = np.empty((N, 3), np.float32)
    verts12 = np.empty((N, 3), np.float32)
    T12 = np.empty(N, np.int32)
    losses = np.empty(6, np.float32)
    lib().dvo_pair_direction(_p(feat1), _p(feat2), _p(verts1), _p(verts2), N, M, ctypes.c_float(neg_alpha_f32(alpha)),
                             int(fps_start), _p(cw), ctypes.c_float(cb), *[_p(m) for m in mats], int(with_map),
                             _p(warped), _p(verts12), _p(T12), _p(losses))
    return dict(warped=warped, verts12=verts12, T12=T12, losses=losses, chamfer_warp=float(losses[0] + losses[1]),
                arap=float(losses[2]), chamfer_self=float(losses[3] + losses[4]), map_sum=float(losses[5]))


def densify(val, idx, M):
    """sparse (val, idx) -> dense topk_pi matrix (models/loss.py:1339-1347)."""
    N = val.shape[0]
    out = np.zeros((N, M), np.float32)
    np.put_along_axis(out, idx.astype(np.int64), val, axis=1)
    return out


def dot_chain(a, b):
    a, b = _f(a), _f(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().dvo_dot_chain(_p(a), _p(b), a.shape[0], b.shape[0], a.shape[1], _p(out))
    return out


def linear(x, w, bias=None, res=None, alpha=None, beta=None, slope=1.0):
    """Point-major 1x1 conv / linear layer + epilogue: x (M,K), w (Co,K) -> (M,Co) (dvo_linear)."""
    x, w = _f(x), _f(w)
    M, K = x.shape
    Co = w.shape[0]
    opt = [None if t is None else _f(t) for t in (bias, res, alpha, beta)]
    out = np.empty((M, Co), np.float32)
    lib().dvo_linear(_p(x), _p(w), M, K, Co, *[_p(t) for t in opt], ctypes.c_float(slope), _p(out))
    return out


def bn_eval_affine(weight, bias, mean, var, eps):
    """Eval-mode BatchNorm as ATen's CPU kernel folds it: alpha = w / sqrt(var + eps) with a correctly rounded fp32
    sqrt and divide, beta = fma(-mean, alpha, b); applied as y = fma(x, alpha, beta)."""
    w, b, m, v = (np.asarray(t, np.float32) for t in (weight, bias, mean, var))
    inv = np.float32(1) / np.sqrt(v + np.float32(eps))
    alpha = (inv * w).astype(np.float32)
    beta = (b.astype(np.float64) - m.astype(np.float64) * alpha.astype(np.float64)).astype(np.float32)
    return alpha, beta
