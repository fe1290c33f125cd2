"""-m gpu: the HIP backward kernels (C ABI `*_bwd_f32`) against fp64 autograd of the reference formulas
(oracle/torch_ref.py).  Gradients are floating point: the bar is a relative L2 error <= 1e-4 per tensor
(north_star's tolerance) unless a test states why it is looser."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvm import ops as _ops
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _ops


def rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).norm() / (ref.norm() + 1e-300))


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("shape,alpha", [((2, 200, 150, 128), 10.0), ((1, 130, 333, 128), 100.0), ((2, 64, 64, 128), 37.5),
                                         ((1, 257, 129, 128), 10.0)])
def test_softcorr_bwd_vs_fp64_autograd(ops, shape, alpha, variant):
    B, N, M, d = shape
    g = torch.Generator().manual_seed(N * 1000 + M)
    scale = 0.25 if alpha >= 30 else 1.0  # keep the softmax from being one-hot so every term carries gradient
    f1 = (torch.randn(B, N, d, generator=g) * scale).cuda()
    f2 = (torch.randn(B, M, d, generator=g) * scale).cuda()
    gval = torch.randn(B, N, 10, generator=g).cuda()
    val, idx, smax, ssum = ops.softcorr(f1, f2, alpha)
    df1, df2 = ops.softcorr_bwd(f1, f2, alpha, val, idx, smax, ssum, gval, variant=variant)
    rval, rf1, rf2 = TR.softcorr_bwd(f1.cpu(), f2.cpu(), ops.neg_alpha_f32(alpha), idx.cpu(), gval.cpu())
    assert rel(val, rval) < 1e-4
    # s = -alpha*D is evaluated in fp32 (|s| ~ alpha*D, one ulp of D moves a weight by ~alpha*D*6e-8), so the
    # gradient's error scales with alpha: 1e-4 holds at alpha = 10; at alpha = 100 the fp32 reference itself
    # sits at a few 1e-4 of this fp64 ground truth
    tol = 1e-4 if alpha <= 40 else 1e-3
    assert rel(df1, rf1) < tol and rel(df2, rf2) < tol, (rel(df1, rf1), rel(df2, rf2))


@pytest.mark.parametrize("shape,alpha", [((2, 300, 250, 128), 50.0), ((1, 333, 130, 128), 100.0), ((2, 2048, 2200, 128), 50.0),
                                         ((1, 4995, 2200, 128), 50.0)])
def test_softcorr_bwd_padded_tiles_stay_finite(ops, shape, alpha):
    """Unit-scale features and a large alpha: the zero-filled padding rows of the last key tile sit much closer than any
    real column (|f| ~ 11 vs a minimum distance ~ 14), their softmax factor overflows, and it must not reach the
    output (0 * inf).  The matrix-core kernel against the scalar one, both passes, sizes that do not tile."""
    B, N, M, d = shape
    g = torch.Generator().manual_seed(N + M)
    f1, f2 = torch.randn(B, N, d, generator=g).cuda(), torch.randn(B, M, d, generator=g).cuda()
    gval = torch.randn(B, N, 10, generator=g).cuda()
    val, idx, smax, ssum = ops.softcorr(f1, f2, alpha)
    a1, a2 = ops.softcorr_bwd(f1, f2, alpha, val, idx, smax, ssum, gval, variant=2)
    s1, s2 = ops.softcorr_bwd(f1, f2, alpha, val, idx, smax, ssum, gval, variant=1)
    assert torch.isfinite(a1).all() and torch.isfinite(a2).all()
    assert rel(a1, s1) < 2e-3 and rel(a2, s2) < 2e-3, (rel(a1, s1), rel(a2, s2))


def test_softcorr_bwd_other_dims_and_duplicates(ops):
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(1, 70, 36, generator=g)
    f2 = torch.randn(1, 90, 36, generator=g)
    f2[0, 3] = f1[0, 5]  # an exact zero distance: contributes no gradient (cdist's backward convention)
    f1, f2 = f1.cuda(), f2.cuda()
    gval = torch.randn(1, 70, 10, generator=g).cuda()
    val, idx, smax, ssum = ops.softcorr(f1, f2, 2.0)
    df1, df2 = ops.softcorr_bwd(f1, f2, 2.0, val, idx, smax, ssum, gval)
    _, rf1, rf2 = TR.softcorr_bwd(f1.cpu(), f2.cpu(), ops.neg_alpha_f32(2.0), idx.cpu(), gval.cpu())
    assert torch.isfinite(df1).all() and torch.isfinite(df2).all()
    assert rel(df1, rf1) < 1e-4 and rel(df2, rf2) < 1e-4


def test_softcorr_bwd_split_and_full_size(ops):
    """B=1 at N=M=2048 takes the split-inner path (atomics); compare the two kernels with each other."""
    g = torch.Generator().manual_seed(6)
    f1 = (torch.randn(1, 2048, 128, generator=g) * 0.3).cuda()
    f2 = (torch.randn(1, 2048, 128, generator=g) * 0.3).cuda()
    gval = torch.randn(1, 2048, 10, generator=g).cuda()
    val, idx, smax, ssum = ops.softcorr(f1, f2, 10.0)
    a1, a2 = ops.softcorr_bwd(f1, f2, 10.0, val, idx, smax, ssum, gval, variant=2)
    b1, b2 = ops.softcorr_bwd(f1, f2, 10.0, val, idx, smax, ssum, gval, variant=1)
    assert rel(a1, b1) < 2e-5 and rel(a2, b2) < 2e-5, (rel(a1, b1), rel(a2, b2))


@pytest.mark.parametrize("C,K,N", [(64, 40, 300), (128, 40, 257), (128, 7, 50), (64, 64, 130), (64, 5, 33), (64, 1, 9), (128, 1, 5)])
def test_n2p_core_fwd_bwd_vs_fp64_autograd(ops, C, K, N):
    B, H = 2, 4
    g = torch.Generator().manual_seed(C + K)
    qkv = torch.randn(B, N, 3 * C, generator=g)
    idx = torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32)
    gout = torch.randn(B, N, C, generator=g)
    out, attn = ops.n2p_core_fwd(qkv.cuda(), idx.cuda(), H)
    dqkv = ops.n2p_core_bwd(qkv.cuda(), idx.cuda(), attn, gout.cuda(), H)
    # fp64 reference: the reference's formulation (models/model.py:339-350) on projected rows
    x = qkv.double().requires_grad_(True)
    q, kp, vp = x[..., :C], x[..., C:2 * C], x[..., 2 * C:]
    gi = idx.long().reshape(B, N * K, 1).expand(-1, -1, C)
    kd = (torch.gather(kp, 1, gi).view(B, N, K, C) - kp[:, :, None]).view(B, N, K, H, C // H)
    vd = (torch.gather(vp, 1, gi).view(B, N, K, C) - vp[:, :, None]).view(B, N, K, H, C // H)
    e = (q.view(B, N, 1, H, C // H) * kd).sum(-1) / (C // H) ** 0.5
    a = torch.softmax(e, dim=2)
    ref = (a.unsqueeze(-1) * vd).sum(2).reshape(B, N, C)
    (ref * gout.double()).sum().backward()
    assert rel(out, ref) < 1e-5 and rel(attn, a) < 1e-5
    assert rel(dqkv, x.grad) < 1e-5, rel(dqkv, x.grad)


@pytest.mark.parametrize("atomics", [False, True])
@pytest.mark.parametrize("C", [3, 30, 128, 64, 200])
def test_sparse_apply_bwd_vs_fp64_autograd(ops, C, atomics):
    B, N, M, k = 2, 190, 75, 10
    g = torch.Generator().manual_seed(C)
    val = torch.rand(B, N, k, generator=g)
    idx = torch.randint(0, M, (B, N, k), generator=g, dtype=torch.int32)
    idx[:, :, -1] = torch.where(idx[:, :, -1] > M // 2, idx[:, :, -1] - M // 2, idx[:, :, -1])   # skewed: rows 0..M/2 are hot
    idx[1, :40] = 5                                                                           # a hub: 400 in-edges, shared by the four waves of its workgroup
    V = torch.randn(B, M, C, generator=g)
    gout = torch.randn(B, N, C, generator=g)
    dval, dV = ops.apply_bwd(val.cuda(), idx.cuda(), V.cuda(), gout.cuda(), atomics=atomics)
    v64, V64 = val.double().requires_grad_(True), V.double().requires_grad_(True)
    rows = torch.gather(V64, 1, idx.long().reshape(B, N * k, 1).expand(-1, -1, C)).view(B, N, k, C)
    ((v64.unsqueeze(-1) * rows).sum(2) * gout.double()).sum().backward()
    assert rel(dval, v64.grad) < 1e-5 and rel(dV, V64.grad) < 1e-5


def test_dist_loss_bwd_vs_fp64_autograd(ops):
    from dvm import nn_ops
    B, N, C, nA, k = 2, 300, 128, 40, 25
    g = torch.Generator().manual_seed(31)
    feat = torch.randn(B, N, C, generator=g)
    v = torch.rand(B, N, 3, generator=g)
    dist = torch.cdist(v, v)
    anchors = torch.randperm(N, generator=g)[:nA]
    gout = torch.randn(B, generator=g)
    f = feat.cuda().requires_grad_(True)
    out = nn_ops.dist_loss(f, dist.cuda(), anchors.cuda(), k)
    (out * gout.cuda()).sum().backward()
    idx = ops.dist_loss(feat.cuda(), dist.cuda(), anchors.cuda(), k, want_idx=True)[1].cpu().long()
    f64 = feat.double().requires_grad_(True)   # the reference's formulation, models/loss.py:1351-1396
    f1 = f64[:, anchors]
    f2 = torch.gather(f64, 1, idx.reshape(B, nA * k, 1).expand(-1, -1, C)).view(B, nA, k, C)
    d2 = ((f2 - f1[:, :, None, :]) ** 2).sum(-1)
    x = torch.where(d2 > 0, torch.sqrt(d2.clamp_min(1e-300)), torch.zeros_like(d2))
    y = torch.stack([dist[b].double()[idx[b], anchors[:, None]] for b in range(B)])
    ref = (1 - torch.abs(torch.nn.functional.cosine_similarity(x, y, dim=2))).sum(1)
    (ref * gout.double()).sum().backward()
    assert rel(out, ref) < 1e-5
    assert rel(f.grad, f64.grad) < 1e-4, rel(f.grad, f64.grad)


def test_rot6d_warp_arap_bwd_vs_fp64_autograd(ops):
    from dvm import nn_ops
    B, N = 2, 400
    g = torch.Generator().manual_seed(77)
    verts = torch.rand(B, N, 3, generator=g).cuda()
    graph = ops.dg_build(verts, torch.tensor([3, 11], dtype=torch.int32).cuda())
    Nn = N // 2
    d6 = (torch.randn(B, Nn, 6, generator=g) * 0.3 + torch.tensor([1., 0, 0, 0, 1, 0])).cuda().requires_grad_(True)
    T = (torch.randn(B, Nn, 3, generator=g) * 0.05).cuda().requires_grad_(True)
    gw = torch.randn(B, N, 3, generator=g).cuda()
    ga = torch.randn(B, generator=g).cuda()
    warped, arap = nn_ops.dg_warp_arap(verts, graph, nn_ops.rot6d(d6), T)
    ((warped * gw).sum() + (arap * ga).sum()).backward()
    # fp64 checker: the oracle's torch restatement under autograd (itself pinned by the reference's gradients below)
    d64 = d6.detach().double().requires_grad_(True)
    t64 = T.detach().double().requires_grad_(True)
    g64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in graph.items() if torch.is_tensor(v)}
    w_ref, a_ref = TR.dg_warp_arap(verts.double(), g64, TR.rot6d(d64), t64)
    ((w_ref * gw.double()).sum() + (a_ref * ga.double()).sum()).backward()
    assert rel(warped, w_ref) < 1e-5 and rel(arap, a_ref) < 1e-5
    assert rel(d6.grad, d64.grad) < 1e-4 and rel(T.grad, t64.grad) < 1e-4, (rel(d6.grad, d64.grad), rel(T.grad, t64.grad))


@pytest.mark.parametrize("name", ["graddg_scape_512", "graddg_rand_256"])
def test_rot6d_warp_arap_bwd_vs_reference_gradients(ops, name):
    """The reference's own autograd through rotation_6d_to_matrix -> DeformationGraph_geod.forward (fp32, recorded by
    tests/golden/make_fixtures.py dg_grad) against the HIP forward + backward kernels on the reference's graph, and
    against the oracle's fp64 restatement, which is what the other backward tests check with."""
    import os
    from dvm import nn_ops
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    cu = lambda k, dt=None: torch.from_numpy(fx[k] if dt is None else fx[k].astype(dt)).cuda()   # noqa: E731
    verts = cu("verts")[None]
    graph = {"nodes_idx": cu("nodes_idx", np.int32)[None], "one_ring": cu("one_ring", np.int32)[None],
             "infl_idx": cu("infl_idx", np.int32)[None], "weights": cu("weights", np.float32)[None]}
    d6 = cu("d6").requires_grad_(True)
    T = cu("T").requires_grad_(True)
    gw, ga = cu("gw")[None], cu("ga").reshape(1)
    warped, arap = nn_ops.dg_warp_arap(verts, graph, nn_ops.rot6d(d6), T)
    ((warped * gw).sum() + (arap * ga).sum()).backward()
    assert rel(warped, cu("warped")) < 2e-6 and rel(arap, cu("arap").reshape(1)) < 1e-5
    assert rel(d6.grad, cu("d6_grad")) < 2e-5 and rel(T.grad, cu("T_grad")) < 2e-5, (rel(d6.grad, cu("d6_grad")), rel(T.grad, cu("T_grad")))
    d64, t64 = cu("d6").double().requires_grad_(True), cu("T").double().requires_grad_(True)
    g64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in graph.items()}
    w_ref, a_ref = TR.dg_warp_arap(verts.double(), g64, TR.rot6d(d64), t64)
    ((w_ref * gw.double()).sum() + (a_ref * ga.double()).sum()).backward()
    assert rel(cu("d6_grad"), d64.grad) < 2e-5 and rel(cu("T_grad"), t64.grad) < 2e-5


def test_chamfer_bwd_vs_autograd(ops):
    from dvm import nn_ops
    g = torch.Generator().manual_seed(78)
    a = torch.rand(2, 300, 3, generator=g).cuda().requires_grad_(True)
    b = torch.rand(2, 170, 3, generator=g).cuda().requires_grad_(True)
    g1, g2 = torch.randn(2, 300, generator=g).cuda(), torch.randn(2, 170, generator=g).cuda()
    d1, d2 = nn_ops.chamfer_nn(a, b)
    ((d1 * g1).sum() + (d2 * g2).sum()).backward()
    a64, b64 = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    D = ((a64[:, :, None] - b64[:, None]) ** 2).sum(-1)
    ((D.min(2)[0] * g1.double()).sum() + (D.min(1)[0] * g2.double()).sum()).backward()
    assert rel(a.grad, a64.grad) < 1e-5 and rel(b.grad, b64.grad) < 1e-5


@pytest.mark.parametrize("B,N", [(2, 300), (1, 128), (1, 1000), (3, 77)])
def test_sa_core_fwd_bwd_vs_fp64_autograd(ops, B, N):
    g = torch.Generator().manual_seed(N)
    p = torch.randn(B, N, 16, generator=g) * 0.7
    v = torch.randn(B, N, 64, generator=g)
    gx = torch.randn(B, N, 64, generator=g)
    xr, stats, cinv = ops.sa_attention_train_fwd(p.cuda(), v.cuda())
    dp, dv = ops.sa_attention_bwd(p.cuda(), v.cuda(), xr, stats, cinv, gx.cuda())
    p64, v64 = p.double().requires_grad_(True), v.double().requires_grad_(True)   # models/model.py:113-121, point-major
    att = torch.softmax(torch.bmm(p64, p64.transpose(1, 2)), dim=-1)
    att = att / (1e-9 + att.sum(dim=1, keepdim=True))
    ref = torch.bmm(att.transpose(1, 2), v64)
    (ref * gx.double()).sum().backward()
    assert rel(xr, ref) < 1e-5
    assert rel(dv, v64.grad) < 1e-4 and rel(dp, p64.grad) < 1e-4, (rel(dv, v64.grad), rel(dp, p64.grad))


@pytest.mark.parametrize("B,C,N,slope,with_res", [(2, 64, 300, 0.2, False), (3, 128, 257, 1.0, True), (1, 384, 1024, 0.2, False),
                                                   (8, 64, 2048, 0.0, False), (2, 16, 5, 1.0, True)])
def test_fused_batchnorm_vs_torch(ops, B, C, N, slope, with_res):
    """dvm_bn_act_train_{fwd,bwd}_f32 == nn.BatchNorm1d (train) around a residual add and a (Leaky)ReLU, in fp64."""
    from dvm import nn_ops
    g = torch.Generator().manual_seed(B * 1000 + C + N)
    x = (torch.randn(B, C, N, generator=g) * 2.0 + torch.randn(1, C, 1, generator=g) * 3.0)
    res = torch.randn(B, C, N, generator=g) if with_res else None
    gout = torch.randn(B, C, N, generator=g)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref = torch.nn.BatchNorm1d(C).double()
    ref.load_state_dict(bn.state_dict())
    bn = bn.cuda().train()
    xg, rg = x.cuda().requires_grad_(True), (res.cuda().requires_grad_(True) if with_res else None)
    y = nn_ops.bn_act(bn, xg, rg, slope=slope)
    (y * gout.cuda()).sum().backward()
    xd = x.double().requires_grad_(True)
    rd = res.double().requires_grad_(True) if with_res else None
    t = ref(xd if rd is None else xd + rd)
    yr = t if slope == 1.0 else torch.where(t > 0, t, t * slope)
    (yr * gout.double()).sum().backward()
    assert rel(y, yr) < 2e-6
    assert rel(xg.grad, xd.grad) < 2e-5, rel(xg.grad, xd.grad)
    if with_res:
        assert rel(rg.grad, rd.grad) < 2e-5
    assert rel(bn.weight.grad, ref.weight.grad) < 2e-5 and rel(bn.bias.grad, ref.bias.grad) < 2e-5
    assert rel(bn.running_mean, ref.running_mean) < 1e-6 and rel(bn.running_var, ref.running_var) < 1e-6
    assert int(bn.num_batches_tracked) == 1
    # eval mode / SyncBatchNorm take the module's own path
    bn.eval()
    ye = nn_ops.bn_act(bn, x.cuda(), None if res is None else res.cuda(), slope=slope)
    te = ref.eval()(x.double() if res is None else x.double() + res.double())
    assert rel(ye, te if slope == 1.0 else torch.where(te > 0, te, te * slope)) < 2e-6
