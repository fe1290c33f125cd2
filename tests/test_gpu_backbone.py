"""-m gpu: LG-Net operators, the reference-named modules and the whole criterion on the MI355X,
against golden vectors from the reference and the torch / C oracles."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from weights_init import reinit  # noqa: E402

from oracle import oracle as O  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvm import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def rank_equal_up_to_ties(idx, ref, scores):
    bad = np.unique(np.argwhere(idx != ref)[:, 0])
    assert len(bad) <= max(1, idx.shape[0] // 50), len(bad)
    for r in bad:
        k = scores[r][ref[r]]
        for c in np.argwhere(idx[r] != ref[r])[:, 0]:
            run = np.argwhere(k == k[c])[:, 0]
            assert len(run) > 1 and sorted(idx[r][run]) == sorted(ref[r][run]), (r, c)


@pytest.mark.parametrize("name", ["knn_rand_256", "knn_scape_1024", "knn_rand_2048"])
def test_knn_neg(ops, golden, name):
    g = golden(name)
    f = g["feat64"]
    idx = host(ops.knn_neg(dev(f), dev(f), 40))[0]
    assert np.array_equal(idx, O.knn_neg(f[0], f[0], 40))          # oracle: bit-exact incl. tie order
    s = TR.knn_scores(torch.from_numpy(f), torch.from_numpy(f))[0].numpy()
    rank_equal_up_to_ties(idx, g["knn_new_idx"][0], s)               # reference: up to exact fp32 ties
    f = g["feat128"]
    a = f[:, g["anchors"]]
    k = g["knn_idx"].shape[-1]
    idx = host(ops.knn_neg(dev(a), dev(f), k))[0]
    assert np.array_equal(idx, O.knn_neg(a[0], f[0], k))
    rank_equal_up_to_ties(idx, g["knn_idx"][0], TR.knn_scores(torch.from_numpy(a), torch.from_numpy(f))[0].numpy())


def test_knn_neg_shapes(ops):
    g = torch.Generator().manual_seed(11)
    for (B, N, M, C, k) in [(2, 70, 300, 36, 17), (1, 129, 65, 64, 64), (1, 10, 600, 128, 500), (1, 300, 3000, 64, 40),
                            (1, 200, 4995, 128, 40), (2, 64, 8192, 64, 64), (1, 70, 2049, 128, 40)]:
        a, b = torch.randn(B, N, C, generator=g), torch.randn(B, M, C, generator=g)
        b[:, 5] = b[:, 3]  # an exact tie
        idx = host(ops.knn_neg(a.cuda(), b.cuda(), k))
        for bb in range(B):
            assert np.array_equal(idx[bb], O.knn_neg(a[bb].numpy(), b[bb].numpy(), k)), (N, M, C, k)


def test_pos_encoding(ops, golden):
    g = golden("bb_posenc")
    out = host(ops.pos_encoding(dev(g["x"])))
    # low octaves tight; the top octaves evaluate sin/cos at |arg| up to 3e19 where the device libm's
    # range reduction decides everything: compare all of it anyway
    np.testing.assert_allclose(out, g["pos"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_sa_layer(ops, golden, mode):
    import models.model as mm
    g = golden("bb_sa_" + mode)
    sa = reinit(mm.SA_Layer(64), salt=2).cuda()
    getattr(sa, mode)()
    with torch.no_grad():
        out = sa(dev(g["x"]))
    np.testing.assert_allclose(host(out), g["out"], rtol=0, atol=1e-4)
    x = dev(g["x"])
    xr = ops.sa_attention(x, sa.k_conv.weight, sa.v_conv.weight, sa.v_conv.bias)
    ref = TR.sa_attention(x, sa.k_conv.weight, sa.v_conv.weight, sa.v_conv.bias)
    np.testing.assert_allclose(host(xr), host(ref), rtol=0, atol=2e-5)


@pytest.mark.parametrize("name,C", [("n2p64", 64), ("n2p128", 128)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_n2p_block(ops, golden, name, C, mode):
    import models.model as mm
    g = golden("bb_%s_%s" % (name, mode))
    blk = reinit((mm.N2PAttention if C == 64 else mm.N2PAttention_DIM)(40), salt=3).cuda()
    getattr(blk, mode)()
    x = dev(g["x"])
    xt = x.transpose(1, 2).contiguous()
    idx = host(ops.knn_neg(xt, xt, 40))
    assert (idx != g["knn_idx"]).any(-1).mean() < 0.02
    with torch.no_grad():
        out = blk(x)
    np.testing.assert_allclose(host(out), g["out"], rtol=0, atol=1e-4)


def test_deformer_reference_signature(ops, golden):
    """Deformer.forward with the reference's own (B,N,k,128) / dense-Pi arguments."""
    import models.loss as ml
    import models.model as mm
    g = golden("deformer_256x256")
    w = golden("deformer_scape_r_weights")
    d = mm.Deformer(10)
    d.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in w.items()})
    d = d.cuda().eval()
    f1, f2, v1, v2 = dev(g["feat1"]), dev(g["feat2"]), dev(g["verts1"]), dev(g["verts2"])
    Pi = ml.knnsearch_t_grad(f1, f2, alpha=float(g["alpha"]))
    crit = ml.GraphDeformLoss_Neural(save_name="t")
    Pk = crit.topk_pi(Pi)
    v12 = torch.matmul(Pk, v2)
    np.testing.assert_allclose(host(v12), g["verts12"], rtol=0, atol=1e-5)
    idx11, idx22 = ml.knn_grad(v1, v1, 10), ml.knn_grad(v2, v2, 10)
    with torch.no_grad():
        out = d(ml.index_points(f1, idx11), ml.index_points(f2, idx22), v1, v12, Pk, dev(g["fps1"]).long())
    np.testing.assert_allclose(host(out), g["deformations"], rtol=0, atol=1e-4)


def _criterion_case(golden, name, cls_name, kw):
    import models.loss as ml
    import models.model as mm
    g = golden(name)
    w = golden("deformer_scape_r_weights")
    d = mm.Deformer(10)
    d.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in w.items()})
    d = d.cuda().train()
    crit = getattr(ml, cls_name)(save_name="t", **kw)
    f1, f2, v1, v2 = dev(g["feat1"]), dev(g["feat2"]), dev(g["verts1"]), dev(g["verts2"])
    dist1, dist2 = torch.cdist(v1, v1), torch.cdist(v2, v2)
    random.seed(int(g["py_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    with torch.no_grad():
        out = crit(f1, f2, dist1, dist2, v1, v2, np.float64(g["alpha"]), d)
    return g, [float(o) for o in out]


def test_criterion_full(golden):
    kw = dict(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=50, N_dist=100, partial=False, w_deform=0.5, w_img=0,
              w_rank=0, w_self_rec=0.5, w_cd=0.1, w_arap=0.01)
    for name in ("loss_full_256", "loss_full_scape_384", "loss_full_300_unit"):
        g, out = _criterion_case(golden, name, "GraphDeformLoss_Neural", kw)
        ref = [float(g[k]) for k in ("loss", "dist_loss", "deform_loss", "map_loss", "self_rec_loss")]
        np.testing.assert_allclose(out, ref, rtol=2e-4, err_msg=name)


def test_criterion_partial(golden):
    kw = dict(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=30, N_dist=60, partial=True, w_deform=1000, w_img=0,
              w_rank=0, w_self_rec=1000, w_cd=0.1, w_arap=0.01)
    for name in ("loss_partial_256x120", "loss_partial_300x170_unit"):
        g, out = _criterion_case(golden, name, "GraphDeformLoss_Neural_Partial", kw)
        ref = [float(g[k]) for k in ("loss", "dist_loss", "deform_loss", "map_loss", "self_rec_loss")]
        np.testing.assert_allclose(out, ref, rtol=2e-4, err_msg=name)


@pytest.mark.parametrize("name,cls,kw", [
    ("loss_full_256", "GraphDeformLoss_Neural", dict(k_dist=50, N_dist=100, partial=False, w_deform=0.5, w_self_rec=0.5)),
    ("loss_partial_256x120", "GraphDeformLoss_Neural_Partial", dict(k_dist=30, N_dist=60, partial=True, w_deform=1000,
                                                                 w_self_rec=1000)),
    # unit-scale features (|f| below the smallest pairwise distance), alpha 80 / 60, sizes that do not tile
    ("loss_full_300_unit", "GraphDeformLoss_Neural", dict(k_dist=50, N_dist=100, partial=False, w_deform=0.5, w_self_rec=0.5)),
    ("loss_partial_300x170_unit", "GraphDeformLoss_Neural_Partial", dict(k_dist=30, N_dist=60, partial=True, w_deform=1000,
                                                                      w_self_rec=1000)),
    # the rank term ||Pi Pi^T - I||_F switched on (models/loss.py:1427-1433; the constructor's default)
    ("loss_full_rank_160", "GraphDeformLoss_Neural", dict(k_dist=30, N_dist=60, partial=False, w_deform=0.5, w_self_rec=0.5, w_rank=0.3)),
    ("loss_partial_rank_144", "GraphDeformLoss_Neural_Partial", dict(k_dist=30, N_dist=60, partial=True, w_deform=1000,
                                                                  w_self_rec=1000, w_rank=0.3)),
])
def test_criterion_backward_matches_reference(golden, name, cls, kw):
    """Training step parity: loss and gradients w.r.t. feat1, feat2 and every Deformer parameter against the
    reference's autograd (fixture)."""
    import models.loss as ml
    import models.model as mm
    g = golden(name)
    w = golden("deformer_scape_r_weights")
    d = mm.Deformer(10)
    d.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in w.items()})
    d = d.cuda().train()
    kw = dict(dict(w_rank=0), **kw)
    crit = getattr(ml, cls)(save_name="t", k_deform=10, w_dist=0.02, w_map=0.005, w_img=0, w_cd=0.1, w_arap=0.01, **kw)
    f1, f2 = dev(g["feat1"]).requires_grad_(True), dev(g["feat2"]).requires_grad_(True)
    v1, v2 = dev(g["verts1"]), dev(g["verts2"])
    random.seed(int(g["py_seed"]))
    torch.manual_seed(int(g["torch_seed"]))
    out = crit(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, np.float64(g["alpha"]), d)
    ref = [float(g[k]) for k in ("loss", "dist_loss", "deform_loss", "map_loss", "self_rec_loss")]
    np.testing.assert_allclose([float(o) for o in out], ref, rtol=2e-4)
    if kw["w_rank"] > 0:   # not part of the 5-tuple: loss - (the four returned terms)
        np.testing.assert_allclose(float(crit.rank_loss), ref[0] - sum(ref[1:]), rtol=1e-4)
    out[0].backward()

    def close(a, b, what):
        a, b = host(a), np.asarray(b)
        scale = np.abs(b).max() + 1e-30
        assert np.abs(a - b).max() <= 2e-3 * scale, (what, np.abs(a - b).max(), scale)

    close(f1.grad, g["g_feat1"], "feat1")
    close(f2.grad, g["g_feat2"], "feat2")
    for k, p in d.named_parameters():
        close(p.grad, g["g_" + k.replace(".", "__")], k)


@pytest.mark.parametrize("N,C,nA,k", [(300, 128, 40, 25), (700, 128, 33, 64), (700, 128, 9, 130), (1100, 128, 50, 500), (600, 128, 21, 511),
                                      (300, 64, 40, 25)])
def test_dist_loss_vs_torch(ops, N, C, nA, k):
    """C = 128, k <= 512: the half-wave-per-row kernel (odd / even / multi-chunk k); other widths: the
    lane-per-row kernel (k itself is capped at 512 by the kNN selection)."""
    g = torch.Generator().manual_seed(21 + k)
    B = 2
    feat = torch.randn(B, N, C, generator=g).cuda()
    v = torch.rand(B, N, 3, generator=g).cuda()
    dist = torch.cdist(v, v)
    anchors = torch.randperm(N, generator=g)[:nA].cuda()
    out = ops.dist_loss(feat, dist, anchors, k)
    ref = TR.dist_loss_term(feat, dist, anchors, k)
    np.testing.assert_allclose(host(out), host(ref), rtol=1e-5)


def test_softcorr_dense_rows_sum_to_one(ops):
    g = torch.Generator().manual_seed(22)
    f1, f2 = torch.randn(2, 150, 128, generator=g).cuda(), torch.randn(2, 90, 128, generator=g).cuda()
    P = ops.softcorr_dense(f1, f2, 12.5)
    np.testing.assert_allclose(host(P.sum(-1)), 1.0, rtol=1e-5)
    ref = torch.softmax(torch.cdist(f1, f2) * ops.neg_alpha_f32(12.5), -1)
    np.testing.assert_allclose(host(P), host(ref), rtol=0, atol=1e-5)


def test_training_driver_runs():
    """The thin driver (reference train.py step sequence) trains: finite losses, parameters move."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "dv-matcher_amd", "train_driver.py"), "--steps", "3", "--warmup", "0",
                          "--batch", "2", "--points", "256"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["grad_bucket_floats"] == 2122644  # 1 822 592 backbone + 300 052 Deformer
    assert all(np.isfinite(res["first_losses"])) and all(np.isfinite(res["last_losses"]))
    assert res["first_losses"] != res["last_losses"]


def test_inference_driver_writes_reference_outputs(tmp_path, ops):
    """SURVEY §8a row 18 (test.py:95-133): T_<a>_<b>.txt 1-based '%i', usefeature_<a>.mat {'uphi'}; the maps equal
    the oracle's exact arg-min on the features the driver saved."""
    import scipy.io
    sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
    import test_driver
    out = str(tmp_path / "res")
    test_driver.main(["--synthetic", "1", "--points", "300", "--out", out])
    f1 = scipy.io.loadmat(os.path.join(out, "feature", "usefeature_s000a.mat"))["uphi"]
    f2 = scipy.io.loadmat(os.path.join(out, "feature", "usefeature_s000b.mat"))["uphi"]
    T12 = np.loadtxt(os.path.join(out, "T", "T_s000a_s000b.txt"), dtype=np.int64)
    T21 = np.loadtxt(os.path.join(out, "T", "T_s000b_s000a.txt"), dtype=np.int64)
    assert f1.shape == (300, 128) and T12.shape == (300,) and T12.min() >= 1 and T12.max() <= 300
    o12, _ = O.argmin_exact(f1.astype(np.float32), f2.astype(np.float32))
    o21, _ = O.argmin_exact(f2.astype(np.float32), f1.astype(np.float32))
    assert np.array_equal(T12, o12 + 1) and np.array_equal(T21, o21 + 1)


def test_deform_driver_fused_equals_reference_sequence(tmp_path, ops):
    """deform.py:219-262: the fused C-ABI path and the call-by-call reference sequence (dense Pi, gathered
    (B,N,k,128) features, per-shape graph objects) write the same deformed cloud."""
    sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
    import deform_driver

    def read_off(p):
        lines = open(p).read().split("\n")
        assert lines[0] == "OFF"
        n = int(lines[1].split()[0])
        return np.array([[float(x) for x in ln.split()] for ln in lines[2:2 + n]])

    a, b = str(tmp_path / "fused"), str(tmp_path / "seq")
    deform_driver.main(["--synthetic", "1", "--points", "256", "--out", a])
    deform_driver.main(["--synthetic", "1", "--points", "256", "--out", b, "--reference-sequence"])
    pa, pb = read_off(os.path.join(a, "deform_s000a_s000b.off")), read_off(os.path.join(b, "deform_s000a_s000b.off"))
    assert pa.shape == (256, 3)
    np.testing.assert_allclose(pa, pb, rtol=0, atol=1e-4)


def test_knn_neg_heavy_ties(ops):
    """Small-integer features make most scores collide: the wave top-k's tie paths (more than 128 keys at the
    threshold -> exact k-th key search, ties taken in column order) against the oracle, k <= 64 and k > 64."""
    g = torch.Generator().manual_seed(12)
    for (N, M, C, k) in [(40, 700, 64, 40), (33, 2048, 128, 64), (20, 2500, 64, 17), (16, 300, 64, 100), (12, 4995, 128, 40),
                         (8, 8000, 64, 64)]:
        a = torch.randint(-1, 2, (1, N, C), generator=g).float()
        b = torch.randint(-1, 2, (1, M, C), generator=g).float()
        b[0, M // 2:] = b[0, : M - M // 2].clone()   # every key twice
        idx = host(ops.knn_neg(a.cuda(), b.cuda(), k))[0]
        assert np.array_equal(idx, O.knn_neg(a[0].numpy(), b[0].numpy(), k)), (N, M, C, k)


def test_geodesic_eval_uses_the_exact_map(ops):
    """SURVEY §8d 'geodesic error': identical features give error 0; the device map equals the oracle's, so the
    geodesic error of any feature pair equals the one computed from the oracle's map."""
    sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
    import eval_geodesic as eg
    rng = np.random.default_rng(3)
    n = 8
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    verts = np.stack([xs.ravel(), ys.ravel(), 0.1 * rng.standard_normal(n * n)], 1)
    faces = np.array([[i * n + j, (i + 1) * n + j, i * n + j + 1] for i in range(n - 1) for j in range(n - 1)] +
                     [[(i + 1) * n + j, (i + 1) * n + j + 1, i * n + j + 1] for i in range(n - 1) for j in range(n - 1)])
    M = eg.geodesic_distmat(verts, faces)
    phi = rng.standard_normal((n * n, 128)).astype(np.float32)
    lm = np.arange(n * n)
    assert eg.mean_geodesic_error(phi, phi, lm, lm, M) == 0.0
    noisy = (phi + 0.9 * rng.standard_normal(phi.shape)).astype(np.float32)
    T = eg.match(noisy, phi)
    To, _ = O.argmin_exact(noisy, phi)
    assert np.array_equal(T, To)
    assert eg.mean_geodesic_error(noisy, phi, lm, lm, M) == float(eg.geodesic_errors(To, lm, lm, M).mean())


def test_criterion_graph_cache_equals_rebuild(golden):
    """The criterion's opt-in per-shape graph cache (GraphDeformLoss_Neural(graph_cache={}), forward(..., shape_ids=...)):
    batches assembled from cached per-shape graphs give the 5-tuple of the criterion that rebuilds them
    (models/loss.py:1325-1337), bit for bit, across batches that share some shapes; entries are per (shape, FPS start)."""
    import models.loss as ml
    import models.model as mm
    w = golden("deformer_scape_r_weights")
    d = mm.Deformer(10)
    d.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in w.items()})
    d = d.cuda().eval()
    kw = dict(k_deform=10, w_dist=0.02, w_map=0.005, k_dist=40, N_dist=60, partial=False, w_deform=0.5, w_img=0, w_rank=0,
              w_self_rec=0.5, w_cd=0.1, w_arap=0.01, save_name="t")
    plain, cached = ml.GraphDeformLoss_Neural(**kw), ml.GraphDeformLoss_Neural(graph_cache={}, **kw)
    g = torch.Generator().manual_seed(21)
    shapes = torch.rand(5, 300, 3, generator=g).cuda()                     # a "dataset" of 5 shapes
    feats = (0.3 * torch.relu(torch.randn(5, 300, 128, generator=g))).cuda()
    anchors = (list(range(0, 120, 2)), list(range(1, 121, 2)))
    for ids1, ids2 in (([0, 1], [2, 3]), ([1, 4], [0, 2]), ([0, 1], [2, 3])):
        v1, v2, f1, f2 = shapes[ids1], shapes[ids2], feats[ids1], feats[ids2]
        starts = (torch.tensor([7 * i % 300 for i in ids1]), torch.tensor([11 * i % 300 for i in ids2]))
        with torch.no_grad():
            a = plain(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, 50.0, d, fps_starts=starts, anchors=anchors)
            b = cached(f1, f2, torch.cdist(v1, v1), torch.cdist(v2, v2), v1, v2, 50.0, d, fps_starts=starts, anchors=anchors, shape_ids=(ids1, ids2))
        assert [float(x) for x in a] == [float(x) for x in b]
    assert len(cached.graph_cache) == 5            # (shape, start): (0,0) (1,7) (2,22) (3,33) (4,28); every other lookup was a hit
