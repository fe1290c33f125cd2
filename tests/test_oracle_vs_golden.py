"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_fixtures.py).  Integer outputs must be bit-exact; floats within 1e-4
(most are far tighter; distances are bit-exact)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_rank_equal_up_to_ties(idx, ref_idx, key_of_ref, max_tie_rows=0.01):
    """idx must equal ref_idx except inside runs of EXACTLY equal keys (the reference's
    std::partial_sort / nth_element order among equal values is unspecified); inside a run the
    two index sets must match."""
    idx, ref_idx = np.asarray(idx), np.asarray(ref_idx)
    bad_rows = np.unique(np.argwhere(idx != ref_idx)[:, 0])
    assert len(bad_rows) <= max(1, int(max_tie_rows * idx.shape[0])), "too many rows differ: %d" % len(bad_rows)
    for r in bad_rows:
        k = key_of_ref[r]
        for c in np.argwhere(idx[r] != ref_idx[r])[:, 0]:
            run = np.argwhere(k == k[c])[:, 0]
            assert len(run) > 1, (r, c, idx[r], ref_idx[r])
            assert sorted(idx[r][run]) == sorted(ref_idx[r][run]), (r, idx[r], ref_idx[r])


def neg_scores(a, b):
    """-|a|^2 + 2ab - |b|^2 with torch CPU ops (only used to find exact ties)."""
    import torch
    a, b = torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))
    inner = -2 * torch.matmul(a, b.T)
    return (-(a ** 2).sum(1, keepdim=True) - inner - (b ** 2).sum(1, keepdim=True).T).numpy()


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@pytest.mark.parametrize("name", names("softcorr_"))
def test_softcorr(golden, name):
    g = golden(name)
    f1, f2 = g["feat1"][0], g["feat2"][0]
    M = f2.shape[0]
    # distances: the exact form is bit-exact with torch.cdist; the matmul form is bit-exact before
    # the sqrt (k-ordered fmaf chain == MKL sgemm) but ATen's vectorised sqrt_ (MKL VML) is not
    # correctly rounded: <=1 ulp apart in <2% of entries.
    dmm = O.cdist(f1[:1], f2)[0]
    np.testing.assert_array_max_ulp(dmm, g["dist_mm_row0"], maxulp=1)
    assert (dmm != g["dist_mm_row0"]).mean() < 0.02
    assert np.array_equal(O.cdist(f1[:1], f2, exact=True)[0], g["dist_exact_row0"])
    val, idx, smax, ssum = O.softcorr(f1, f2, float(g["alpha"]))
    # integer outputs: arg-min map and top-10 columns wherever the reference's value is non-zero
    T, _ = O.argmin_exact(f1, f2)
    assert float(g["exact_gap"]) > 0
    assert np.array_equal(T, g["T12"][0, :, 0])
    rv, ri = g["topk_val"][0], g["topk_idx"][0]
    # columns must agree wherever the reference's value is unique in its row (equal subnormal /
    # zero values make torch.topk's choice arbitrary without changing the dense matrix)
    pad = np.pad(rv, ((0, 0), (1, 1)), constant_values=-1.0)
    uniq = (rv != pad[:, :-2]) & (rv != pad[:, 2:]) & (rv > 0)
    assert uniq.sum() > rv.shape[0]
    assert np.array_equal(idx[uniq], ri[uniq])
    assert np.array_equal(idx[:, 0], ri[:, 0])
    np.testing.assert_allclose(val, rv, rtol=0, atol=1e-4)
    np.testing.assert_allclose(val, rv, rtol=3e-4, atol=1e-30)  # alpha x 1 ulp of the sqrt
    if "Pi_topk_dense" in g:
        np.testing.assert_allclose(O.densify(val, idx, M), g["Pi_topk_dense"][0], rtol=0, atol=1e-4)
    v12 = O.apply(val, idx, g["verts2"][0])
    np.testing.assert_allclose(v12, g["verts12"][0], rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", names("knn_"))
def test_knn(golden, name):
    g = golden(name)
    v = g["verts"][0]
    ref = g["knn_grad_idx"][0]
    assert_rank_equal_up_to_ties(O.knn_cdist(v, v, 10), ref, np.take_along_axis(O.cdist(v, v), ref.astype(np.int64), 1))
    f = g["feat64"][0]
    ref = g["knn_new_idx"][0]
    assert_rank_equal_up_to_ties(O.knn_neg(f, f, 40), ref, np.take_along_axis(neg_scores(f, f), ref.astype(np.int64), 1))
    f = g["feat128"][0]
    k = g["knn_idx"].shape[-1]
    a = f[g["anchors"]]
    ref = g["knn_idx"][0]
    assert_rank_equal_up_to_ties(O.knn_neg(a, f, k), ref, np.take_along_axis(neg_scores(a, f), ref.astype(np.int64), 1))


@pytest.mark.parametrize("name", names("dg_"))
def test_dg(golden, name):
    g = golden(name)
    v = g["verts"]
    N = v.shape[0]
    assert np.array_equal(O.fps(v, N // 2, int(g["fps_start"])), g["nodes_idx"])
    b = O.dg_build(v, int(g["fps_start"]))
    assert np.array_equal(b["nodes_idx"], g["nodes_idx"])
    assert np.array_equal(b["one_ring"], g["one_ring"])
    assert np.array_equal(b["infl_idx"], g["infl_idx"])
    np.testing.assert_array_max_ulp(b["dists"], g["dists"], maxulp=1)  # see test_softcorr on sqrt_
    np.testing.assert_allclose(b["sigma"], float(g["sigma"]), rtol=1e-12)
    np.testing.assert_allclose(b["weights"], g["weights"], rtol=0, atol=2e-7)
    R, T = O.rot6d(np.concatenate([g["T"][0], g["d6"][0]], -1))
    np.testing.assert_allclose(R, g["R"][0], rtol=0, atol=1e-6)
    warped, arap, sr = O.dg_warp_arap(v, b, R, T)
    np.testing.assert_allclose(warped, g["warped"][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(arap, float(g["arap"]), rtol=1e-5)
    np.testing.assert_allclose(sr, float(g["sr"]), rtol=1e-5)


@pytest.mark.parametrize("name", ["deformer_256x256", "deformer_300x200"])
def test_deformer_and_chamfer(golden, name):
    g = golden(name)
    w = golden("deformer_scape_r_weights")
    B = g["feat1"].shape[0]
    for b in range(B):
        f1, f2, v1, v2 = g["feat1"][b], g["feat2"][b], g["verts1"][b], g["verts2"][b]
        val, idx, _, _ = O.softcorr(f1, f2, float(g["alpha"]))
        v12 = O.apply(val, idx, v2)
        np.testing.assert_allclose(v12, g["verts12"][b], rtol=0, atol=1e-5)
        idx11, idx22 = O.knn_cdist(v1, v1, 10), O.knn_cdist(v2, v2, 10)
        out = O.deformer(w, f1, f2, v1, v12, idx11, idx22, val, idx, g["fps1"][b])
        np.testing.assert_allclose(out, g["deformations"][b], rtol=0, atol=1e-4)
        d1, d2, i1, i2 = O.chamfer(g["verts12"][b], v2)  # same inputs as the stand-in saw
        np.testing.assert_allclose(d1, g["ch_d1"][b], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(d2, g["ch_d2"][b], rtol=1e-5, atol=1e-9)
        assert (i1 == g["ch_i1"][b]).mean() > 0.999 and (i2 == g["ch_i2"][b]).mean() > 0.999


def test_aten_sum_order():
    """dvo_rownorm2 restates ATen's cascade sum; torch is the live reference of that op here."""
    import torch
    g = torch.Generator().manual_seed(0)
    for K in (1, 3, 5, 7, 8, 9, 31, 64, 100, 128, 130, 257, 1000, 1152, 2048, 5000):
        x = torch.randn(257, K, generator=g)
        assert np.array_equal(O.rownorm2(x.numpy()), x.pow(2).sum(-1).numpy()), K


def _linear_chain_cases(g):
    for i in range(int(g["n"])):
        yield (g["x%d" % i], g["w%d" % i][:, :, 0], g.get("b%d" % i), g["bn%d" % i], g["y%d" % i], g["z%d" % i])


def test_linear_chain_pinned_to_reference_conv(golden):
    """dvo_linear (the checker of dvm_linear_f32) against nn.Conv1d(k=1) [+ eval BatchNorm + LeakyReLU] outputs recorded
    from torch evaluated by ONE thread in the build container, bit for bit, at every reduction length of LG-Net: a
    K-blocked fma chain (blocks of 384 while more than 768 remain, then one block or two halves), bias after the chain,
    BatchNorm as fma(y, alpha, beta)."""
    g = golden("linear_chain")
    for x, w, b, bn, y, z in _linear_chain_cases(g):
        a, be = O.bn_eval_affine(bn[0], bn[1], bn[2], bn[3], 1e-5)
        for bb in range(x.shape[0]):
            xt = np.ascontiguousarray(x[bb].T)
            assert np.array_equal(O.linear(xt, w, bias=b).T, y[bb]), w.shape
            assert np.array_equal(O.linear(xt, w, bias=b, alpha=a, beta=be, slope=0.2).T, z[bb]), w.shape


def test_linear_chain_vs_live_torch():
    """The same rule against torch.matmul on THIS host for many K (MKL's sgemm: identical on every x86 host we saw);
    Conv1d goes through oneDNN whose kernels vary with the CPU: compared only where this host's single-thread Conv1d
    equals its own matmul.  Also shows that the multi-threaded Conv1d is a different function."""
    import torch
    nt = torch.get_num_threads()
    g = torch.Generator().manual_seed(3)
    try:
        torch.set_num_threads(1)
        for K in [4, 20, 64, 128, 256, 384, 385, 400, 512, 700, 768, 769, 772, 1000, 1152, 1156, 2048, 2304]:
            W = torch.randn(40, K, generator=g) / K ** 0.5
            x = torch.randn(K, 70, generator=g)
            mm = torch.matmul(W, x)
            if not np.array_equal(O.linear(x.t().contiguous().numpy(), W.numpy()).T, mm.numpy()):
                pytest.skip("this host's sgemm blocks K differently from the build container's (K=%d)" % K)
            conv = torch.nn.functional.conv1d(x[None], W[:, :, None])[0]
            if torch.equal(conv, mm):
                assert np.array_equal(O.linear(x.t().contiguous().numpy(), W.numpy()).T, conv.numpy())
        conv = torch.nn.Conv1d(1152, 384, 1, bias=False)
        x = torch.randn(1, 1152, 2048, generator=g)
        with torch.no_grad():
            y1 = conv(x)
            torch.set_num_threads(8)
            y8 = conv(x)
        print("1-thread vs 8-thread Conv1d identical on this host:", torch.equal(y1, y8))
    finally:
        torch.set_num_threads(nt)


@pytest.mark.parametrize("name", ["graddg_scape_512", "graddg_rand_256"])
def test_torch_ref_warp_arap_gradients_pinned(name):
    """oracle/torch_ref.py's rot6d + warp/ARAP restatement (the fp64 checker of the HIP backward kernels) against the
    reference's own outputs and autograd gradients (tests/golden/make_fixtures.py dg_grad)."""
    import torch
    from oracle import torch_ref as TR
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")))
    t = lambda k: torch.from_numpy(fx[k])   # noqa: E731
    g = {"nodes_idx": t("nodes_idx")[None], "one_ring": t("one_ring")[None], "infl_idx": t("infl_idx")[None],
         "weights": t("weights").double()[None]}
    d6, T = t("d6").double().requires_grad_(True), t("T").double().requires_grad_(True)
    w, a = TR.dg_warp_arap(t("verts").double()[None], g, TR.rot6d(d6), T)
    ((w * t("gw").double()[None]).sum() + (a * t("ga").double().reshape(1)).sum()).backward()
    rel = lambda x, y: float((x.detach().double() - y.double()).norm() / y.double().norm())   # noqa: E731
    assert rel(w, t("warped").reshape(w.shape)) < 1e-6 and rel(a, t("arap").reshape(1)) < 1e-6
    assert rel(d6.grad, t("d6_grad")) < 2e-6 and rel(T.grad, t("T_grad")) < 2e-6
