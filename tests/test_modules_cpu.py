"""CPU-side checks of the reference-named boundary modules (no compute: that needs the GPU) and of
the torch restatements used as the checker for the floating-point backbone kernels."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from weights_init import reinit  # noqa: E402

from oracle import torch_ref as TR  # noqa: E402


def test_state_dict_contract(golden):
    """Same keys, order-insensitive, and shapes as the reference's Uni3FC / Deformer."""
    import models.model as mm
    g = golden("bb_state_dict_keys")
    sd = mm.Uni3FC(k=40).state_dict()
    ref = dict(zip(g["keys"].tolist(), g["shapes"].tolist()))
    assert set(sd) == set(ref) and len(sd) == 281
    for k, v in sd.items():
        assert str(tuple(v.shape)) == ref[k], k
    assert list(mm.Deformer(10).state_dict().keys()) == g["dkeys"].tolist()
    # tied / aliased tensors, as in the reference
    net = mm.Uni3FC()
    assert net.sa1.q_conv.weight is net.sa1.k_conv.weight
    assert net.conv[1] is net.bn and net.sa2.conv1[1] is net.sa2.bn1


def test_reference_checkpoint_loads(golden):
    import models.model as mm
    w = golden("deformer_scape_r_weights")
    d = mm.Deformer(10)
    d.load_state_dict({k.replace("__", "."): torch.from_numpy(v) for k, v in w.items()})


def test_modules_refuse_cpu_tensors():
    import models.loss as ml
    import models.model as mm
    from dvm._lib import DvmError
    with pytest.raises(DvmError):
        ml.knnsearch_t(torch.randn(1, 8, 128), torch.randn(1, 8, 128))
    with pytest.raises(DvmError):
        mm.SA_Layer(64)(torch.randn(1, 64, 32))
    with pytest.raises(ValueError):                      # no features and no image backbone to make them
        mm.Uni3FC()(torch.randn(1, 3, 32), None, None)
    with pytest.raises(DvmError):                        # the rendering kernels refuse CPU tensors as well
        mm.Uni3FC()(torch.randn(1, 3, 32), None, lambda img: img)


def test_torch_ref_posenc(golden):
    g = golden("bb_posenc")
    out = TR.pos_encoding(torch.from_numpy(g["x"])).numpy()
    assert np.array_equal(out, g["pos"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_torch_ref_sa_layer(golden, mode):
    import models.model as mm
    g = golden("bb_sa_" + mode)
    sa = reinit(mm.SA_Layer(64), salt=2)
    getattr(sa, mode)()
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        xr = TR.sa_attention(x, sa.k_conv.weight, sa.v_conv.weight, sa.v_conv.bias)
        out = x + sa.act(sa.after_norm(sa.trans_conv(x - xr)))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name,C", [("n2p64", 64), ("n2p128", 128)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_torch_ref_n2p(golden, name, C, mode):
    import models.model as mm
    g = golden("bb_%s_%s" % (name, mode))
    blk = reinit((mm.N2PAttention if C == 64 else mm.N2PAttention_DIM)(40), salt=3)
    getattr(blk, mode)()
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        att = TR.n2p_attention(x, torch.from_numpy(g["knn_idx"]), blk.q_conv.weight, blk.k_conv.weight, blk.v_conv.weight)
        y = blk.bn1(x + att)
        out = blk.bn2(y + blk.ff(y))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-5)


def test_geodesic_eval_host_side():
    """eval/geo_mat.py + eval/main.m restatement (host parts): a flat 6 x 6 grid mesh — edge-path geodesics, area
    normalisation, and the error of a map that is exact except for two landmarks."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dv-matcher_amd"))
    import eval_geodesic as eg
    n = 6
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    verts = np.stack([xs.ravel(), ys.ravel(), np.zeros(n * n)], 1)
    vid = lambda i, j: i * n + j
    faces = []
    for i in range(n - 1):
        for j in range(n - 1):
            faces += [[vid(i, j), vid(i + 1, j), vid(i, j + 1)], [vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)]]
    faces = np.array(faces)
    assert abs(eg.mesh_area(verts, faces) - 25.0) < 1e-12
    M = eg.geodesic_distmat(verts, faces, normalize=False)
    assert abs(M[vid(0, 0), vid(0, 5)] - 5.0) < 1e-12           # along an edge row
    assert abs(M[vid(0, 0), vid(5, 0)] - 5.0) < 1e-12
    assert abs(M[vid(0, 5), vid(5, 0)] - 5.0 * np.sqrt(2)) < 1e-9  # the mesh diagonals run that way
    Mn = eg.geodesic_distmat(verts, faces)
    assert np.allclose(Mn, M / 5.0)
    T = np.arange(n * n)
    T[3], T[10] = 4, 16                                           # two wrong matches, one edge away each
    lm = np.arange(n * n)
    err = eg.geodesic_errors(T, lm, lm, Mn)
    assert np.count_nonzero(err) == 2 and abs(err.sum() - 2 * (1.0 / 5.0)) < 1e-12
