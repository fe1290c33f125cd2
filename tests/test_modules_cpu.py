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
        sd = {"sa." + k: v for k, v in sa.state_dict().items()}
        out = TR.sa_layer(sd, "sa", x, train=(mode == "train"))
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
        sd = {"blk." + k: v for k, v in blk.state_dict().items()}
        out = TR.n2p_block(sd, "blk", x, train=(mode == "train"), idx=torch.from_numpy(g["knn_idx"]))
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


@pytest.mark.parametrize("name,mode", [("bb_uni3fc_eval", "eval"), ("bb_uni3fc_train", "train")])
def test_torch_ref_uni3fc_pinned_to_reference(golden, name, mode):
    """The whole-network restatement (oracle/torch_ref.py::uni3fc, used as checker on the GPU at sizes without a fixture)
    against the canonical (1-thread) reference run: same neighbour sets at all 7 layers, features to float rounding."""
    from weights_init import dino_from_seed
    import models.model as mm
    g = golden(name)
    sd = reinit(mm.Uni3FC(k=40), salt=4).state_dict()
    B, _, N = g["xyz"].shape
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        log = []
        with torch.no_grad():
            feat, cf = TR.uni3fc(sd, torch.from_numpy(g["xyz"]), dino_from_seed(int(g["dino_seed"]), B, N), train=(mode == "train"), log=log)
    finally:
        torch.set_num_threads(nt)
    assert len(log) == 7
    for l in range(7):
        assert np.array_equal(np.sort(log[l].numpy(), -1), np.sort(g["knn_idx"][l].astype(np.int64), -1)), l
    np.testing.assert_allclose(cf.numpy(), g["cfeats"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(feat.numpy(), g["feat"], rtol=0, atol=2e-5 * max(1.0, float(np.abs(g["feat"]).max()) / 8))


def test_reference_noise_summary_is_what_the_docs_say(golden):
    """DESIGN §2 / tests/test_gpu_network.py quote these: with unit-gain weights the reference evaluated by 1 and by 8
    CPU threads disagrees on about half of the points of a 1024-point SCAPE shape; with damped weights it is stable."""
    s = golden("bb_noise_summary")
    assert float(s["rows_gt_2e-3_t8"]) > 0.3 and float(s["T12_agree_t8"]) < 0.9
    g = golden("bb_uni3fc_scape1024_eval")
    self_err = np.abs(g["feat_t8"] - g["feat"]).max(-1)
    assert (self_err > 1e-4).mean() < 0.05 and self_err.max() < 2e-3
    assert 0 < (self_err > 1e-4).sum()          # ... but even there single neighbour flips happen


def test_bn_act_pm_module_fallback_and_counter_batching():
    """bn_act_pm away from the fused kernels (CPU tensors here; SyncBatchNorm / eval mode on a GPU take the same branch):
    the module itself on a transposed view — statistics, running buffers and activation as nn.BatchNorm1d over (B,C,N)
    gives them; and the batched counter context leaves module-path counters to the module."""
    from dvm import nn_ops
    torch.manual_seed(3)
    x, res = torch.randn(2, 50, 8), torch.randn(2, 50, 8)
    a, b = torch.nn.BatchNorm1d(8), torch.nn.BatchNorm1d(8)
    b.load_state_dict(a.state_dict())
    with nn_ops.batched_counter_updates():
        y = nn_ops.bn_act_pm(a, x, res, slope=0.2)
    ref = torch.nn.functional.leaky_relu(b((x + res).transpose(1, 2)), 0.2).transpose(1, 2)
    assert torch.equal(y, ref)
    assert torch.equal(a.running_var, b.running_var) and int(a.num_batches_tracked) == int(b.num_batches_tracked) == 1
    a.eval(), b.eval()
    assert torch.equal(nn_ops.bn_act_pm(a, x, None, slope=0.0), torch.relu(b(x.transpose(1, 2))).transpose(1, 2))


def test_grad_accumulation_switch_roundtrip():
    from dvm import nn_ops
    prev = nn_ops.fuse_grad_accumulation(True)
    try:
        p = torch.nn.Parameter(torch.zeros(3))
        assert nn_ops._grad_buffer(p) is None                       # no .grad yet: autograd's own path
        p.grad = torch.ones(3)
        assert nn_ops._grad_buffer(p) is p.grad
        assert nn_ops._grad_buffer(p * 2) is None                   # not a leaf
        nn_ops.fuse_grad_accumulation(False)
        assert nn_ops._grad_buffer(p) is None
    finally:
        nn_ops.fuse_grad_accumulation(prev)


def test_train_pointer_table_cache_follows_the_parameters():
    """Uni3FC._train_state() caches the 167-entry parameter table of the native training node: same entry while the parameters
    stay where they are (an in-place optimizer step, a state_dict load), a new one after a conversion (`_apply`), a mode change,
    or invalidate_train_state(); on CPU parameters the table says the native path does not apply."""
    from models.model import Uni3FC
    net = Uni3FC(k=8)
    st = net._train_state()
    assert len(st[0]) == 167 and len(st[4]) == 167 and st[5] is False      # CPU parameters: not for the native path
    assert net._train_state() is st
    with torch.no_grad():
        net.conv[0].weight.add_(1.0)                                        # in-place update: same storage, same entry
    assert net._train_state() is st and st[4][0].data_ptr() == net.conv[0].weight.data_ptr()
    assert torch.equal(st[4][0], net.conv[0].weight.detach())
    net.load_state_dict(net.state_dict())
    st2 = net._train_state()
    assert st2 is not st and st2[4][0].data_ptr() == net.conv[0].weight.data_ptr()
    net.double()
    st3 = net._train_state()
    assert st3 is not st2 and st3[4][0].dtype == torch.float64
    net.eval()
    assert net._train_state() is not st3
    st4 = net._train_state()
    net.invalidate_train_state()
    assert net._train_state() is not st4


def test_aten_cpu_leg_matches_the_c_oracle(golden):
    """bench.py's second cpu_baseline figure (oracle/torch_ref.py::pair_direction_aten: the pair path in the reference's dense ATen
    formulation) computes what the C oracle computes: arg-max map equal up to fp32 ties of the two cdist forms, losses to 1e-4."""
    from oracle import oracle as O
    w = golden("deformer_scape_r_weights")
    g = torch.Generator().manual_seed(31)
    N, M = 384, 300
    f1, f2 = torch.randn(N, 128, generator=g).numpy(), torch.randn(M, 128, generator=g).numpy()
    v1, v2 = torch.rand(N, 3, generator=g).numpy(), torch.rand(M, 3, generator=g).numpy()
    for with_map in (True, False):
        a = TR.pair_direction_aten(w, f1, f2, v1, v2, 100.0, 5, with_map=with_map)
        o = O.pair_direction(w, f1, f2, v1, v2, 100.0, 5, with_map=with_map)
        assert (a["T12"] != o["T12"]).mean() < 5e-3
        np.testing.assert_allclose(a["losses"], o["losses"], rtol=2e-4, atol=1e-7)
