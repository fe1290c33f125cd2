"""Dataset mirror: OFF reader, cache tuples, pair enumeration and item layout against
vectors recorded from the reference's own classes (tests/golden/make_fixtures_dataset.py)."""
import os

import numpy as np
import pytest
import torch

from models import dataset as ds


@pytest.fixture(scope="module")
def gold(golden):
    return golden("dataset_items")


def _cache_lists(g):
    names = [str(n) for n in g["names"]]
    verts = [torch.from_numpy(g["verts%d" % i]) for i in range(len(names))]
    fps = [torch.from_numpy(g["fps%d" % i]) for i in range(len(names))]
    dist = [torch.from_numpy(g["dist%d" % i]) for i in range(len(names))]
    return verts, names, fps, dist


def test_off_reader_matches_reference(gold, tmp_path):
    p = tmp_path / "a.off"
    p.write_text(str(gold["off_text"]))
    pts = np.asarray(ds.load_off_point_cloud(str(p)), dtype=np.float64)
    assert np.array_equal(pts, gold["off_points"])
    verts, faces = ds.read_mesh(str(p))
    assert np.array_equal(verts, gold["off_points"])
    assert faces.tolist() == [[0, 1, 2], [1, 2, 3]]


def test_off_reader_variants(tmp_path):
    p = tmp_path / "b.off"
    p.write_text("OFF 3 1 0\n# a comment\n0 0 0\n1 0 0  # trailing\n\n0 1 0\n4 0 1 2 0\n")
    assert ds.load_off_point_cloud(str(p)) == [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]
    _, faces = ds.read_mesh(str(p))
    assert faces.tolist() == [[0, 1, 2], [0, 2, 0]]                  # quad fan-triangulated
    bad = tmp_path / "c.off"
    bad.write_text("PLY\n1 0 0\n0 0 0\n")
    with pytest.raises(ValueError):
        ds.load_off_point_cloud(str(bad))
    short = tmp_path / "d.off"
    short.write_text("OFF\n3 0 0\n0 0 0\n")
    with pytest.raises(ValueError):
        ds.load_off_point_cloud(str(short))


@pytest.mark.parametrize("dsname", ["scape_r", "fourleg"])
def test_train_items_match_reference(gold, tmp_path, dsname):
    verts, names, fps, dist = _cache_lists(gold)
    torch.save((verts, names, fps, dist), str(tmp_path / ("cache_%s_train.pt" % dsname)))
    d = ds.Dataset(str(tmp_path), name=dsname, train=True, use_cache=True)
    assert np.array_equal(np.asarray(d.combinations), gold["%s_combinations" % dsname])
    assert len(d) == len(names) * (len(names) - 1)
    for p in (0, len(d) // 2, len(d) - 1):
        item = d[p]
        for s in ("shape1", "shape2"):
            assert np.array_equal(item[s]["xyz"].numpy(), gold["%s_item%d_%s_xyz" % (dsname, p, s)])
            assert np.array_equal(item[s]["dist"].numpy(), gold["%s_item%d_%s_dist" % (dsname, p, s)])
            assert item[s]["name"] == str(gold["%s_item%d_%s_name" % (dsname, p, s)])
            assert item[s]["feat"].numel() == 0


def test_inference_items_match_reference(gold, tmp_path):
    verts, names, fps, _ = _cache_lists(gold)
    torch.save((verts, names, fps), str(tmp_path / "cache_scape_r_test_test.pt"))
    t = ds.testDataset(str(tmp_path), name="scape_r", train=False, use_cache=True)
    assert np.array_equal(np.asarray(t.combinations), gold["test_combinations"])
    item = t[7]
    for s in ("shape1", "shape2"):
        assert np.array_equal(item[s]["xyz"].numpy(), gold["test_item7_%s_xyz" % s])
        assert item[s]["name"] == str(gold["test_item7_%s_name" % s])
        assert item[s]["dist"].numel() == int(gold["test_item7_%s_dist_numel" % s]) == 0


def test_pair_enumeration_special_sets(gold, tmp_path):
    verts, names, fps, dist = _cache_lists(gold)
    torch.save((verts, names, fps, dist), str(tmp_path / "cache_amass_ssft_train.pt"))
    d = ds.Dataset(str(tmp_path), name="amass_ssft", train=True)
    animals = [i for i, n in enumerate(names) if n.split("_")[0] in ("horse", "cat", "dog")]
    others = [i for i in range(len(names)) if i not in animals]
    want = [(a, b) for a in animals for b in animals if a != b] + [(a, b) for a in others for b in others if a != b]
    assert d.combinations == want                                     # never an animal against a human
    many = [torch.zeros(1, 3)] * 41
    torch.save((many, ["s%02d" % i for i in range(41)], [torch.zeros(1, dtype=torch.long)] * 41),
               str(tmp_path / "cache_tosca_test_test.pt"))
    t = ds.testDataset(str(tmp_path), name="tosca", train=False)
    groups = [(0, 11), (11, 17), (17, 26), (26, 30), (30, 38), (38, 41)]
    assert len(t) == sum((hi - lo) * (hi - lo - 1) for lo, hi in groups)
    assert all(any(lo <= a < hi and lo <= b < hi for lo, hi in groups) for a, b in t.combinations)


def test_feature_files(gold, tmp_path):
    import scipy.io as sio
    verts, names, fps, dist = _cache_lists(gold)
    torch.save((verts, names, fps, dist), str(tmp_path / "cache_scape_r_train.pt"))
    d = ds.Dataset(str(tmp_path), name="scape_r", train=True, with_dino=True, feat_mat=True)
    with pytest.raises(FileNotFoundError):
        d[0]
    os.makedirs(str(tmp_path / "feat"))
    feats = {}
    for i, n in enumerate(names):
        feats[n] = np.random.RandomState(i).randn(verts[i].shape[0], 6).astype(np.float32)
        sio.savemat(str(tmp_path / "feat" / (n + ".mat")), {"feat": feats[n]})
    item = d[3]
    i1, i2 = d.combinations[3]
    assert np.array_equal(item["shape1"]["feat"].numpy(), feats[names[i1]][fps[i1].numpy()])
    assert np.array_equal(item["shape2"]["feat"].numpy(), feats[names[i2]][fps[i2].numpy()])
    with pytest.raises(NotImplementedError):
        ds.Dataset(str(tmp_path), name="scape_r", train=True, with_dino=True, feat_mat=False)[0]


def test_cal_geo_known_distances():
    # a 6x5 unit grid with the 4-neighbour graph: intrinsic distance = Manhattan distance
    xs, ys = np.meshgrid(np.arange(6.0), np.arange(5.0), indexing="ij")
    V = np.stack([xs.ravel(), ys.ravel(), np.zeros(30)], 1)
    d = ds.cal_geo(V, k=4).numpy()
    man = np.abs(V[:, None, 0] - V[None, :, 0]) + np.abs(V[:, None, 1] - V[None, :, 1])
    inner = [i for i in range(30) if 0 < V[i, 0] < 5 and 0 < V[i, 1] < 4]
    assert np.allclose(d[np.ix_(inner, inner)], man[np.ix_(inner, inner)])   # corners pick a diagonal as 4th neighbour
    assert np.allclose(d, d.T) and np.all(np.diag(d) == 0)
    assert np.all(d <= man + 1e-6) and np.all(d >= np.linalg.norm(V[:, None] - V[None], axis=-1) - 1e-6)
    # mesh mode: two triangles sharing an edge, path must follow edges
    Vm = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=np.float64)
    dm = ds.cal_geo(Vm, faces=np.array([[0, 1, 2], [1, 3, 2]])).numpy()
    assert np.isclose(dm[0, 3], 2.0) and np.isclose(dm[1, 2], np.sqrt(2.0))
    # two far-apart clusters: bridged by the straight line, still finite
    two = np.concatenate([np.random.RandomState(0).rand(12, 3), np.random.RandomState(1).rand(12, 3) + 100.0])
    dt = ds.cal_geo(two, k=3).numpy()
    assert np.isfinite(dt).all() and dt[0, 20] > 100.0
    assert ds.cal_geo(np.zeros((0, 3))).shape == (0, 0)


def test_shape_to_device_and_no_gpu_build(gold, tmp_path):
    verts, names, fps, dist = _cache_lists(gold)
    torch.save((verts, names, fps, dist), str(tmp_path / "cache_scape_r_train.pt"))
    item = ds.shape_to_device(ds.Dataset(str(tmp_path), name="scape_r")[1], torch.device("cpu"))
    assert item["shape1"]["xyz"].device.type == "cpu"
    if not torch.cuda.is_available():
        os.makedirs(str(tmp_path / "shapes_test"))
        (tmp_path / "shapes_test" / "x.off").write_text(str(gold["off_text"]))
        with pytest.raises(RuntimeError, match="HIP device"):         # building needs the GPU; no host stand-in
            ds.testDataset(str(tmp_path), name="scape_r", train=False)


# ------------------------------------------------------------------ GPU: build from .off files
def _write_off(path, verts, faces=None):
    faces = [] if faces is None else faces
    with open(path, "w") as f:
        f.write("OFF\n%d %d 0\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
        for t in faces:
            f.write("3 %d %d %d\n" % tuple(t))


@pytest.mark.gpu
def test_build_from_off_files(tmp_path):
    from oracle import oracle as orc
    rs = np.random.RandomState(5)
    os.makedirs(str(tmp_path / "shapes_train"))
    clouds = {}
    for n, name in ((300, "b_shape"), (257, "a_shape"), (64, "DS_skipped"), (411, "c_shape")):
        clouds[name] = rs.rand(n, 3).astype(np.float32)
        _write_off(str(tmp_path / "shapes_train" / (name + ".off")), clouds[name])
    torch.manual_seed(3)
    d = ds.Dataset(str(tmp_path), name="se-ornet-tosca", train=True)          # point-cloud OFF, FPS order cut to 1024
    assert d.used_shapes == ["a_shape", "b_shape", "c_shape"]
    assert os.path.exists(str(tmp_path / "cache_se-ornet-tosca_train.pt"))
    for i, name in enumerate(d.used_shapes):
        v = clouds[name]
        assert np.array_equal(d.verts_list[i].numpy(), v)
        fps = d.fps_list[i].numpy()
        assert sorted(fps.tolist()) == list(range(len(v)))                     # a full FPS order is a permutation
        want = orc.fps(v, len(v), int(fps[0]))                 # same order as the CPU restatement
        assert np.array_equal(fps, want)
        assert d.dist_list[i].shape == (len(v), len(v))
    item = d[0]
    i1, _ = d.combinations[0]
    assert np.array_equal(item["shape1"]["xyz"].numpy(), clouds[d.used_shapes[i1]][d.fps_list[i1].numpy()])
    again = ds.Dataset(str(tmp_path), name="se-ornet-tosca", train=True)       # second time: from the cache, identical
    assert all(torch.equal(a, b) for a, b in zip(d.fps_list, again.fps_list))
    assert all(torch.equal(a, b) for a, b in zip(d.dist_list, again.dist_list))


@pytest.mark.gpu
def test_partial_views(tmp_path):
    rs = np.random.RandomState(9)
    n = 400
    verts = [torch.from_numpy(rs.rand(n, 3).astype(np.float32)) for _ in range(2)]
    dist = [torch.cdist(v, v) for v in verts]
    fps = [torch.randperm(n)[:300] for _ in range(2)]
    torch.save((verts, ["s0", "s1"], fps, dist), str(tmp_path / "cache_scape_partial_train.pt"))
    os.makedirs(str(tmp_path / "index_partial"))
    views = {}
    for s in ("s0", "s1"):
        for v in range(1, 13):
            size = 60 if v % 2 else 150 + v                                    # odd views are too small
            views[(s, v)] = np.sort(rs.permutation(n)[:size])
            np.savetxt(str(tmp_path / "index_partial" / ("index_%s_view_%d.txt" % (s, v))), views[(s, v)], fmt="%d")
    d = ds.PartialDataset(str(tmp_path), name="scape_partial", train=True)
    d.n_partial = 100
    for _ in range(4):
        item = d[0]                                                            # (s0 full, s1 partial)
        assert item["shape1"]["xyz"].shape == (300, 3)
        xyz2, dist2 = item["shape2"]["xyz"], item["shape2"]["dist"]
        assert xyz2.shape == (100, 3) and dist2.shape == (100, 100)
        full = verts[1]
        ids = [int(torch.nonzero((full == p).all(1))[0, 0]) for p in xyz2]
        assert len(set(ids)) == 100
        assert any(set(ids) <= set(views[("s1", v)].tolist()) for v in range(2, 13, 2))
        assert torch.equal(dist2, dist[1][ids][:, ids])


@pytest.mark.gpu
def test_drivers_read_a_dataset_directory(tmp_path):
    """End to end from .off files: models/dataset.py builds the caches, train_driver.py trains on Dataset items,
    test_driver.py writes T_<a>_<b>.txt for every ordered pair of testDataset."""
    import json
    import subprocess
    import sys
    import scipy.io as sio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(2)
    names = ["p0", "p1", "p2"]
    for split in ("shapes_train", "shapes_test"):
        os.makedirs(str(tmp_path / split))
    os.makedirs(str(tmp_path / "feat"))
    for n in names:
        v = rs.rand(320, 3).astype(np.float32)
        for split in ("shapes_train", "shapes_test"):
            _write_off(str(tmp_path / split / (n + ".off")), v)
        sio.savemat(str(tmp_path / "feat" / (n + ".mat")), {"feat": rs.randn(320, 1152).astype(np.float32)})
    out = subprocess.run([sys.executable, os.path.join(root, "dv-matcher_amd", "train_driver.py"), "--steps", "2", "--warmup", "0",
                          "--batch", "2", "--points", "256", "--data-root", str(tmp_path), "--data-name", "se-ornet-tosca"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["points"] == 256 and all(np.isfinite(res["first_losses"])) and all(np.isfinite(res["last_losses"]))
    sys.path.insert(0, os.path.join(root, "dv-matcher_amd"))
    import test_driver
    res_dir = str(tmp_path / "res")
    test_driver.main(["--data-root", str(tmp_path), "--data-name", "se-ornet-tosca", "--out", res_dir])
    for a in names:
        for b in names:
            if a != b:
                T = np.loadtxt(os.path.join(res_dir, "T", "T_%s_%s.txt" % (a, b)), dtype=np.int64)
                assert T.shape == (320,) and T.min() >= 1 and T.max() <= 320


@pytest.mark.gpu
def test_gpu_geodesics_equal_dijkstra():
    """dvm_graph_geodesics_f64 (one workgroup per source, in-LDS relaxation) against scipy's Dijkstra on the same graph:
    a k-NN graph of a cloud, a mesh, and two disconnected pieces (+inf between them)."""
    from scipy.sparse.csgraph import shortest_path
    rs = np.random.RandomState(4)
    V = rs.rand(1500, 3)
    g = ds._dedup_min(*_knn_edges(V, 8), len(V))
    want = shortest_path(g, method="D", directed=False)
    got = ds._shortest_paths_gpu(g, len(V))
    assert np.array_equal(got, want)                                   # the same fp64 path sums, bit for bit
    two = np.concatenate([rs.rand(40, 3), rs.rand(40, 3) + 10.0])
    g2 = ds._dedup_min(*_knn_edges(two, 3), len(two))
    got2 = ds._shortest_paths_gpu(g2, len(two))
    assert np.array_equal(got2, shortest_path(g2, method="D", directed=False)) and np.isinf(got2[0, 50])
    d = ds.cal_geo(V).numpy()                                          # the public entry takes the GPU path on this box
    assert np.array_equal(d, want.astype(np.float32))


def _knn_edges(V, k):
    from scipy.spatial import cKDTree
    n = len(V)
    _, nbr = cKDTree(V).query(V, k + 1)
    src, dst = np.repeat(np.arange(n), k + 1), nbr.reshape(-1)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    return src, dst, np.linalg.norm(V[src] - V[dst], axis=1)
