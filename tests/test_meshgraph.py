"""Mesh-mode deformation graph: the QSlim decimation (host code, like the reference's) and the graph /
warp / ARAP built on it, against vectors recorded from the reference (tests/golden/make_fixtures_meshgraph.py)."""
import numpy as np
import pytest
import torch

CASES = ["a", "b", "c"]


@pytest.mark.parametrize("case", CASES)
def test_decimation_matches_reference(golden, case):
    from lib import mesh_sampling as ms
    g = golden("meshgraph_" + case)
    M, A, D = ms.generate_transform_matrices(ms.Mesh(v=g["verts"], f=g["faces"]), [2])
    assert np.array_equal(M[1].f, g["ds_faces"])                       # the same edges collapsed in the same order
    assert np.array_equal(M[1].v, g["ds_verts"])
    assert np.array_equal(D[0].nonzero()[1], g["nodes_idx"])
    assert np.array_equal(np.stack([A[1].row, A[1].col], 1), g["adjacency"])
    n = len(g["verts"])
    assert len(g["nodes_idx"]) == int(np.ceil(n / 2)) and D[0].shape == (len(g["nodes_idx"]), n)


def test_decimation_edge_cases():
    from lib import mesh_sampling as ms
    # a tetrahedron asked to keep everything: nothing collapses
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
    nf, mtx = ms.qslim_decimator_transformer(ms.Mesh(v=v, f=f), factor=1.0)
    assert np.array_equal(nf, f) and mtx.shape == (4, 4)
    with pytest.raises(Exception):
        ms.qslim_decimator_transformer(ms.Mesh(v=v, f=f))
    # connectivity: symmetric, no self loops, an edge shared by two faces counted once per direction pair
    c = ms.get_vert_connectivity(v, f).toarray()
    assert np.array_equal(c > 0, ~np.eye(4, dtype=bool)) and ms.get_vertices_per_edge(v, f).shape == (6, 2)
    q = ms.vertex_quadrics(ms.Mesh(v=v, f=f))
    hom = np.concatenate([v, np.ones((4, 1))], 1)
    assert np.allclose([hom[i] @ q[i] @ hom[i] for i in range(4)], 0.0, atol=1e-12)   # every vertex lies on its own planes


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_mesh_graph_matches_reference(golden, case):
    from lib.deformation_graph_point import DeformationGraph_geod
    g = golden("meshgraph_" + case)
    dev = torch.device("cuda", 0)
    verts = torch.from_numpy(g["verts"]).float().to(dev)
    dg = DeformationGraph_geod()
    dg.construct_graph(verts, g["faces"], g["geod"], dev)
    assert dg.max_neigh_num == 18
    assert np.array_equal(np.asarray(dg.nodes_idx), g["nodes_idx"])
    assert np.array_equal(dg.one_ring_neigh.numpy(), g["one_ring"])
    assert np.array_equal(dg.influence_nodes_idx.cpu().numpy(), g["infl_idx"])
    np.testing.assert_allclose(dg.dists.numpy(), g["dists"], rtol=0, atol=0)
    np.testing.assert_allclose(float(dg.sigma), g["sigma"], rtol=1e-12)
    np.testing.assert_allclose(dg.weights.cpu().numpy(), g["weights"], rtol=1e-6, atol=1e-7)
    warped, arap, sr = dg(verts, torch.from_numpy(g["R"]).to(dev), torch.from_numpy(g["T"]).to(dev))
    assert warped.shape == (1, len(g["verts"]), 3)
    np.testing.assert_allclose(warped.cpu().numpy(), g["warped"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(float(arap), g["arap"], rtol=1e-5)
    np.testing.assert_allclose(float(sr), g["sr"], rtol=1e-5)
