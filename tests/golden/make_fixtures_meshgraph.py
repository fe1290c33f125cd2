"""Golden vectors for the mesh-mode deformation graph (runs in the BUILD container only).

TEST INFRASTRUCTURE.  Runs the *reference's* lib.mesh_sampling.generate_transform_matrices and
lib.deformation_graph_point.DeformationGraph_geod.construct_graph / forward (stub-imported; psbody.mesh.Mesh replaced
by a two-field container, which is all this path uses of it) on small closed meshes and records inputs + outputs as
tests/golden/meshgraph_*.npz.

    cd /tmp && python /root/repo/tests/golden/make_fixtures_meshgraph.py
"""
import os
import sys

import numpy as np
import torch
from scipy.sparse.csgraph import shortest_path
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


class Mesh:
    def __init__(self, v=None, f=None):
        self.v, self.f = np.asarray(v), np.asarray(f)


def closed_mesh(seed, n, bumpy):
    rs = np.random.RandomState(seed)
    p = rs.randn(n, 3)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    faces = ConvexHull(p).simplices.astype(np.int64)
    v = p * (1.0 + bumpy * rs.rand(n, 1)) * np.array([1.0, 0.7, 1.3])
    return v.astype(np.float32).astype(np.float64), faces


def edge_geodesics(v, f):
    import scipy.sparse as sp
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    w = np.linalg.norm(v[e[:, 0]] - v[e[:, 1]], axis=1)
    g = sp.coo_matrix((w, (e[:, 0], e[:, 1])), shape=(len(v), len(v))).tocsr()
    return shortest_path(g.maximum(g.T), method="D", directed=False).astype(np.float32)


def main():
    _, _, rdg = ref_import.import_reference()
    sys.path.insert(0, ref_import.REF)
    import lib.mesh_sampling as rms
    sys.path.remove(ref_import.REF)
    rms.Mesh = Mesh
    rdg.Mesh = Mesh
    rdg.generate_transform_matrices = rms.generate_transform_matrices
    for tag, seed, n, bumpy in (("a", 1, 120, 0.2), ("b", 2, 333, 0.05), ("c", 3, 64, 0.5)):
        v, f = closed_mesh(seed, n, bumpy)
        M, A, D = rms.generate_transform_matrices(Mesh(v=v, f=f), [2])
        geod = edge_geodesics(v, f)
        dg = rdg.DeformationGraph_geod()
        dg.one_ring_neigh = []
        dg.construct_graph(torch.from_numpy(v).float(), f, geod, torch.device("cpu"))
        g = torch.Generator().manual_seed(seed)
        Nn = len(dg.nodes_idx)
        R = torch.eye(3).expand(1, Nn, 3, 3) + 0.1 * torch.randn(1, Nn, 3, 3, generator=g)
        T = 0.05 * torch.randn(1, Nn, 3, generator=g)
        warped, arap, sr = dg(torch.from_numpy(v).float(), R, T)
        out = dict(verts=v, faces=f, geod=geod, ds_faces=M[1].f, ds_verts=M[1].v, nodes_idx=np.asarray(dg.nodes_idx),
                   adjacency=np.stack([A[1].row, A[1].col], 1), one_ring=dg.one_ring_neigh.numpy(),
                   infl_idx=dg.influence_nodes_idx.numpy(), dists=dg.dists.numpy(), weights=dg.weights.numpy(),
                   sigma=float(dg.sigma), R=R.numpy(), T=T.numpy(), warped=warped.numpy(), arap=float(arap), sr=float(sr))
        path = os.path.join(HERE, "meshgraph_%s.npz" % tag)
        np.savez_compressed(path, **out)
        print("wrote %s (%.1f KB): %d -> %d nodes" % (path, os.path.getsize(path) / 1024, n, Nn))


if __name__ == "__main__":
    main()
