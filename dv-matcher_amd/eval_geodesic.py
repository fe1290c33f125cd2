#!/usr/bin/env python3
"""Geodesic-error evaluation of hard correspondence maps — restatement of the reference's
eval/geo_mat.py:15-41 (geodesic matrix = Dijkstra over the mesh edges) and eval/main.m:16-45 (per pair: nearest
neighbour of the source features among the target's, error = geodesic distance on the target between the matched
vertex and the ground-truth vertex, averaged).

The nearest-neighbour search is the hot-path operator `knnsearch_t` (exact-difference arg-min, HIP kernel); the
geodesic matrices are preprocessing on the host, like the reference's own script.  Ground-truth files (`.vts`,
`M_*.mat`) are not shipped with the reference: this module takes arrays.

  err = mean_geodesic_error(phi_src, phi_tgt, vts_src, vts_tgt, M_tgt)
"""
import numpy as np


def mesh_area(verts, faces):
    """eval/surfaceArea.m: sum of triangle areas."""
    v = np.cross(verts[faces[:, 0]] - verts[faces[:, 1]], verts[faces[:, 0]] - verts[faces[:, 2]])
    return float(np.sqrt((v ** 2).sum(1)).sum() / 2)


def geodesic_distmat(verts, faces, normalize=True):
    """eval/geo_mat.py:15-41: shortest paths over the mesh edges weighted by their Euclidean lengths; with
    `normalize` divided by sqrt(surface area) (the M matrices the MATLAB evaluation loads)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import shortest_path
    verts = np.asarray(verts, np.float64)
    faces = np.asarray(faces, np.int64)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e = np.unique(np.sort(e, 1), axis=0)
    w = np.linalg.norm(verts[e[:, 0]] - verts[e[:, 1]], axis=1)
    n = verts.shape[0]
    adj = coo_matrix((np.concatenate([w, w]), (np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]]))), shape=(n, n))
    import torch
    if torch.cuda.is_available() and n * 8 <= 150 * 1024:   # same values as scipy's Dijkstra, one workgroup per source
        from models.dataset import _shortest_paths_gpu
        geo = _shortest_paths_gpu(adj.tocsr(), n)
    else:
        geo = shortest_path(adj.tocsr(), directed=False)
    if np.isinf(geo).any():
        raise ValueError("mesh graph is not connected")
    return geo / np.sqrt(mesh_area(verts, faces)) if normalize else geo


def match(phi_src, phi_tgt):
    """T[i] = argmin_j |phi_src[i] - phi_tgt[j]| (0-based), on the device (models/loss.py:91-95 semantics)."""
    import torch
    from dvm import ops
    a = torch.as_tensor(np.ascontiguousarray(phi_src), dtype=torch.float32).cuda()[None]
    b = torch.as_tensor(np.ascontiguousarray(phi_tgt), dtype=torch.float32).cuda()[None]
    return ops.argmin_exact(a, b)[0].cpu().numpy().astype(np.int64)


def geodesic_errors(T, vts_src, vts_tgt, M_tgt):
    """eval/main.m:34-41 with a precomputed map T (source vertex -> target vertex, 0-based): for every ground-truth
    landmark l, error_l = M_tgt[T[vts_src[l]], vts_tgt[l]]."""
    T = np.asarray(T).reshape(-1)
    return np.asarray(M_tgt)[T[np.asarray(vts_src)], np.asarray(vts_tgt)]


def mean_geodesic_error(phi_src, phi_tgt, vts_src, vts_tgt, M_tgt):
    return float(geodesic_errors(match(phi_src, phi_tgt), vts_src, vts_tgt, M_tgt).mean())
