"""QSlim-style mesh decimation and the level-to-level transforms built from it — host code, as in the reference
(lib/mesh_sampling.py:12-216, itself adapted from CoMA's mesh_operations.py), which runs it once per shape when a
mesh-mode deformation graph is built (lib/deformation_graph_point.py:203-231, deform.py:167-216).

Same functions, arguments and results.  The decimation is restated around the same binary heap of (cost, edge)
entries, popped, re-priced and re-pushed in the same order; what differs is bookkeeping only:
  * the reference renames a collapsed vertex in every queue entry by scanning the whole queue after each collapse
    (O(V * E) overall); here an entry resolves its endpoints through a union-find `alias` table whenever the heap
    compares or pops it — the tuples the heap sees are the same at every comparison, so ties break identically;
  * the face planes come from one batched SVD instead of one call per face (same LAPACK routine per matrix); their
    quadrics are accumulated face by face, corner by corner, and the collapse costs use the reference's own matrix
    expressions, so every cost has the same bits and near-ties order the same way.
`Mesh` is a two-field container standing in for psbody.mesh.Mesh (only .v and .f are used on this path).
"""
import heapq
import math

import numpy as np
import scipy.sparse as sp


class Mesh:
    def __init__(self, v=None, f=None):
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f, dtype=np.int64)


def _directed_face_edges(faces):
    """(3F,) sources and targets of the faces' directed edges, in corner order 0->1, 1->2, 2->0."""
    f = np.asarray(faces, dtype=np.int64)
    return f.reshape(-1), np.roll(f, -1, axis=1).reshape(-1)


def get_vert_connectivity(mesh_v, mesh_f):
    """Sparse #verts x #verts adjacency (csc): entry (a, b) counts how often a-b occurs as a face edge, either way round
    (lib/mesh_sampling.py:12-30 of the reference; only the non-zero pattern is used downstream)."""
    n = len(mesh_v)
    src, dst = _directed_face_edges(mesh_f)
    key, cnt = np.unique(np.concatenate([dst * n + src, src * n + dst]), return_counts=True)   # column-major keys
    col, row = np.divmod(key, n)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(indptr, col + 1, 1)
    return sp.csc_matrix((cnt.astype(np.float64), row, np.cumsum(indptr)), shape=(n, n))


def get_vertices_per_edge(mesh_v, mesh_f):
    """E x 2 array of vertex pairs, every undirected edge once with the lower index first, ordered by the higher index
    (then the lower) — the order in which a column-major adjacency lists them."""
    src, dst = _directed_face_edges(mesh_f)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    n = len(mesh_v)
    key = np.unique(hi * n + lo)
    return np.stack([key % n, key // n], axis=1)


def vertex_quadrics(mesh):
    """(N,4,4): per vertex, the sum over its faces of the outer product of the face's normalised plane equation."""
    tri = np.concatenate([mesh.v[mesh.f], np.ones((len(mesh.f), 3, 1))], axis=2)      # (F,3,4) rows [x y z 1]
    _, _, vt = np.linalg.svd(tri)                                                      # null vector of each 3x4 = its plane
    eq = vt[:, -1, :]
    eq = eq / np.linalg.norm(eq[:, :3], axis=1, keepdims=True)
    q = np.zeros((len(mesh.v), 4, 4))
    for f_idx in range(len(mesh.f)):               # the reference's accumulation order (face by face, corner by corner)
        outer = np.outer(eq[f_idx], eq[f_idx])
        for k in range(3):
            q[mesh.f[f_idx, k], :, :] += outer
    return q


def _get_sparse_transform(faces, num_original_verts):
    """Faces re-indexed to the surviving vertices (in ascending original order) and the V' x V selection matrix."""
    survivors = np.unique(faces)
    new_faces = np.searchsorted(survivors, faces).reshape(-1, 3)
    pick = sp.csc_matrix((np.ones(len(survivors)), (np.arange(len(survivors)), survivors)),
                         shape=(len(survivors), num_original_verts))
    return new_faces, pick


def qslim_decimator_transformer(mesh, factor=None, n_verts_desired=None):
    """-> (new_faces (F',3), mtx (V' x V) selecting the surviving vertices)."""
    if factor is None and n_verts_desired is None:
        raise ValueError("qslim_decimator_transformer: give `factor` or `n_verts_desired`")
    if n_verts_desired is None:
        n_verts_desired = math.ceil(len(mesh.v) * factor)
    Qv = vertex_quadrics(mesh)
    v = mesh.v
    alias = np.arange(len(v))

    def resolve(i):
        root = i
        while alias[root] != root:
            root = alias[root]
        while alias[i] != root:
            alias[i], i = root, alias[i]
        return root

    class Entry:
        """(cost, (r, c)) with the endpoints read through `alias`: what the reference's queue holds after its renames."""
        __slots__ = ("cost", "r", "c")

        def __init__(self, cost, r, c):
            self.cost, self.r, self.c = cost, r, c

        def key(self):
            return (self.cost, (resolve(self.r), resolve(self.c)))

        def __lt__(self, other):
            return self.key() < other.key()

    homog = np.concatenate([v, np.ones((len(v), 1))], axis=1)          # (V,4) homogeneous coordinates

    def quadric_error(Q, i):
        """h_i^T Q h_i, evaluated as the (1x4)(4x4)(4x1) matrix chain so that near-equal costs order as in the reference."""
        h = homog[i:i + 1]
        return float((h @ Q @ h.T)[0, 0])

    def collapse_cost(r, c):
        Qsum = Qv[r] + Qv[c]
        destroy_c, destroy_r = quadric_error(Qsum, r), quadric_error(Qsum, c)
        return destroy_c, destroy_r, min(destroy_c, destroy_r), Qsum

    edges = get_vertices_per_edge(mesh.v, mesh.f)
    adj = sp.csc_matrix((np.ones(len(edges), dtype=np.int64), (edges[:, 0], edges[:, 1])), shape=(len(v), len(v)))
    adj = (adj + adj.T).tocoo()
    queue = []
    for r, c in zip(adj.row, adj.col):
        if r > c:
            continue
        heapq.heappush(queue, Entry(collapse_cost(r, c)[2], int(r), int(c)))
    faces = mesh.f.copy()
    nverts_total = len(v)
    while nverts_total > n_verts_desired:
        e = heapq.heappop(queue)
        r, c = resolve(e.r), resolve(e.c)
        if r == c:
            continue
        destroy_c, destroy_r, cost, Qsum = collapse_cost(r, c)
        if cost > e.cost:
            heapq.heappush(queue, Entry(cost, r, c))
            continue
        to_destroy, to_keep = (c, r) if destroy_c < destroy_r else (r, c)
        faces[faces == to_destroy] = to_keep
        alias[to_destroy] = to_keep
        Qv[r] = Qsum
        Qv[c] = Qsum
        degenerate = (faces[:, 0] == faces[:, 1]) | (faces[:, 1] == faces[:, 2]) | (faces[:, 2] == faces[:, 0])
        faces = faces[~degenerate].copy()
        nverts_total = len(np.unique(faces.flatten()))
    return _get_sparse_transform(faces, len(mesh.v))


def generate_transform_matrices(mesh, factors):
    """M: the meshes (mesh, then one per factor, each 1/factor of the previous); A: their adjacency matrices (coo);
    D: the down-sampling transforms between consecutive ones (coo)."""
    M, A, D = [mesh], [get_vert_connectivity(mesh.v, mesh.f).tocoo()], []
    for factor in (1.0 / x for x in factors):
        ds_f, ds_D = qslim_decimator_transformer(M[-1], factor=factor)
        D.append(ds_D.tocoo())
        new_mesh = Mesh(v=ds_D.dot(M[-1].v), f=ds_f)
        M.append(new_mesh)
        A.append(get_vert_connectivity(new_mesh.v, new_mesh.f).tocoo())
    return M, A, D
