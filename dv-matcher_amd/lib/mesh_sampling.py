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


def get_vert_connectivity(mesh_v, mesh_f):
    """Sparse #verts x #verts matrix with a nonzero wherever two vertices share an edge."""
    n = len(mesh_v)
    vpv = sp.csc_matrix((n, n))
    for i in range(3):
        IS, JS = mesh_f[:, i], mesh_f[:, (i + 1) % 3]
        mtx = sp.csc_matrix((np.ones(len(IS)), (IS.flatten(), JS.flatten())), shape=(n, n))
        vpv = vpv + mtx + mtx.T
    return vpv


def get_vertices_per_edge(mesh_v, mesh_f):
    """E x 2 array of vertex pairs, every edge once (lower index first)."""
    vc = sp.coo_matrix(get_vert_connectivity(mesh_v, mesh_f))
    result = np.stack([vc.row, vc.col], axis=1)
    return result[result[:, 0] < result[:, 1]]


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
    verts_left = np.unique(faces.flatten())
    IS = np.arange(len(verts_left))
    mp = np.arange(0, np.max(faces.flatten()) + 1)
    mp[verts_left] = IS
    new_faces = mp[faces.copy().flatten()].reshape((-1, 3))
    mtx = sp.csc_matrix((np.ones(len(verts_left)), (IS, verts_left)), shape=(len(verts_left), num_original_verts))
    return new_faces, mtx


def qslim_decimator_transformer(mesh, factor=None, n_verts_desired=None):
    """-> (new_faces (F',3), mtx (V' x V) selecting the surviving vertices)."""
    if factor is None and n_verts_desired is None:
        raise Exception('Need either factor or n_verts_desired.')
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

    one = np.array([1]).reshape(-1, 1)

    def collapse_cost(r, c):
        Qsum = Qv[r, :, :] + Qv[c, :, :]
        p1 = np.vstack((v[r].reshape(-1, 1), one))
        p2 = np.vstack((v[c].reshape(-1, 1), one))
        destroy_c = float(p1.T.dot(Qsum).dot(p1)[0, 0])
        destroy_r = float(p2.T.dot(Qsum).dot(p2)[0, 0])
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
