"""Embedded-deformation graph on a point cloud — MI355X path.

Drop-in for the reference's `lib.deformation_graph_point` (same class name, attributes and call
signatures: reference lib/deformation_graph_point.py:71-261).  The graph is built and applied
by the HIP kernels of dv-matcher_amd/csrc/dvm_graph.hip through the C ABI (include/dvm.h);
nothing is copied to the host and scipy's KDTree is not used.
"""
import numpy as np
import torch
import torch.nn as nn

from dvm import ops


def farthest_point_sample(xyz, npoint, start=None):
    """xyz (B,N,3) -> centroids (B,npoint) int64.  The reference draws the first centroid with
    torch.randint(0, N, (B,)) (lib/deformation_graph_point.py:24); pass `start` to fix it."""
    B, N, _ = xyz.shape
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long)
    start = torch.as_tensor(start).to(xyz.device)
    return ops.fps(xyz, npoint, start).long()


def compute_edge_lengths(vertices, faces):
    """(F,3) lengths of the edges leaving each face corner (reference lib/deformation_graph_point.py:325-341)."""
    face_vertices = vertices[faces]
    edge_vectors = torch.roll(face_vertices, -1, dims=1) - face_vertices
    return torch.norm(edge_vectors, dim=2)


class DeformationGraph_geod(nn.Module):
    """Same public surface as the reference class; the graph tensors live on the device.

    Attributes after construct_graph_euclidean (names as in the reference):
      nodes_idx (Nn,) numpy int64, nodes (Nn,3), one_ring_neigh (Nn,9) numpy int64,
      influence_nodes_idx (N,3) int64, dists (N,3), weights (N,3), sigma, max_neigh_num = 9.
    """

    def __init__(self, radius=0.1, k=3, sampling_strategy='qslim'):
        super().__init__()
        self.radius = radius
        self.k = k
        self.max_neigh_num = 18
        self.sampling_strategy = sampling_strategy
        self.one_ring_neigh = []
        self.nodes_idx = None
        self.weights = None
        self.influence_nodes_idx = []
        self.dists = []
        self._g = None  # device-side graph (int32 / fp32 tensors with a leading batch dim of 1)

    @classmethod
    def from_batch(cls, g, b, vertices):
        """View element `b` of a batched dvm.ops.dg_build result as a graph object."""
        self = cls()
        self._g = {k: v[b:b + 1] for k, v in g.items()}
        self._publish(vertices)
        return self

    def _publish(self, vertices):
        g = self._g
        self.max_neigh_num = 9
        self.nodes_idx = g["nodes_idx"][0].cpu().numpy().astype(np.int64)
        self.nodes = vertices[g["nodes_idx"][0].long()]
        self.one_ring_neigh = g["one_ring"][0].cpu().numpy().astype(np.int64)
        self.influence_nodes_idx = g["infl_idx"][0].long()
        self.dists = g["dists"][0]
        self.weights = g["weights"][0]
        self.sigma = g["sigma"][0]
        self.pre_idx = self.influence_nodes_idx[:, :1]

    def construct_graph_euclidean(self, vertices=None, geod=None, device=None, start=None):
        """vertices (N,3) tensor.  `geod` (the N x N cdist the reference passes in as a host array) is
        not needed: the kernels recompute the same matmul-form distances on the device.  `device`
        selects the HIP device when `vertices` is a CPU tensor (as in the reference's call)."""
        if self.sampling_strategy != 'qslim':
            raise NotImplementedError("only the 'qslim' (FPS) sampling of the reference is implemented")
        v = torch.as_tensor(vertices, dtype=torch.float32)
        if not v.is_cuda:
            v = v.to(device if device is not None else "cuda")
        N = v.shape[0]
        if start is None:  # same RNG draw as the reference's FPS (shape (1,), CPU generator)
            start = torch.randint(0, N, (1,), dtype=torch.long)
        start = torch.as_tensor(start).reshape(1).to(v.device)
        self._g = ops.dg_build(v[None], start)
        self._publish(v)

    def construct_graph(self, vertices=None, faces=None, geod=None, device=None):
        """Mesh-mode graph (reference lib/deformation_graph_point.py:203-231, used by deform.py:167-216): the nodes are
        the vertices that survive a QSlim decimation to half the vertex count, a node's ring is its neighbours in the
        decimated mesh (padded to max_neigh_num = 18 with the node itself), every vertex is skinned to its 3 geodesically
        nearest nodes (`geod`: the (V,V) geodesic matrix the caller supplies) with Gaussian weights of width
        20 x the decimated mesh's mean edge length.  The build is host code, as in the reference (lib/mesh_sampling.py);
        the graph then lives on the device and forward() runs on the HIP kernels."""
        if self.sampling_strategy != 'qslim':
            raise NotImplementedError("only the 'qslim' sampling of the reference is implemented")
        from lib.mesh_sampling import Mesh, generate_transform_matrices
        v_host = torch.as_tensor(vertices).detach().cpu().double().numpy()
        faces = np.asarray(faces, dtype=np.int64)
        self.faces = faces
        self.max_neigh_num = 18
        M, A, D = generate_transform_matrices(Mesh(v=v_host, f=faces), [2])
        self.graph = M[1]
        self.nodes_idx = D[0].nonzero()[1]
        adj = A[1].toarray()
        ring = []
        for i in range(adj.shape[0]):
            nb = adj[i].nonzero()[0].tolist()
            if len(nb) > self.max_neigh_num:
                raise ValueError("node %d has %d neighbours (max_neigh_num = %d)" % (i, len(nb), self.max_neigh_num))
            ring.append(nb + [i] * (self.max_neigh_num - len(nb)))
        self.one_ring_neigh = torch.tensor(ring)
        geod_mat = torch.from_numpy(-np.asarray(geod)[self.nodes_idx]).transpose(1, 0)
        dists, infl = geod_mat.topk(self.k, dim=-1)
        self.dists = -dists
        self.pre_idx = geod_mat.topk(1, dim=-1)[1]
        gv, gf = torch.tensor(self.graph.v), torch.tensor(self.graph.f.astype(np.int64))
        self.sigma = 20 * torch.mean(compute_edge_lengths(gv, gf))
        w = torch.exp(-(self.dists ** 2) / (2 * self.sigma * self.sigma))
        self.weights = (w / w.sum(1).reshape(-1, 1)).float()
        dev = torch.device(device) if device is not None else (vertices.device if torch.is_tensor(vertices) and vertices.is_cuda
                                                                 else torch.device("cuda"))
        self.influence_nodes_idx = infl.to(dev)
        self.weights = self.weights.to(dev)
        self._g = dict(nodes_idx=torch.from_numpy(np.asarray(self.nodes_idx)).int().to(dev)[None],
                       one_ring=self.one_ring_neigh.int().to(dev)[None], infl_idx=infl.int().to(dev)[None],
                       weights=self.weights[None], dists=self.dists.float().to(dev)[None])
        self._mesh_mode = True

    def forward(self, vertices, opt_d_rotations, opt_d_translations):
        """vertices (N,3), rotations (1,Nn,3,3), translations (1,Nn,3) -> (warped (1,N,3), arap, sr)."""
        if self._g is None:
            raise RuntimeError("construct_graph_euclidean() has not been called")
        fn = ops.dg_warp_arap_graph if getattr(self, "_mesh_mode", False) else ops.dg_warp_arap
        warped, arap, sr = fn(vertices[None], self._g, opt_d_rotations, opt_d_translations)
        return warped, arap[0], sr[0]
