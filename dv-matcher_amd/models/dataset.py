"""Shape-pair datasets feeding the correspondence path.

Mirrors the public surface of the reference's `models/dataset.py` / `models/dataset_partial.py`:
`load_off_point_cloud` (16-27), `cal_geo` (49-54), `Dataset` (56-340), `testDataset` (343-569),
`shape_to_device` (572-583) and the partial-view `__getitem__` of dataset_partial.py:226-284 --
same constructor arguments, the same cache file names and cache tuples
(`(verts_list, used_shapes, fps_list, dist_list)` for training, `(verts_list, used_shapes, fps_list)`
with the `_test.pt` suffix for inference), the same per-dataset truncation of the FPS order
(1024 / 5000 / 4995) and the same `{"shape1": {...}, "shape2": {...}}` items, so a cache written by
the reference loads here and vice versa.

What differs, and why:
  * the farthest-point ordering of every shape runs on the GPU through the C ABI (`dvm_fps_f32`,
    include/dvm.h); building a cache therefore needs a HIP device, reading one does not;
  * `cal_geo` -- the reference calls potpourri3d's `PointCloudHeatSolver` (a C++ dependency absent
    from the reference tree and from this image).  It is replaced by exact shortest paths on the
    symmetric k-nearest-neighbour graph of the cloud (or on the mesh edges when faces are given),
    i.e. what the reference's own eval/geo_mat.py:15-41 uses for its geodesic matrices -- computed by
    a HIP kernel (dvm_graph_geodesics_f64: 5000 points in ~20 ms instead of 19 s of scipy Dijkstra,
    same values); scipy is used when no GPU is present.  The two
    approximate the same intrinsic distance; they are not bit-comparable ("parity unpinned" for
    this function only -- a cache built by the reference can be used when the heat-method values
    themselves are wanted);
  * visual features are read (`feat/<shape>.mat`, key 'feat') but not produced: the FeatUp/DINOv2
    projection is §8f-1, and asking for it raises instead of substituting something else.
"""
import os
import random
from itertools import permutations
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset as _TorchDataset

# dataset names whose .off files are bare point clouds (models/dataset.py:183)
_POINT_CLOUD_SETS = ("spleen", "spleen_test", "se-ornet-tosca", "step2_posed_templates_5k_off",
                     "clothes_align_5k_off", "clothes_heavy_off")
_FULL_RES_SETS = ("step2_posed_templates_5k_off", "clothes_align_5k_off", "clothes_heavy_off")
_ANIMALS = ("cat", "centaur", "dog", "gorilla", "horse")


def _is_point_cloud_set(name):
    return name in _POINT_CLOUD_SETS or "clothes_shape" in name


def _is_full_res_set(name):
    return name in _FULL_RES_SETS or "clothes_shape" in name


# ------------------------------------------------------------------ OFF files
def _off_tokens(file_path):
    with open(file_path, "r") as f:
        head = f.readline().split()
        if not head or not head[0].startswith("OFF"):
            raise ValueError("%s: not an OFF file" % file_path)
        rest = head[0][3:].split() + head[1:]          # tolerate "OFF nv nf ne" on one line
        body = f.read()
    lines = [ln.split("#", 1)[0].split() for ln in body.splitlines()]
    lines = [ln for ln in lines if ln]
    if len(rest) >= 2:
        counts, lines = rest, lines
    else:
        counts, lines = lines[0], lines[1:]
    return int(counts[0]), int(counts[1]), lines


def load_off_point_cloud(file_path):
    """Vertex block of an OFF file as a list of [x, y, z] (reference models/dataset.py:16-27)."""
    nv, _, lines = _off_tokens(file_path)
    if len(lines) < nv:
        raise ValueError("%s: %d vertices announced, %d lines present" % (file_path, nv, len(lines)))
    return [[float(t[0]), float(t[1]), float(t[2])] for t in lines[:nv]]


def read_mesh(file_path):
    """(verts (V,3) float64, faces (F,3) int64) of a triangle OFF file -- the role of
    potpourri3d.read_mesh at models/dataset.py:186.  Polygons are fan-triangulated."""
    nv, nf, lines = _off_tokens(file_path)
    verts = np.asarray([[float(t[0]), float(t[1]), float(t[2])] for t in lines[:nv]], dtype=np.float64).reshape(nv, 3)
    faces = []
    for t in lines[nv:nv + nf]:
        n = int(t[0])
        ids = [int(x) for x in t[1:1 + n]]
        faces.extend([ids[0], ids[i], ids[i + 1]] for i in range(1, n - 1))
    return verts, np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def read_file(file_path):
    """One integer per line (partial-view index files, models/dataset_partial.py:244-247)."""
    with open(file_path, "r") as f:
        return [int(float(ln)) for ln in f.read().split()]


# ------------------------------------------------------------------ geodesics
def cal_geo(V, faces=None, k=8):
    """(N,N) float32 matrix of intrinsic distances of a point cloud / mesh (see module docstring)."""
    from scipy.sparse.csgraph import shortest_path
    from scipy.spatial import cKDTree
    V = np.asarray(V, dtype=np.float64).reshape(-1, 3)
    n = V.shape[0]
    if n == 0:
        return torch.zeros(0, 0)
    if faces is not None and len(faces):
        F = np.asarray(faces, dtype=np.int64)
        src = np.concatenate([F[:, 0], F[:, 1], F[:, 2]])
        dst = np.concatenate([F[:, 1], F[:, 2], F[:, 0]])
    else:
        kk = min(k + 1, n)
        _, nbr = cKDTree(V).query(V, kk)
        nbr = nbr.reshape(n, kk)
        src = np.repeat(np.arange(n), kk)
        dst = nbr.reshape(-1)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = np.linalg.norm(V[src] - V[dst], axis=1)
    g = _dedup_min(src, dst, w, n)                 # one undirected edge per pair, shortest copy kept
    if torch.cuda.is_available() and n * 8 <= 150 * 1024:
        d = _shortest_paths_gpu(g, n)              # one workgroup per source (dvm_graph_geodesics_f64): ~1000x scipy at N = 5000
    else:
        d = shortest_path(g, method="D", directed=False)
    far = ~np.isfinite(d)
    if far.any():                                   # disconnected pieces: bridge with the straight-line distance
        eu = np.linalg.norm(V[:, None, :] - V[None, :, :], axis=-1) if n <= 4096 else None
        if eu is None:
            rows, cols = np.nonzero(far)
            d[rows, cols] = np.linalg.norm(V[rows] - V[cols], axis=1)
        else:
            d[far] = eu[far]
    return torch.from_numpy(d.astype(np.float32))


def _shortest_paths_gpu(g, n):
    """csr graph -> (n,n) float64 numpy matrix of shortest path lengths, computed on the GPU."""
    from dvm import ops
    g = g.tocsr()
    deg = np.diff(g.indptr)
    K = max(int(deg.max()), 1)
    nbr = np.full((n, K), -1, dtype=np.int32)
    wts = np.zeros((n, K), dtype=np.float64)
    col = np.arange(len(g.indices)) - np.repeat(g.indptr[:-1], deg)
    row = np.repeat(np.arange(n), deg)
    nbr[row, col] = g.indices
    wts[row, col] = g.data
    dev = torch.device("cuda", torch.cuda.current_device())
    return ops.graph_geodesics(torch.from_numpy(nbr).to(dev), torch.from_numpy(wts).to(dev)).cpu().numpy()


def _dedup_min(src, dst, w, n):
    from scipy.sparse import csr_matrix
    a, b = np.minimum(src, dst), np.maximum(src, dst)
    key = a * n + b
    order = np.lexsort((w, key))
    key, a, b, w = key[order], a[order], b[order], w[order]
    first = np.ones(len(key), dtype=bool)
    first[1:] = key[1:] != key[:-1]
    a, b, w = a[first], b[first], np.maximum(w[first], 1e-12)
    return csr_matrix((np.concatenate([w, w]), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n))


# ------------------------------------------------------------------ FPS ordering
def farthest_point_sample(verts, npoint, start=None, device=None):
    """FPS order of `verts` (N,3) -> (npoint,) int64 on the CPU; the role of misc/utils.py:460-475.
    Runs the HIP kernel; there is no host implementation in the product."""
    if not torch.cuda.is_available():
        raise RuntimeError("farthest_point_sample needs a HIP device (dvm_fps_f32); load a cache instead")
    from lib.deformation_graph_point import farthest_point_sample as _fps
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    x = torch.as_tensor(verts, dtype=torch.float32).to(dev).unsqueeze(0)
    if start is None:
        start = torch.randint(0, x.shape[1], (1,), dtype=torch.long)
    return _fps(x, npoint, torch.as_tensor(start).reshape(1)).squeeze(0).cpu()


def _fps_keep(name):
    if name in ("spleen", "spleen_test", "se-ornet-tosca"):
        return 1024
    if name == "fourleg":
        return 5000
    return 4995


def _pairs(name, used_shapes):
    if name == "amass_ssft":                        # models/dataset.py:120-125
        is_animal = [any(a in s for a in _ANIMALS) for s in used_shapes]
        animals = [i for i, f in enumerate(is_animal) if f]
        others = [i for i, f in enumerate(is_animal) if not f]
        return list(permutations(animals, 2)) + list(permutations(others, 2))
    return list(permutations(range(len(used_shapes)), 2))


_TOSCA_TEST_GROUPS = ((0, 11), (11, 17), (17, 26), (26, 30), (30, 38), (38, 41))   # models/dataset.py:407-413


class _ShapeCollection(_TorchDataset):
    """Loading, caching and FPS ordering shared by the training and inference datasets."""
    _with_dist = True
    _cache_suffix = ""

    def __init__(self, root_dir, name="scape-remeshed", k_eig=128, n_fmap=30, n_cfmap=20, with_wks=None, with_sym=False,
                 use_cache=True, op_cache_dir=None, train=True, with_dino=False, feat_mat=False):
        self.with_dino, self.feat_mat = with_dino, feat_mat
        self.k_eig, self.n_fmap, self.n_cfmap = k_eig, n_fmap, n_cfmap
        self.root_dir = self.cache_dir = root_dir
        self.op_cache_dir, self.with_sym, self.name = op_cache_dir, with_sym, name
        self.device = "cuda:0"
        self.feat_list = []
        if with_sym:
            raise NotImplementedError("with_sym needs corres/*.sym.vts files, which the reference does not load either")
        split = "train" if train else "test"
        wks_suf = "" if with_wks is None else "wks_"
        cache = os.path.join(self.cache_dir, "cache_%s_%s%s%s.pt" % (name, wks_suf, split, self._cache_suffix))
        self.cache_path = cache
        if use_cache and os.path.exists(cache):
            blob = torch.load(cache, weights_only=False)
            if self._with_dist:
                self.verts_list, self.used_shapes, self.fps_list, self.dist_list = blob
            else:
                self.verts_list, self.used_shapes, self.fps_list = blob[:3]
            self.combinations = self._enumerate()
            return
        shapes_dir = Path(root_dir) / ("shapes_" + split)
        self.used_shapes = sorted(x.stem for x in shapes_dir.iterdir() if "DS_" not in x.stem)
        self.verts_list, self.fps_list, self.dist_list = [], [], []
        for shape in self.used_shapes:
            path = str(shapes_dir / (shape + ".off"))
            if _is_point_cloud_set(name):
                verts, faces = np.asarray(load_off_point_cloud(path), dtype=np.float64), None
            else:
                verts, faces = read_mesh(path)
            if self._with_dist:
                self.dist_list.append(cal_geo(verts, faces))
            verts = torch.from_numpy(np.ascontiguousarray(verts)).float()
            fps = farthest_point_sample(verts, verts.shape[0])
            self.verts_list.append(verts)
            self.fps_list.append(fps[:_fps_keep(name)])
        self.combinations = self._enumerate()
        if use_cache:
            blob = (self.verts_list, self.used_shapes, self.fps_list) + ((self.dist_list,) if self._with_dist else ())
            torch.save(blob, cache)

    def _enumerate(self):
        return _pairs(self.name, self.used_shapes)

    def __len__(self):
        return len(self.combinations)

    def _feat(self, i, order=None):
        if not self.with_dino:
            return torch.tensor([])
        if self.feat_mat:
            import scipy.io as sio
            path = os.path.join(self.root_dir, "feat", self.used_shapes[i] + ".mat")
            if not os.path.exists(path):
                raise FileNotFoundError("%s: visual features are read, not produced, here (SURVEY.md §8f-1)" % path)
            feat = torch.tensor(sio.loadmat(path)["feat"], dtype=torch.float32)
        elif self.feat_list:
            feat = self.feat_list[i]
        else:
            raise NotImplementedError("with_dino without feat_mat needs the FeatUp/DINOv2 projection (SURVEY.md §8f-1)")
        return feat if order is None else feat[order]


class Dataset(_ShapeCollection):
    """Training pairs: every ordered pair of shapes, each delivered in FPS order with the matching
    sub-block of its geodesic matrix (reference models/dataset.py:233-340)."""

    def _shape(self, i):
        if self.with_dino and _is_full_res_set(self.name):           # models/dataset.py:240-268: full resolution
            return {"xyz": self.verts_list[i], "feat": self._feat(i), "name": self.used_shapes[i], "dist": self.dist_list[i]}
        fps = self.fps_list[i]
        return {"xyz": self.verts_list[i][fps], "feat": self._feat(i, fps), "name": self.used_shapes[i],
                "dist": self.dist_list[i][fps][:, fps]}

    def __getitem__(self, idx):
        idx1, idx2 = self.combinations[idx]
        return {"shape1": self._shape(idx1), "shape2": self._shape(idx2)}


class testDataset(_ShapeCollection):
    """Inference pairs: full-resolution vertices, no geodesics (reference models/dataset.py:343-569)."""
    _with_dist = False
    _cache_suffix = "_test"

    def _enumerate(self):
        if self.name == "tosca":
            return [p for lo, hi in _TOSCA_TEST_GROUPS for p in permutations(range(lo, hi), 2)]
        return list(permutations(range(len(self.used_shapes)), 2))

    def _shape(self, i):
        return {"xyz": self.verts_list[i], "feat": self._feat(i), "name": self.used_shapes[i], "dist": torch.tensor([])}

    def __getitem__(self, idx):
        idx1, idx2 = self.combinations[idx]
        return {"shape1": self._shape(idx1), "shape2": self._shape(idx2)}


class PartialDataset(Dataset):
    """Full source shape against a partial view of the target (reference models/dataset_partial.py:226-284):
    one of twelve precomputed view index files `index_partial/index_<shape>_view_<v>.txt` with more than
    `n_partial` points is drawn, re-ordered by FPS and cut to `n_partial` (2200)."""
    n_partial = 2200
    resamples_coordinates = True   # a target's coordinates are a fresh random view on every access: never key a cache by its name

    def __getitem__(self, idx):
        idx1, idx2 = self.combinations[idx]
        shape1 = self._shape(idx1)
        if self.name in ("shrec16_cuts", "shrec16_holes"):
            fps2 = self.fps_list[idx2][:1024]
            shape1 = {"xyz": self.verts_list[idx1], "dist": self.dist_list[idx1], "name": self.used_shapes[idx1]}
            shape2 = {"xyz": self.verts_list[idx2][fps2], "dist": self.dist_list[idx2][fps2][:, fps2], "name": self.used_shapes[idx2]}
            return {"shape1": shape1, "shape2": shape2}
        views = list(range(1, 13))
        random.shuffle(views)
        for view in views:
            path = os.path.join(self.root_dir, "index_partial", "index_%s_view_%d.txt" % (self.used_shapes[idx2], view))
            part = torch.tensor(read_file(path)).long().reshape(-1)
            if part.shape[0] > self.n_partial:
                break
        else:
            raise ValueError("%s: no view with more than %d points" % (self.used_shapes[idx2], self.n_partial))
        verts2 = self.verts_list[idx2][part]
        fps2 = farthest_point_sample(verts2, verts2.shape[0])[:self.n_partial]
        dist2 = self.dist_list[idx2][part][:, part]
        shape1 = {k: shape1[k] for k in ("xyz", "dist", "name")}
        shape2 = {"xyz": verts2[fps2], "dist": dist2[fps2][:, fps2], "name": self.used_shapes[idx2]}
        return {"shape1": shape1, "shape2": shape2}


def shape_to_device(dict_shape, device):
    """Move xyz / feat / dist of both shapes to `device` (reference models/dataset.py:572-583)."""
    for k, v in dict_shape.items():
        if "shape" in k:
            for name in ("xyz", "feat", "dist"):
                if v.get(name) is not None:
                    v[name] = v[name].to(device)
        else:
            dict_shape[k] = v.to(device)
    return dict_shape
