"""LG-Net (`Uni3FC`) and `Deformer` — MI355X path behind the reference's module API.

Mirrors the public names, constructor arguments, forward signatures and state_dict keys of the
reference's `models/model.py` (Uni3FC 480-761, N2PAttention 325-395, SA_Layer 97-123, Deformer
454-478, MLP 433-452, knn_new 267-278, index_points 255-264) so that the reference's train.py /
test.py / deform.py call sequence runs unchanged and its checkpoints load with
`load_state_dict`.  Hot operators go through the C ABI (include/dvm.h, dvm.ops); the dense
1x1 convolutions / BatchNorm stay on PyTorch-ROCm as BASELINE.json's north_star prescribes.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from dvm import ops
from dvm import nn_ops


# ------------------------------------------------------------------ index helpers
def index_points(points, idx):
    """points (B,N,C), idx (B,N,K) -> (B,N,K,C)."""
    B, N, K = idx.shape
    flat = idx.reshape(B, N * K, 1).expand(-1, -1, points.shape[-1]).long()
    return torch.gather(points, 1, flat).view(B, N, K, -1)


def index_points_idx(points, idx):
    """points (B,N,C), idx (B,S) -> (B,S,C)."""
    return torch.gather(points, 1, idx.long().unsqueeze(-1).expand(-1, -1, points.shape[-1]))


def knn_new(a, b, k):
    """Feature-space kNN: the k largest of -|a|^2 + 2ab - |b|^2, nearest first (int64, like torch.topk)."""
    return ops.knn_neg(a, b, k).long()


def farthest_point_sample(xyz, npoint):
    from lib.deformation_graph_point import farthest_point_sample as _fps
    return _fps(xyz.unsqueeze(0), npoint)


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) with the reference's parameter names/shapes (weight (Cout,Cin,1)), evaluated as the GEMM
    it is on the library's own fp32 matrix-core kernel (dvm_linear_f32: the reference's single-thread fma chain, bit for
    bit — no BLAS / convolution library in the forward; the two backward GEMMs are library calls)."""

    def forward(self, x):
        return nn_ops.conv1x1(x, self.weight, self.bias)


def _folded(owner, tag, sources, build):
    """Inference-time derived parameters (BatchNorm folded into a conv, stacked q/k/v weights ...) are rebuilt only when
    one of their source tensors was written (optimizer step, load_state_dict) or moved: ~150 tiny launches per forward
    otherwise.  Kept outside the module's parameters/buffers, so state_dict is untouched."""
    key = tuple((t.data_ptr(), t._version) for t in sources if t is not None)
    cache = owner.__dict__.setdefault("_dvm_folded", {})
    hit = cache.get(tag)
    if hit is None or hit[0] != key:
        hit = (key, build())
        cache[tag] = hit
    return hit[1]


def invalidate_folded(module):
    """Drop every derived inference-time tensor under `module`.  The caches notice in-place writes through the tensors
    themselves (optimizer steps, load_state_dict, copy_) by their version counters; a write that bypasses them — through
    `p.data`, a raw pointer, another process — does not bump a version: call this after such a write."""
    for m in module.modules():
        m.__dict__.pop("_dvm_folded", None)
        m.__dict__.pop("_pos_cache", None)


def _bn_sources(bn):
    # num_batches_tracked: the fused training kernel writes the running statistics through raw pointers (no version
    # bump on those two tensors); the batch counter is incremented by a torch op on every such update
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked)


def _bn_affine(bn):
    """Eval-mode BatchNorm as ATen's CPU kernel evaluates it: y = fma(x, alpha, beta) with alpha = w / sqrt(var + eps)
    (correctly rounded fp32 sqrt and divide), beta = fma(-mean, alpha, b).  The 4 x C numbers are folded on the host
    (numpy's fp32 sqrt / divide are IEEE; device rsqrt is not) once per parameter version."""
    def build():
        w, b, m, v = (t.detach().cpu().numpy().astype(np.float32) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        alpha = (np.float32(1) / np.sqrt(v + np.float32(bn.eps))) * w
        beta = (b.astype(np.float64) - m.astype(np.float64) * alpha.astype(np.float64)).astype(np.float32)
        dev = bn.weight.device
        return torch.from_numpy(alpha).to(dev), torch.from_numpy(beta).to(dev)
    return _folded(bn, "affine", _bn_sources(bn), build)


def _conv_bn_pm(conv, bn, xt, slope=0.2, res=None, prefix=None):
    """Inference, point-major: 1x1 conv -> [+ res] -> eval-mode BatchNorm -> (Leaky)ReLU in ONE launch (dvm_linear_f32:
    the reference's single-thread fp32 chain on the matrix cores, epilogue fused).  xt (B,N,Cin) -> (B,N,Cout); with
    `prefix` (B,1,Cg) the conv runs over cat((prefix broadcast over the points, xt), channels) without building it."""
    return ops.linear(xt, conv.weight, bias=conv.bias, res=res, bn=_bn_affine(bn), slope=slope, prefix=prefix)


# ------------------------------------------------------------------ attention blocks
class SA_Layer(nn.Module):
    """Offset self-attention with shared q/k weights and column re-normalisation."""

    def __init__(self, channels):
        super().__init__()
        self.bn1 = nn.BatchNorm1d(64)  # unused in forward (kept: the reference registers it)
        self.conv1 = nn.Sequential(nn.Conv1d(128, 64, kernel_size=1, bias=False), self.bn1,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.q_conv = nn.Conv1d(channels, channels // 4, 1, bias=False)
        self.k_conv = nn.Conv1d(channels, channels // 4, 1, bias=False)
        self.q_conv.weight = self.k_conv.weight  # tied
        self.v_conv = PointwiseConv1d(channels, channels, 1)
        self.trans_conv = PointwiseConv1d(channels, channels, 1)
        self.after_norm = nn.BatchNorm1d(channels)
        self.act = nn.ReLU()
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        x_r = nn_ops.sa_attention(x, self.k_conv.weight, self.v_conv.weight, self.v_conv.bias)
        return x + nn_ops.bn_act(self.after_norm, self.trans_conv(x - x_r), slope=0.0)

    def train_pm(self, xt):
        """Autograd path on point-major activations (B,N,64)."""
        x_r = nn_ops.sa_attention_pm(xt, self.k_conv.weight, self.v_conv.weight, self.v_conv.bias)
        t = nn_ops.linear_pm(xt - x_r, self.trans_conv.weight, self.trans_conv.bias)
        return xt + nn_ops.bn_act_pm(self.after_norm, t, slope=0.0)

    def infer_pm(self, xt):
        """Inference on point-major activations (B,N,64): no transposes, BatchNorm folded into trans_conv."""
        p = ops.linear(xt, self.k_conv.weight)
        v = ops.linear(xt, self.v_conv.weight, bias=self.v_conv.bias)
        x_r = ops.sa_attention_pm(p, v)
        return xt + _conv_bn_pm(self.trans_conv, self.after_norm, xt - x_r, slope=0.0)


class _N2P(nn.Module):
    """Neighbour-to-point attention over the K feature-space nearest neighbours (4 heads)."""

    def __init__(self, k, C):
        super().__init__()
        self.heads = 4
        self.K = k
        self.group_type = 'diff'
        self.q_conv = nn.Conv2d(C, C, 1, bias=False)
        self.k_conv = nn.Conv2d(C, C, 1, bias=False)
        self.v_conv = nn.Conv2d(C, C, 1, bias=False)
        self.softmax = nn.Softmax(dim=-1)
        self.ff = nn.Sequential(PointwiseConv1d(C, 4 * C, 1, bias=False), nn.LeakyReLU(0.2),
                                PointwiseConv1d(4 * C, C, 1, bias=False))
        self.bn1 = nn.BatchNorm1d(C)
        self.bn2 = nn.BatchNorm1d(C)

    def forward(self, x):
        att = nn_ops.n2p_attention(x, self.K, self.q_conv.weight, self.k_conv.weight, self.v_conv.weight, self.heads)
        x = nn_ops.bn_act(self.bn1, x, att)
        return nn_ops.bn_act(self.bn2, x, self.ff(x))

    def train_pm(self, xt):
        """Autograd path on point-major activations (B,N,C)."""
        att = nn_ops.n2p_attention_pm(xt, self.K, self.q_conv.weight, self.k_conv.weight, self.v_conv.weight, self.heads)
        xt = nn_ops.bn_act_pm(self.bn1, xt, att)
        h = nn_ops.linear_pm(xt, self.ff[0].weight, slope=self.ff[1].negative_slope)
        return nn_ops.bn_act_pm(self.bn2, xt, nn_ops.linear_pm(h, self.ff[2].weight))

    def infer_pm(self, xt):
        """Inference on point-major activations (B,N,C): q/k/v by ONE GEMM, gather-attention on the HIP kernel, the two
        eval-mode BatchNorms as per-channel affines."""
        C = xt.shape[-1]
        xt = xt.contiguous()
        idx = ops.knn_neg(xt, xt, self.K)
        w = _folded(self, "qkv", (self.q_conv.weight, self.k_conv.weight, self.v_conv.weight), lambda: torch.cat(
            [self.q_conv.weight.reshape(C, C), self.k_conv.weight.reshape(C, C), self.v_conv.weight.reshape(C, C)], 0))
        att = ops.n2p_core_fwd(ops.linear(xt, w), idx, self.heads)[0]
        s1, t1 = _bn_affine(self.bn1)
        xt = torch.addcmul(t1, xt + att, s1)
        h = ops.linear(xt, self.ff[0].weight, slope=self.ff[1].negative_slope)
        return _conv_bn_pm(self.ff[2], self.bn2, h, slope=1.0, res=xt)     # bn2(x + ff(x)), one launch

    @staticmethod
    def split_heads(x, heads):
        B, C, N, K = x.shape
        return x.view(B, heads, C // heads, N, K).permute(0, 1, 3, 4, 2).contiguous()


class N2PAttention(_N2P):
    def __init__(self, k):
        super().__init__(k, 64)


class N2PAttention_DIM(_N2P):
    def __init__(self, k):
        super().__init__(k, 128)


# ------------------------------------------------------------------ Deformer
class MLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dims=[], bias=True, act=nn.ELU()):
        super().__init__()
        self.input_dim, self.output_dim, self.hidden_dims, self.act = input_dim, output_dim, hidden_dims, act
        dims = [input_dim] + list(hidden_dims)
        fc = []
        for a, b in zip(dims[:-1], dims[1:]):
            fc += [nn.Linear(a, b, bias=bias), act]
        fc.append(nn.Linear(dims[-1], output_dim, bias=bias))
        if not hidden_dims:
            fc.append(act)
        self.linear = nn.Sequential(*fc)

    def forward(self, x):
        return self.linear(x)


class Deformer(nn.Module):
    """Per-node [translation(3), rotation-6D(6)] from pooled local features (reference 454-478)."""

    def __init__(self, k):
        super().__init__()
        self.k = k
        self.conv_layer = nn.Conv2d(in_channels=k, out_channels=1, kernel_size=(1, 1))
        self.deformation_decoder_layer = MLP(input_dim=128 * 2 + 3 * 2, output_dim=3 + 6, hidden_dims=[512, 256, 128],
                                             bias=True, act=nn.ELU())

    def weight_list(self, device=None):
        sd = {k: v for k, v in self.state_dict().items()}
        return ops.deformer_weight_list(sd, device if device is not None else next(self.parameters()).device)

    def forward_sparse(self, feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1):
        """Fused path used by the criterion: raw features + kNN indices + sparse Pi (no (B,N,k,128)
        gathers, no dense Pi).  Shapes as in include/dvm.h::dvm_deformer_fwd_f32."""
        return nn_ops.deformer_sparse(self, feat1, feat2, verts1, verts12, idx11, idx22, pi_val, pi_idx, fps1)

    def forward(self, feat1_conv, feat2_conv, verts1, verts12, Pi_12, fps1):
        """Reference signature: feat*_conv (B,N,k,128) gathered features, dense Pi_12 (B,N,M), fps1 (B,Nn)."""
        w = self.conv_layer.weight.view(1, 1, -1, 1)
        g1 = (feat1_conv * w).sum(2) + self.conv_layer.bias
        g2 = (feat2_conv * w).sum(2) + self.conv_layer.bias
        g2 = torch.matmul(Pi_12, g2)
        z = torch.cat([index_points_idx(verts1, fps1), index_points_idx(g1, fps1), index_points_idx(verts12, fps1),
                       index_points_idx(g2, fps1)], dim=-1)
        return nn_ops.mlp(self, z)


# ------------------------------------------------------------------ LG-Net
def rotate_point_cloud_batch_torch(cloud, angle, axis='z'):
    """cloud (B,3,N) -> (B,N,3) rotated about `axis` (reference models/model.py:65-94; the matrix is built from
    float64 cos/sin and cast to float32, so cos(-pi/2) stays 6.1e-17 rather than 0)."""
    c, s = math.cos(angle), math.sin(angle)
    rows = {'z': [[c, -s, 0], [s, c, 0], [0, 0, 1]], 'y': [[c, 0, s], [0, 1, 0], [-s, 0, c]],
            'x': [[1, 0, 0], [0, c, -s], [0, s, c]]}
    if axis not in rows:
        raise ValueError("Axis must be 'x', 'y', or 'z'")
    rot = torch.tensor(rows[axis], dtype=torch.float64).float().to(cloud.device)
    return torch.bmm(cloud.permute(0, 2, 1), rot[None].expand(cloud.shape[0], -1, -1))


class _VisualProjection:
    """Point cloud -> three depth renderings -> image backbone -> per-point visual features (SURVEY §8f-1;
    reference models/model.py:584-710 and Uni3FC_DINO_proj 815-985).  The rendering and the back-projection are
    HIP kernels (dvm_proj2img_f32, dvm_i2p_f32); `upsampler` — FeatUp's DINOv2 in the reference — is any callable
    (3B,3,224,224) -> (3B,C,h,w) and runs on PyTorch-ROCm."""
    img_size = 224

    def proj2img(self, pc):
        """pc (B,N,3) -> (img (B,3,224,224), pc_min (B,1,2), grid_size (B,1,1), (offset_x (B,1), offset_y (B,1)))."""
        B = pc.shape[0]
        img, pc_min, grid, off = ops.proj2img(pc)
        return img, pc_min.view(B, 1, 2), grid.view(B, 1, 1), (off[:, 0:1], off[:, 1:2])

    def I2P(self, pc, f, pc_min, grid_size, offsets):
        """Image features f (B,C,h,w) at every point's pixel of the bicubically 224x224-resized map: (B,N,C)."""
        B = pc.shape[0]
        return ops.i2p(pc, f, pc_min.reshape(B, 2), grid_size.reshape(B), torch.cat(offsets, dim=1).float())

    def views(self, x):
        pts_1 = rotate_point_cloud_batch_torch(x, -math.pi / 2, axis='z')
        pts_2 = torch.cat((pts_1[..., 2:3], pts_1[..., 0:2]), dim=-1)
        pts_3 = torch.cat((pts_1[..., 1:3], pts_1[..., 0:1]), dim=-1)
        return [p.contiguous() for p in (pts_1, pts_2, pts_3)]

    def visual_features(self, x, upsampler):
        """x (B,3,N) -> (B,N,3C): L2-normalised back-projected features of the three views, side by side."""
        if upsampler is None:
            raise ValueError("dino_feat is None: an image backbone (`upsampler`) is needed to produce the visual features")
        B, _, N = x.shape
        with torch.no_grad():
            pts = self.views(x)
            proj = [ops.proj2img(p) for p in pts]
            img_feats = upsampler(torch.cat([pr[0] for pr in proj], dim=0))
            C = img_feats.shape[1]
            out = torch.empty(B, N, 3 * C, dtype=torch.float32, device=x.device)
            for v in range(3):
                _, pc_min, grid, off = proj[v]
                ops.i2p(pts[v], img_feats[v * B:(v + 1) * B], pc_min, grid, off, normalize=True, out=out, col=v * C)
        return out


class Uni3FC_DINO_proj(nn.Module, _VisualProjection):
    """The feature pre-computation module of the dataset build (reference models/model.py:815-985):
    forward(x (B,3,N), upsampler) -> (B,N,3C) visual features (1152 for DINOv2 ViT-S/14 through FeatUp)."""

    def __init__(self):
        super().__init__()
        self.device = 'cuda:0'

    def forward(self, x, upsampler):
        return self.visual_features(x, upsampler)


_side_streams = {}


def _two_branches(x, main, side):
    """LG-Net's local (kNN attention) and global (self-attention) chains share only their input `x`: `main()` runs on the
    current stream, `side()` concurrently on a per-device helper stream, joined before returning (main(), side()).  Neither
    chain fills 256 CUs on its own (their kernels are one workgroup per CU or fewer at 8 x 2048 points); autograd replays
    each chain's backward on the stream its forward ran on."""
    if not x.is_cuda:
        return main(), side()
    cur = torch.cuda.current_stream(x.device)
    helper = _side_streams.get((x.device, cur.cuda_stream))
    if helper is None:
        helper = _side_streams[(x.device, cur.cuda_stream)] = torch.cuda.Stream(x.device)
    helper.wait_stream(cur)
    x.record_stream(helper)
    with torch.cuda.stream(helper):
        b = side()
    a = main()
    cur.wait_stream(helper)
    b.record_stream(cur)
    return a, b


def join_side_streams(device=None):
    _join_branch_streams(device)


def _join_branch_streams(device=None):
    """Make the current stream wait for every helper stream `_two_branches` has used on `device`.  With the fused gradient
    accumulation (dvm.nn_ops.fuse_grad_accumulation) the weight-gradient kernels of the helper-stream chain write straight
    into `p.grad` during backward; autograd orders the caller's stream after them only through its leaf-stream
    bookkeeping.  A training step calls this after backward() and before anything reads the gradients (all-reduce,
    optimizer) so that the ordering is explicit."""
    cur = torch.cuda.current_stream(device)
    for (dev, _), helper in _side_streams.items():
        if dev == cur.device:
            cur.wait_stream(helper)


class Uni3FC(nn.Module, _VisualProjection):
    # Which of this module's equivalent execution paths a call takes (class defaults; an instance attribute overrides).  They are
    # not tuning switches: the per-layer paths are what a call falls back to when the native nodes do not apply (inputs that
    # require grad, SyncBatchNorm, non-fp32 parameters, a kNN tap), and the tests that pin a native node against them set these.
    native_forward = True      # eval forward as ONE dvm_uni3fc_fwd_f32 call (tests/test_gpu_network.py::test_native_forward_is_the_python_path)
    native_train = True        # training forward + backward as ONE autograd node (tests/test_gpu_train_native.py)
    merge_pair_calls = True    # forward_pair: the step's two network calls as one native call with two groups (same file)
    point_major_train = True   # training in the point-major layout; False: the reference's (B,C,N) layout, per layer (tests/test_gpu_train_pm.py)

    def __init__(self, k=40):
        super().__init__()
        self.device = 'cuda:0'
        self.k = k
        self.emb_dims = 512
        self.out = 128
        self.bn = nn.BatchNorm1d(384)
        self.bn0 = nn.BatchNorm1d(64)
        self.bn1 = nn.BatchNorm1d(self.emb_dims)
        self.bn2 = nn.BatchNorm1d(self.emb_dims)
        self.bn3 = nn.BatchNorm1d(128)
        self.bn4 = nn.BatchNorm1d(128)
        self.bn5 = nn.BatchNorm1d(128)
        self.bn6 = nn.BatchNorm1d(128)

        def block(cin, cout, bn):
            return nn.Sequential(PointwiseConv1d(cin, cout, kernel_size=1, bias=False), bn, nn.LeakyReLU(negative_slope=0.2))

        self.conv = block(1152, 384, self.bn)
        self.conv0 = block(384, 64, self.bn0)
        self.conv1 = block(256, self.emb_dims, self.bn1)
        self.conv2 = block(256, self.emb_dims, self.bn2)
        self.conv3 = block(256 + self.emb_dims, 128, self.bn3)
        self.conv4 = block(256 + self.emb_dims, 128, self.bn4)
        self.conv5 = block(256, 128, self.bn5)
        self.conv6 = block(512, 128, self.bn6)
        self.n2p_attention1 = N2PAttention(self.k)
        self.n2p_attention2 = N2PAttention(self.k)
        self.n2p_attention3 = N2PAttention(self.k)
        self.n2p_attention4 = N2PAttention(self.k)
        self.n2p_attention5 = N2PAttention_DIM(self.k)
        self.n2p_attention6 = N2PAttention_DIM(self.k)
        self.n2p_attention7 = N2PAttention_DIM(self.k)
        self.sa1 = SA_Layer(64)
        self.sa2 = SA_Layer(64)
        self.sa3 = SA_Layer(64)
        self.sa4 = SA_Layer(64)

    def pos_encoding_sin_wave(self, coor):
        """64-octave sin/cos encoding of the batch-normalised coordinates: (B,3,N) -> (B,384,N).  With
        `self.sync_minmax = True` under torch.distributed the range is taken over all ranks' shards."""
        return nn_ops.pos_encoding(coor, sync=getattr(self, "sync_minmax", False))

    def _forward_infer(self, x, dino_feat):
        """Eval-mode forward with activations kept point-major (B,N,C): every conv + BatchNorm is one GEMM with bias,
        neighbour rows are contiguous for the gather kernels, nothing is transposed between layers.  (Training keeps the
        reference's (B,C,N) layout: there MIOpen's BatchNorm kernels want it and the step is GEMM/launch-bound.)"""
        B, _, N = x.shape
        if self.native_forward and x.is_cuda and not getattr(self, "sync_minmax", False):
            # the same launches, enqueued by ONE native call (dvm_uni3fc_fwd_f32): no Python between the ~250 kernels
            with torch.no_grad():
                return ops.uni3fc_forward(self._native_table(), x, dino_feat.contiguous(), self.k)
        with torch.no_grad():
            blk = lambda seq, xt, **kw: _conv_bn_pm(seq[0], seq[1], xt, seq[2].negative_slope, **kw)  # noqa: E731
            f = blk(self.conv, dino_feat)
            tmp = blk(self.conv0, f + self.pos_encoding_sin_wave(x).transpose(1, 2))

            def global_branch():
                x1g = self.sa1.infer_pm(tmp)
                x2g = self.sa2.infer_pm(x1g)
                x3g = self.sa3.infer_pm(x2g)
                glo = torch.cat((x1g, x2g, x3g, self.sa4.infer_pm(x3g)), dim=-1)
                return blk(self.conv4, glo, prefix=blk(self.conv2, glo).amax(dim=1, keepdim=True))

            def local_branch():
                x1 = self.n2p_attention1.infer_pm(tmp)
                x2 = self.n2p_attention2.infer_pm(x1)
                x3 = self.n2p_attention3.infer_pm(x2)
                loc = torch.cat((x1, x2, x3, self.n2p_attention4.infer_pm(x3)), dim=-1)
                return blk(self.conv3, loc, prefix=blk(self.conv1, loc).amax(dim=1, keepdim=True))

            y = torch.cat(_two_branches(tmp, local_branch, global_branch), dim=-1)
            y1 = blk(self.conv5, y)
            y2 = self.n2p_attention5.infer_pm(y1)
            y3 = self.n2p_attention6.infer_pm(y2)
            y4 = self.n2p_attention7.infer_pm(y3)
            out = blk(self.conv6, torch.cat((y1, y2, y3, y4), dim=-1))
            return out.contiguous().view(B, N, self.out), tmp

    def _native_table(self):
        """The weight table of dvm_uni3fc_fwd_f32 (order: include/dvm.h), rebuilt when a parameter or a BatchNorm statistic was
        written: conv weights as they are, every eval-mode BatchNorm folded by _bn_affine, q | k | v stacked."""
        convs = (self.conv, self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6)
        sas = (self.sa1, self.sa2, self.sa3, self.sa4)
        n2ps = (self.n2p_attention1, self.n2p_attention2, self.n2p_attention3, self.n2p_attention4, self.n2p_attention5,
                self.n2p_attention6, self.n2p_attention7)

        def build():
            ts = []
            for seq in convs:
                ts += [seq[0].weight.reshape(seq[0].weight.shape[0], -1), *_bn_affine(seq[1])]
            for sa in sas:
                ts += [sa.k_conv.weight.reshape(16, 64), sa.v_conv.weight.reshape(64, 64), sa.v_conv.bias,
                       sa.trans_conv.weight.reshape(64, 64), sa.trans_conv.bias, *_bn_affine(sa.after_norm)]
            for m in n2ps:
                C = m.q_conv.weight.shape[0]
                ts += [torch.cat([m.q_conv.weight.reshape(C, C), m.k_conv.weight.reshape(C, C), m.v_conv.weight.reshape(C, C)], 0),
                       *_bn_affine(m.bn1), m.ff[0].weight.reshape(4 * C, C), m.ff[2].weight.reshape(C, 4 * C), *_bn_affine(m.bn2)]
            return ops.uni3fc_weight_table([t.detach() for t in ts])
        return _folded(self, "native_table", [t for t in list(self.parameters()) + list(self.buffers())], build)

    def _train_table(self):
        """(tensors of dvm_uni3fc_train_fwd_f32's parameter table in include/dvm.h's order, positions of the trainable ones,
        the trainable ones, BatchNorm modules whose batch counter a forward bumps)."""
        convs = (self.conv, self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6)
        sas = (self.sa1, self.sa2, self.sa3, self.sa4)
        n2ps = (self.n2p_attention1, self.n2p_attention2, self.n2p_attention3, self.n2p_attention4, self.n2p_attention5,
                self.n2p_attention6, self.n2p_attention7)
        slots, bns = [], []   # (module, attribute) of every table entry, in table order

        def bn(m):
            bns.append(m)
            return [(m, "weight"), (m, "bias"), (m, "running_mean"), (m, "running_var")]
        for seq in convs:
            slots += [(seq[0], "weight")] + bn(seq[1])
        for sa in sas:
            slots += [(sa.k_conv, "weight"), (sa.v_conv, "weight"), (sa.v_conv, "bias"), (sa.trans_conv, "weight"), (sa.trans_conv, "bias")] + bn(sa.after_norm)
        for m in n2ps:
            slots += [(m.q_conv, "weight"), (m.k_conv, "weight"), (m.v_conv, "weight")] + bn(m.bn1) + [(m.ff[0], "weight"), (m.ff[2], "weight")] + bn(m.bn2)
        ts = [getattr(m, a) for m, a in slots]
        where = [i for i, t in enumerate(ts) if isinstance(t, nn.Parameter)]
        self.__dict__["_tt_slots"] = slots
        return ts, where, [ts[i] for i in where], bns

    def _train_state(self):
        """_train_table() once per parameter set, not three times per step: (tensors, positions of the trainable ones, the
        trainable ones, BatchNorm modules, detached aliases for the pointer table, parameters fit the native path).  The 167
        attribute walks, `detach()`s and layout checks cost the host ~1 ms per step.  The entry is dropped whenever the module
        is converted (`_apply`: .to / .cuda / .float), its mode changes or a state_dict is loaded, and it is re-validated on EVERY
        use over ALL 167 entries: the owning module still holds the very tensor object that was cached (a replaced Parameter or
        buffer, weight tying, parametrizations) and that tensor still lives at the cached address (`p.data = ...`) — two dictionary
        look-ups and a data_ptr() per entry, ~40 us per step.  A mismatch rebuilds the table; nothing reads stale storage."""
        c = self.__dict__.get("_tt_cache")
        if c is not None:
            slots, ts, ptrs = c[0], c[2][0], c[1]
            for (m, a), t, ptr in zip(slots, ts, ptrs):
                cur = m._parameters.get(a)
                if cur is None:
                    cur = m._buffers.get(a)
                if cur is not t or t.data_ptr() != ptr:
                    c = None
                    break
            if c is not None:
                return c[2]
        ts, where, trainable, bns = self._train_table()
        ok = (not any(t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda for t in ts)
              and len({(m.eps, m.momentum) for m in bns}) == 1 and bns[0].momentum is not None
              and all(m.track_running_stats for m in bns))
        state = (ts, where, trainable, bns, [t.detach() for t in ts], ok)
        self.__dict__["_tt_cache"] = (self.__dict__.pop("_tt_slots"), [t.data_ptr() for t in ts], state)
        return state

    def __getstate__(self):
        """(pickling / torch.save(net): without the cached pointer table — it holds a second reference to every tensor)"""
        d = dict(super().__getstate__()) if hasattr(super(), "__getstate__") else dict(self.__dict__)
        d.pop("_tt_cache", None)
        d.pop("_tt_slots", None)
        return d

    def invalidate_train_state(self):
        self.__dict__.pop("_tt_cache", None)

    def _apply(self, fn, *a, **kw):
        self.invalidate_train_state()
        return super()._apply(fn, *a, **kw)

    def train(self, mode=True):
        self.invalidate_train_state()
        return super().train(mode)

    def load_state_dict(self, *a, **kw):
        self.invalidate_train_state()
        return super().load_state_dict(*a, **kw)

    def _native_train_ok(self, x, dino_feat):
        """The native training path takes plain data tensors (no gradient w.r.t. x / dino_feat), fp32 contiguous parameters and
        one (eps, momentum) for all BatchNorms; anything else goes through the autograd path below."""
        if not self.native_train or not x.is_cuda:
            return False
        if x.requires_grad or dino_feat.requires_grad or x.dtype != torch.float32 or dino_feat.dtype != torch.float32:
            return False
        return self._train_state()[5]

    def _forward_train_native(self, x, dino_feat):
        """Training forward + backward as ONE autograd node over dvm_uni3fc_train_{fwd,bwd}_f32 (csrc/dvm_uni3fc_train.hip): the
        launches of _forward_train_pm and of its autograd graph without the ~1500 Python / autograd hops per call."""
        ts, where, trainable, bns, det, _ = self._train_state()
        with torch.no_grad():
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        meta = (det, where, self.k, bns[0].eps, bns[0].momentum) + self._sync_meta()
        self.__dict__["native_train_calls"] = self.__dict__.get("native_train_calls", 0) + 1
        return nn_ops.uni3fc_train(meta, x.contiguous(), dino_feat.contiguous(), trainable)

    def _sync_meta(self):
        """Data-parallel training with batch statistics over ALL ranks (train_driver.py --sync-stats): `self.sync_stats` holds the
        collective (dvm.dist.TorchCollective) the native node hands its BatchNorm totals and position-encoding range to — the
        node's meta then carries it behind (deferred-statistics list, groups)."""
        coll = getattr(self, "sync_stats", None)
        return () if coll is None else (None, 1, coll)

    def forward_pair(self, x1, dino1, x2, dino2, upsampler=None):
        """The two network calls of a training step (train.py:100-101: `Uni3FC(verts1^T, dino1)`, `Uni3FC(verts2^T, dino2)`) — same
        results as calling forward twice.  In train mode on the native path, with equal point counts, the two calls are ONE native
        call with two groups (BatchNorm statistics, position-encoding range and running-statistics updates per call, in call order:
        bit-identical forward, half the launches); otherwise simply two calls.  -> ((feat1, cfeats1), (feat2, cfeats2)).
        (Two native nodes side by side on two streams were measured slower for the whole step — 21.3 -> 23.2 ms — and removed.)"""
        native = (self.merge_pair_calls and self.training and x1.is_cuda and dino1 is not None and dino2 is not None
                  and tuple(x1.shape) == tuple(x2.shape) and self.point_major_train and torch.is_grad_enabled()
                  and all(type(m) is nn.BatchNorm1d for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))
                  and self._native_train_ok(x1, dino1) and self._native_train_ok(x2, dino2) and ops.knn_tap() is None)
        if not native:
            return self.forward(x1, dino1, upsampler), self.forward(x2, dino2, upsampler)
        ts, where, trainable, bns, det, _ = self._train_state()
        with torch.no_grad():
            torch._foreach_add_([m.num_batches_tracked for m in bns], 2)
        meta = (det, where, self.k, bns[0].eps, bns[0].momentum)
        self.__dict__["native_train_calls"] = self.__dict__.get("native_train_calls", 0) + 1
        return nn_ops.uni3fc_train_merged(meta + self._sync_meta(), x1, dino1, x2, dino2, trainable)

    def _forward_train_pm(self, x, dino_feat):
        """Autograd forward with activations kept point-major (B,N,C), the layout dino_feat arrives in and the kNN /
        attention cores work in: the convs are dvm_linear_f32 GEMMs forward, dvm_linear_f32 / dvm_linear_wgrad_f32
        backward, the BatchNorms the fused row-major kernels; one transpose (the position encoding) in the whole pass."""
        B, _, N = x.shape
        blk = lambda seq, xt: nn_ops.bn_act_pm(seq[1], nn_ops.linear_pm(xt, seq[0].weight), slope=seq[2].negative_slope)  # noqa: E731
        with nn_ops.batched_counter_updates():       # 52 `num_batches_tracked += 1` as one launch
            return self._train_pm_body(x, dino_feat, blk, B, N)

    def _train_pm_body(self, x, dino_feat, blk, B, N):
        f = blk(self.conv, dino_feat)
        tmp = blk(self.conv0, f + self.pos_encoding_sin_wave(x).transpose(1, 2))

        def global_branch():
            x1g = self.sa1.train_pm(tmp)
            x2g = self.sa2.train_pm(x1g)
            x3g = self.sa3.train_pm(x2g)
            glo = torch.cat((x1g, x2g, x3g, self.sa4.train_pm(x3g)), dim=-1)
            gmax = blk(self.conv2, glo).max(dim=1, keepdim=True)[0].expand(-1, N, -1)
            return blk(self.conv4, torch.cat((gmax, glo), dim=-1))

        def local_branch():
            x1 = self.n2p_attention1.train_pm(tmp)
            x2 = self.n2p_attention2.train_pm(x1)
            x3 = self.n2p_attention3.train_pm(x2)
            loc = torch.cat((x1, x2, x3, self.n2p_attention4.train_pm(x3)), dim=-1)
            lmax = blk(self.conv1, loc).max(dim=1, keepdim=True)[0].expand(-1, N, -1)
            return blk(self.conv3, torch.cat((lmax, loc), dim=-1))

        y = torch.cat(_two_branches(tmp, local_branch, global_branch), dim=-1)
        y1 = blk(self.conv5, y)
        y2 = self.n2p_attention5.train_pm(y1)
        y3 = self.n2p_attention6.train_pm(y2)
        y4 = self.n2p_attention7.train_pm(y3)
        out = blk(self.conv6, torch.cat((y1, y2, y3, y4), dim=-1))
        return out.view(B, N, self.out), tmp

    def forward(self, x, dino_feat, upsampler=None):
        """x (B,3,N), dino_feat (B,N,1152) -> (feat (B,N,128), cfeats (B,N,64))."""
        if dino_feat is None:
            dino_feat = self.visual_features(x, upsampler)
        B, _, N = x.shape
        # the no-autograd path: eval mode AND nothing that could want a gradient (eval-mode fine-tuning with
        # frozen BatchNorm statistics must still reach the parameters)
        wants_grad = torch.is_grad_enabled() and (dino_feat.requires_grad or x.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if not self.training and not wants_grad:
            return self._forward_infer(x, dino_feat)
        if self.training and self.point_major_train and \
                all(type(m) is nn.BatchNorm1d for m in self.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)):
            if self._native_train_ok(x, dino_feat):
                return self._forward_train_native(x, dino_feat)
            if getattr(self, "sync_stats", None) is not None:
                # plain BatchNorm modules + a collective: only the native node combines the statistics over the ranks — the
                # autograd path would silently normalise per shard
                raise RuntimeError("Uni3FC.sync_stats is set but this call cannot take the native training node (native_train = False, "
                                   "inputs that require grad, non-fp32 / non-contiguous parameters): convert the BatchNorms with "
                                   "torch.nn.SyncBatchNorm.convert_sync_batchnorm and set sync_minmax instead")
            return self._forward_train_pm(x, dino_feat.contiguous())
        # channel-major fallback (eval-mode fine-tuning, SyncBatchNorm): the reference's own layout.
        # conv -> BatchNorm -> LeakyReLU blocks: the GEMM, then ONE fused statistics + normalise + activation pass
        blk = lambda seq, t: nn_ops.bn_act(seq[1], seq[0](t), slope=seq[2].negative_slope)  # noqa: E731
        f = blk(self.conv, dino_feat.permute(0, 2, 1))
        tmp = blk(self.conv0, f + self.pos_encoding_sin_wave(x))
        x1 = self.n2p_attention1(tmp)
        x1g = self.sa1(tmp)
        x2 = self.n2p_attention2(x1)
        x2g = self.sa2(x1g)
        x3 = self.n2p_attention3(x2)
        x3g = self.sa3(x2g)
        x4 = self.n2p_attention4(x3)
        x4g = self.sa4(x3g)
        loc = torch.cat((x1, x2, x3, x4), dim=1)
        glo = torch.cat((x1g, x2g, x3g, x4g), dim=1)
        lmax = blk(self.conv1, loc).max(dim=-1, keepdim=True)[0].expand(-1, -1, N)
        gmax = blk(self.conv2, glo).max(dim=-1, keepdim=True)[0].expand(-1, -1, N)
        y = torch.cat((blk(self.conv3, torch.cat((lmax, loc), dim=1)), blk(self.conv4, torch.cat((gmax, glo), dim=1))), dim=1)
        y1 = blk(self.conv5, y)
        y2 = self.n2p_attention5(y1)
        y3 = self.n2p_attention6(y2)
        y4 = self.n2p_attention7(y3)
        out = blk(self.conv6, torch.cat((y1, y2, y3, y4), dim=1))
        return out.transpose(2, 1).contiguous().view(B, N, self.out), tmp.permute(0, 2, 1)
