"""Plain-torch fp32 restatements of the floating-point backbone operators.

TEST INFRASTRUCTURE ONLY (same status as oracle/dvm_oracle.c): the checker for the HIP
kernels K3-K5, the positional encoding and the dist-loss term.  Written from the math, in the
reference's (B,C,N) layout; pinned against the golden vectors in tests/test_oracle_vs_golden.py.
"""
import math

import torch
import torch.nn.functional as F


def knn_scores(a, b):
    """(-|a|^2 - (-2 a b^T)) - |b|^2 : (B,N,C),(B,M,C) -> (B,N,M)   (models/model.py:267-278)."""
    inner = -2 * torch.matmul(a, b.transpose(2, 1))
    return -(a ** 2).sum(2, keepdim=True) - inner - (b ** 2).sum(2, keepdim=True).transpose(2, 1)


def pos_encoding(coor):
    """(B,3,N) -> (B,384,N)   (models/model.py:544-561)."""
    nc = 2 * ((coor - coor.min()) / (coor.max() - coor.min())) - 1
    freqs = math.pi * (2 ** torch.arange(64, dtype=torch.float, device=coor.device))
    k = nc.unsqueeze(-1) * freqs.view(1, 1, 1, -1)
    x = torch.cat([torch.sin(k), torch.cos(k)], -1)
    return x.transpose(-1, -2).reshape(coor.shape[0], -1, coor.shape[-1])


def sa_attention(x, w_qk, w_v, b_v):
    """x (B,64,N) -> x_r (B,64,N)   (models/model.py:113-121)."""
    p = F.conv1d(x, w_qk)
    v = F.conv1d(x, w_v, b_v)
    att = torch.softmax(torch.bmm(p.transpose(1, 2), p), dim=-1)
    att = att / (1e-9 + att.sum(dim=1, keepdim=True))
    return torch.bmm(v, att)


def n2p_attention(x, idx, wq, wk, wv, heads=4):
    """x (B,C,N), idx (B,N,K) -> (B,C,N)   (models/model.py:339-350): attention of each point over
    the differences to its K neighbours."""
    B, C, N = x.shape
    K = idx.shape[-1]
    xt = x.transpose(1, 2)
    nb = torch.gather(xt, 1, idx.reshape(B, N * K, 1).expand(-1, -1, C).long()).view(B, N, K, C)
    diff = (nb - xt[:, :, None, :]).permute(0, 3, 1, 2)  # (B,C,N,K)
    q = F.conv2d(x.unsqueeze(-1), wq)
    k = F.conv2d(diff, wk)
    v = F.conv2d(diff, wv)
    D = C // heads

    def split(t):
        return t.view(B, heads, D, N, -1).permute(0, 1, 3, 4, 2)  # (B,H,N,K,D)

    q, k, v = split(q), split(k), split(v)
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), dim=-1)
    return (att @ v).squeeze(3).permute(0, 1, 3, 2).reshape(B, C, N)


def _bn(sd, name, x, train, eps=1e-5):
    if train:
        return F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], True, 0.1, eps)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        False, 0.1, eps)


def sa_layer(sd, name, x, train):
    """SA_Layer.forward (models/model.py:113-123) from a state_dict."""
    xr = sa_attention(x, sd[name + ".k_conv.weight"], sd[name + ".v_conv.weight"], sd[name + ".v_conv.bias"])
    t = F.conv1d(x - xr, sd[name + ".trans_conv.weight"], sd[name + ".trans_conv.bias"])
    return x + torch.relu(_bn(sd, name + ".after_norm", t, train))


def n2p_block(sd, name, x, train, K=40, idx=None, log=None):
    """N2PAttention[_DIM].forward (models/model.py:339-354 / 375-390) from a state_dict; `idx` forces the neighbour
    sets (teacher forcing), `log` receives the sets used."""
    if idx is None:
        xt = x.transpose(1, 2)
        idx = knn_scores(xt, xt).topk(K, dim=-1)[1]
    if log is not None:
        log.append(idx)
    att = n2p_attention(x, idx, sd[name + ".q_conv.weight"], sd[name + ".k_conv.weight"], sd[name + ".v_conv.weight"])
    x = _bn(sd, name + ".bn1", x + att, train)
    ff = F.conv1d(F.leaky_relu(F.conv1d(x, sd[name + ".ff.0.weight"]), 0.2), sd[name + ".ff.2.weight"])
    return _bn(sd, name + ".bn2", x + ff, train)


def uni3fc(sd, x, dino, train=False, knn_idx=None, log=None):
    """Uni3FC.forward with dino_feat given (models/model.py:711-761) from a state_dict, in plain torch ops and the
    reference's (B,C,N) layout.  x (B,3,N), dino (B,N,1152) -> (feat (B,N,128), cfeats (B,N,64)).
    knn_idx: optional sequence of 7 (B,N,K) index tensors used instead of the network's own kNN searches."""
    take = (lambda i: None) if knn_idx is None else (lambda i: torch.as_tensor(knn_idx[i]).long())

    def blk(n, t):
        return F.leaky_relu(_bn(sd, "bn" + n, F.conv1d(t, sd["conv%s.0.weight" % n]), train), 0.2)

    f = blk("", dino.permute(0, 2, 1))
    tmp = blk("0", f + pos_encoding(x))
    xs, gs = [tmp], [tmp]
    for i in range(4):
        xs.append(n2p_block(sd, "n2p_attention%d" % (i + 1), xs[-1], train, idx=take(i), log=log))
        gs.append(sa_layer(sd, "sa%d" % (i + 1), gs[-1], train))
    loc, glo = torch.cat(xs[1:], 1), torch.cat(gs[1:], 1)
    N = x.shape[2]
    lmax = blk("1", loc).max(dim=-1, keepdim=True)[0].repeat(1, 1, N)
    gmax = blk("2", glo).max(dim=-1, keepdim=True)[0].repeat(1, 1, N)
    y = torch.cat((blk("3", torch.cat((lmax, loc), 1)), blk("4", torch.cat((gmax, glo), 1))), 1)
    ys = [blk("5", y)]
    for i in range(3):
        ys.append(n2p_block(sd, "n2p_attention%d" % (i + 5), ys[-1], train, idx=take(4 + i), log=log))
    out = blk("6", torch.cat(ys, 1))
    return out.transpose(2, 1).contiguous(), tmp.permute(0, 2, 1)


def dist_loss_term(feat, dist, anchors, k):
    """(B,N,C), (B,N,N), (nA,) -> (B,)   (models/loss.py:1361-1394)."""
    f1 = feat[:, anchors]
    idx = knn_scores(f1, feat).topk(k, dim=-1)[1]
    B, nA, _ = idx.shape
    f2 = torch.gather(feat, 1, idx.reshape(B, -1, 1).expand(-1, -1, feat.shape[-1])).view(B, nA, k, -1)
    x = torch.norm(f2 - f1[:, :, None, :], dim=-1)
    y = torch.stack([dist[b][idx[b].reshape(-1), anchors.repeat_interleave(k)].view(nA, k) for b in range(B)])
    return (1 - torch.abs(F.cosine_similarity(x, y, dim=2))).sum(1)


def softcorr_bwd(f1, f2, neg_alpha, idx, gval):
    """fp64 autograd through softmax(neg_alpha * cdist(f1, f2)) gathered at idx (reference
    models/loss.py:110-114 + 1339-1347): returns (val, d_f1, d_f2) for L = sum(val * gval).
    f1 (B,N,d), f2 (B,M,d) float32 tensors; idx (B,N,k) int; gval (B,N,k)."""
    a = f1.detach().double().requires_grad_(True)
    b = f2.detach().double().requires_grad_(True)
    diff = a[:, :, None, :] - b[:, None, :, :]
    D = torch.sqrt((diff * diff).sum(-1).clamp_min(1e-300))
    P = torch.softmax(D * float(neg_alpha), dim=-1)
    val = torch.gather(P, 2, idx.long())
    (val * gval.double()).sum().backward()
    return val.detach(), a.grad, b.grad


# ------------------------------------------------------------------ visual-feature injection
_IMG = 224


def piyg_lut():
    """The 256 x 3 float32 'PiYG' table (ColorBrewer anchors, linear interpolation) — restated, see tools/gen_piyg_lut.py."""
    import numpy as np
    anchors = [(142, 1, 82), (197, 27, 125), (222, 119, 174), (241, 182, 218), (253, 224, 239), (247, 247, 247),
               (230, 245, 208), (184, 225, 134), (127, 188, 65), (77, 146, 33), (39, 100, 25)]
    anch = np.asarray(anchors, dtype=np.float64) / 255.0
    n = 256
    xs = np.linspace(0.0, 1.0, len(anch)) * (n - 1)
    xind = np.linspace(0.0, 1.0, n) * (n - 1)
    ind = np.searchsorted(xs, xind)[1:-1]
    dist = (xind[1:-1] - xs[ind - 1]) / (xs[ind] - xs[ind - 1])
    cols = [np.clip(np.concatenate([[anch[0, c]], dist * (anch[ind, c] - anch[ind - 1, c]) + anch[ind - 1, c], [anch[-1, c]]]), 0, 1)
            for c in range(3)]
    return torch.from_numpy(np.stack(cols, 1).astype(np.float32))


def proj2img(pc):
    """models/model.py:584-650 + 563-581 on CPU tensors.  pc (B,N,3) -> (img (B,3,224,224), pc_min (B,1,2),
    grid_size (B,1,1), (offset_x (B,1), offset_y (B,1)))."""
    B, N, _ = pc.shape
    offs = torch.tensor([[i, j] for i in range(-2, 3) for j in range(-2, 3)], dtype=torch.float32)
    pc_range = pc.max(dim=1)[0] - pc.min(dim=1)[0]
    grid_size = (pc_range[:, :2].max(dim=-1)[0] / (_IMG - 3)).view(B, 1, 1)
    pc_min = pc.min(dim=1)[0][:, :2].unsqueeze(1)
    idx_xy = torch.floor((pc[:, :, :2] - pc_min) / grid_size)
    dense = (idx_xy.unsqueeze(2) + offs[None, None]).view(B, N * 25, 2) + 1
    center = torch.floor((dense.max(dim=1)[0] + dense.min(dim=1)[0]) / 2).int()
    offset_x = _IMG / 2 - center[:, 0:1] - 1
    offset_y = _IMG / 2 - center[:, 1:2] - 1
    dense = dense + torch.cat([offset_x, offset_y], dim=1).unsqueeze(1)
    z = pc[:, :, 2:3].expand(-1, -1, 25).reshape(B, N * 25)
    dense = dense + (dense < 0).to(torch.int32) - (dense > _IMG - 1).to(torch.int32)
    assert dense.min() >= 0 and dense.max() <= _IMG - 1
    flat = (dense[:, :, 0] * _IMG + dense[:, :, 1]).long()
    acc = torch.zeros(B, _IMG * _IMG, dtype=torch.float32).scatter_add_(1, flat, z)     # torch_scatter 'sum'
    img = acc.view(B, _IMG, _IMG)
    zero_mask = img == 0
    v = (torch.sigmoid(img) - 0.485) / 0.229
    lut = piyg_lut()
    out = torch.empty(B, 3, _IMG, _IMG)
    for b in range(B):
        d = (v[b] - v[b].min()) / (v[b].max() - v[b].min())
        k = torch.clamp((d * 256).long(), max=255)
        col = lut[k]                                                                    # (224,224,3)
        col[torch.isnan(d)] = 0
        out[b] = col.permute(2, 0, 1)
    out[zero_mask.unsqueeze(1).expand(-1, 3, -1, -1)] = -1
    return out, pc_min, grid_size, (offset_x, offset_y)


def i2p(pc, f, pc_min, grid_size, offsets):
    """models/model.py:653-678 on CPU tensors: bicubic resize of f to 224x224, gather at each point's pixel."""
    B, N, _ = pc.shape
    C = f.shape[1]
    idx = torch.floor((pc[:, :, :2] - pc_min) / grid_size) + 1 + torch.cat(offsets, dim=1).unsqueeze(1)
    assert idx.min() >= 0 and idx.max() <= _IMG - 1
    dense = F.interpolate(f, size=(_IMG, _IMG), mode='bicubic').reshape(B, C, -1).permute(0, 2, 1)
    flat = (idx[:, :, 0] * _IMG + idx[:, :, 1]).long()
    return torch.gather(dense, 1, flat.unsqueeze(-1).expand(-1, -1, C))


def visual_features(x, upsampler):
    """models/model.py:683-710: x (B,3,N) -> (B,N,3C)."""
    c, s = math.cos(-math.pi / 2), math.sin(-math.pi / 2)
    rot = torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float64).float()
    pts_1 = torch.bmm(x.permute(0, 2, 1), rot[None].repeat(x.shape[0], 1, 1))
    pts = [pts_1, torch.cat((pts_1[..., 2:3], pts_1[..., 0:2]), -1), torch.cat((pts_1[..., 1:3], pts_1[..., 0:1]), -1)]
    proj = [proj2img(p) for p in pts]
    feats = upsampler(torch.cat([p[0] for p in proj], 0))
    B = x.shape[0]
    return torch.cat([F.normalize(i2p(pts[v], feats[v * B:(v + 1) * B], *proj[v][1:]), dim=-1) for v in range(3)], -1)


def rot6d(d6):
    """6D -> rotation matrix by Gram-Schmidt (reference models/loss.py:39-45): rows b1, b2, b1 x b2.  Any dtype; under
    autograd this is the checker of dvm_rot6d_{fwd,bwd}_f32."""
    first, second = d6[..., 0:3], d6[..., 3:6]
    b1 = F.normalize(first, dim=-1)
    b2 = F.normalize(second - (b1 * second).sum(dim=-1, keepdim=True) * b1, dim=-1)
    return torch.stack([b1, b2, torch.linalg.cross(b1, b2, dim=-1)], dim=-2)


def dg_warp_arap(verts, g, R, T):
    """Embedded-deformation warp and ARAP energy of a batch of graphs, differentiable w.r.t. R and T (the reference's
    DeformationGraph_geod.forward, lib/deformation_graph_point.py:233-261, batched): every point moves with its 3
    influencing nodes, weighted; ARAP compares each node's 9 ring neighbours after the motion with their rotated rest
    offsets, summed and divided by the node count.  verts (B,N,3); g: dict of nodes_idx (B,Nn), infl_idx (B,N,3),
    weights (B,N,3), one_ring (B,Nn,9); R (B,Nn,3,3); T (B,Nn,3) -> warped (B,N,3), arap (B,).  Checker of
    dvm_dg_warp_{fwd,bwd}_f32 (tests/test_gpu_backward.py), pinned by tests/golden/graddg_*.npz."""
    B, N, _ = verts.shape
    Nn = R.shape[1]
    take = lambda src, idx: torch.gather(src, 1, idx.unsqueeze(-1).expand(-1, -1, src.shape[-1]))  # noqa: E731
    nodes = take(verts, g["nodes_idx"].long())                                   # (B,Nn,3) rest positions of the nodes
    infl = g["infl_idx"].long().reshape(B, N * 3)
    node_of = take(nodes, infl).view(B, N, 3, 3)
    rot_of = take(R.reshape(B, Nn, 9), infl).view(B, N, 3, 3, 3)
    shift_of = take(T, infl).view(B, N, 3, 3)
    moved = torch.einsum("bnsij,bnsj->bnsi", rot_of, verts.unsqueeze(2) - node_of) + node_of + shift_of
    warped = (moved * g["weights"].unsqueeze(-1)).sum(dim=2)
    ring = g["one_ring"].long().reshape(B, Nn * 9)
    ring_rest = take(nodes, ring).view(B, Nn, 9, 3)
    ring_shift = take(T, ring).view(B, Nn, 9, 3)
    rest_offset = nodes.unsqueeze(2) - ring_rest
    residual = (nodes + T).unsqueeze(2) - (ring_rest + ring_shift) - torch.einsum("bnij,bnqj->bnqi", R, rest_offset)
    return warped, residual.square().sum(dim=(1, 2, 3)) / Nn


def deform_terms_dense(feat_s, feat_t, verts_s, verts_t, alpha, g, knn_s, knn_t, dparams, with_map=True, topk=10, cols_out=None):
    """One direction of GraphDeformLoss_Neural.deform() for P pairs in the reference's DENSE formulation, differentiable, in the dtype
    of its inputs (run it in float64 as the checker of dvm_criterion_[dir_]train_{fwd,bwd}_f32): models/loss.py:110-114 + 1339-1347
    (Pi = topk10(softmax(-alpha cdist))), :1408-1409 (Pi @ verts), models/model.py:454-478 (Deformer: Conv2d(k->1) pooling of the xyz
    neighbours' features, Pi @ pooled targets, node rows [v, g, verts12, Pi g], MLP with ELU), models/loss.py:1258-1264 + 39-45 (rot6d
    + identity), lib/deformation_graph_point.py:233-261 (warp, ARAP), models/loss.py:1216-1226 (Chamfer), :1237 (map term).
    feat_s (P,N,C), feat_t (P,M,C), verts_* (P,*,3); g = the SOURCES' batched graph (nodes_idx, one_ring, infl_idx, weights: constants);
    knn_s (P,N,k) / knn_t (P,M,k) xyz-kNN (constants); dparams = [conv_w (k,), conv_b (1,), W0, b0, W1, b1, W2, b2, W3, b3].
    -> terms (P,6): [map numerator, mean d(warped->t), mean d(t->warped), mean d(verts12->t), mean d(t->verts12), ARAP]
    (the first six columns of the native node's table).  One pair at a time: the N x M matrices of a 4995 x 2200 pair are 88 MB each."""
    P, N, C = feat_s.shape
    M = feat_t.shape[1]
    conv_w, conv_b, W0, b0, W1, b1, W2, b2, W3, b3 = dparams
    rows = []
    iden = torch.tensor([1, 0, 0, 0, 1, 0], dtype=feat_s.dtype, device=feat_s.device)
    for p in range(P):
        f1, f2, v1, v2 = feat_s[p], feat_t[p], verts_s[p], verts_t[p]
        pi = torch.softmax(-alpha * torch.cdist(f1[None], f2[None])[0], dim=-1)
        val, col = torch.topk(pi, topk, dim=-1)                                   # (N,10): kept values, no renormalisation
        if cols_out is not None:
            cols_out.append(col)                                                   # (column 0 = the arg-max map)
        v12 = (val.unsqueeze(-1) * v2[col]).sum(1)                                 # Pi @ verts2
        i11, i22 = knn_s[p].long(), knn_t[p].long()
        g1 = (f1[i11] * conv_w.view(1, -1, 1)).sum(1) + conv_b                     # Conv2d(k -> 1, 1x1) over the neighbours
        g2 = (f2[i22] * conv_w.view(1, -1, 1)).sum(1) + conv_b
        nodes = g["nodes_idx"][p].long()
        g12n = (val[nodes].unsqueeze(-1) * g2[col[nodes]]).sum(1)                  # (Pi @ g2) at the nodes
        z = torch.cat([v1[nodes], g1[nodes], v12[nodes], g12n], dim=-1)
        h = F.elu(F.linear(z, W0, b0))
        h = F.elu(F.linear(h, W1, b1))
        h = F.elu(F.linear(h, W2, b2))
        d9 = F.linear(h, W3, b3)
        gp = {k: g[k][p:p + 1] for k in ("nodes_idx", "one_ring", "infl_idx", "weights")}
        warped, arap = dg_warp_arap(v1[None], gp, rot6d(d9[None, :, 3:] + iden), d9[None, :, :3])
        cd = []
        for a in (warped[0], v12):
            D = (a.unsqueeze(1) - v2.unsqueeze(0)).square().sum(-1)               # squared distances (chamfer_3DDist semantics)
            cd += [D.min(1)[0].mean(), D.min(0)[0].mean()]
        if with_map:
            # einsum('bij,bjkm->bikm', Pi, verts2[idx22]) against verts12[idx11]   (models/loss.py:1237, 1410)
            rhs = (val.view(N, topk, 1, 1) * v2[i22[col]]).sum(1)                  # (N,k,3)
            mp = (v12[i11] - rhs).square().sum()
        else:
            mp = v12.sum() * 0
        rows.append(torch.stack([mp, cd[0], cd[1], cd[2], cd[3], arap[0]]))
    return torch.stack(rows)


def pair_direction_aten(weights, feat1, feat2, verts1, verts2, alpha, fps_start, with_map=True):
    """One direction of the pair path (BASELINE configs[1]) the way the REFERENCE'S CPU RUN executes it: dense N x M tensors on ATen /
    MKL with every host thread (cdist, softmax, topk, matmul — deform_terms_dense above), the xyz kNN as cdist + topk
    (models/loss.py:97-101).  The deformation graph comes from the C oracle (oracle.dg_build: the reference's Python FPS loop + scipy
    KDTree takes 0.14 - 0.40 s per shape, SURVEY §8a-13 — this leg does not charge it).  bench.py times it as the second, BLAS-backed
    cpu_baseline figure beside the scalar-chain C oracle; tests/test_modules_cpu.py pins its outputs to that oracle's.
    numpy in (one pair: feat (N,128) / (M,128), verts (N,3) / (M,3)) -> dict(T12, losses[6]) with the C oracle's layout:
    [mean d(warped->t), mean d(t->warped), ARAP, mean d(verts12->t), mean d(t->verts12), map numerator]."""
    import numpy as np
    from . import oracle as O
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    f1, f2, v1, v2 = t(feat1), t(feat2), t(verts1), t(verts2)
    with torch.no_grad():
        g = O.dg_build(verts1, fps_start)
        gt = {k: torch.from_numpy(np.ascontiguousarray(g[k]))[None] for k in ("nodes_idx", "one_ring", "infl_idx", "weights")}
        knn_s = torch.cdist(v1[None], v1[None])[0].topk(10, dim=-1, largest=False)[1][None]
        knn_t = torch.cdist(v2[None], v2[None])[0].topk(10, dim=-1, largest=False)[1][None]
        cw, cb, mats = O._mlp_args(weights)
        dparams = [torch.from_numpy(cw), torch.tensor([cb])] + [torch.from_numpy(m) for m in mats]
        cols = []
        terms = deform_terms_dense(f1[None], f2[None], v1[None], v2[None], float(alpha), gt, knn_s, knn_t, dparams, with_map, cols_out=cols)[0]
    return dict(T12=cols[0][:, 0].numpy().astype(np.int32), losses=terms[[1, 2, 5, 3, 4, 0]].numpy())
