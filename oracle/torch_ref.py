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
