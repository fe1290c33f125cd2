"""Attention cores with inputs well outside the unit scale the parity tests use (large logits, sizes that do not
tile): finite outputs / gradients and agreement with an fp64 torch evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import torch
from dvm import ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (B, N, sc) in ((1, 257, 1.0), (2, 1000, 2.5), (1, 4995, 2.0), (2, 333, 4.0)):
    p = (torch.randn(B, N, 16, generator=g) * sc)
    v = torch.randn(B, N, 64, generator=g)
    gx = torch.randn(B, N, 64, generator=g)
    xr, stats, cinv = ops.sa_attention_train_fwd(p.to(dev), v.to(dev))
    dp, dv = ops.sa_attention_bwd(p.to(dev), v.to(dev), xr, stats, cinv, gx.to(dev))
    pd, vd = p.double().requires_grad_(True), v.double().requires_grad_(True)
    E = pd @ pd.transpose(1, 2)
    A = torch.softmax(E, -1)
    A = A / (1e-9 + A.sum(1, keepdim=True))
    ref = (vd.transpose(1, 2) @ A).transpose(1, 2)
    (ref * gx.double()).sum().backward()
    xe = ops.sa_attention_pm(p.to(dev), v.to(dev))       # the inference path: both contractions as fp16x2-split products
    print("SAev B=%d N=%d scale %.1f: finite %s  fwd %.1e" % (B, N, sc, bool(torch.isfinite(xe).all()), rel(xe, ref)))
    print("SA   B=%d N=%d scale %.1f: finite %s  fwd %.1e  dp %.1e  dv %.1e" %
          (B, N, sc, bool(torch.isfinite(xr).all() and torch.isfinite(dp).all() and torch.isfinite(dv).all()), rel(xr, ref), rel(dp, pd.grad),
           rel(dv, vd.grad)))
for (B, N, C, sc) in ((1, 300, 64, 1.0), (2, 777, 128, 3.0), (1, 4995, 128, 2.0)):
    K = 40
    qkv = torch.randn(B, N, 3 * C, generator=g) * sc
    x = torch.randn(B, N, C, generator=g)
    idx = ops.knn_neg(x.to(dev), x.to(dev), K)
    gout = torch.randn(B, N, C, generator=g)
    out, attn = ops.n2p_core_fwd(qkv.to(dev), idx)
    dq = ops.n2p_core_bwd(qkv.to(dev), idx, attn, gout.to(dev))
    qd = qkv.double().requires_grad_(True)
    q, kp, vp = qd[..., :C], qd[..., C:2 * C], qd[..., 2 * C:]
    ii = idx.long().cpu()
    gather = lambda t: torch.gather(t, 1, ii.reshape(B, N * K, 1).expand(-1, -1, C)).view(B, N, K, C)  # noqa: E731
    kd = gather(kp) - kp.unsqueeze(2)
    vd = gather(vp) - vp.unsqueeze(2)
    H, D = 4, C // 4
    e = (q.view(B, N, 1, H, D) * kd.view(B, N, K, H, D)).sum(-1) / (D ** 0.5)
    a = torch.softmax(e, dim=2)
    ref = (a.unsqueeze(-1) * vd.view(B, N, K, H, D)).sum(2).reshape(B, N, C)
    (ref * gout.double()).sum().backward()
    print("N2P  B=%d N=%d C=%d scale %.1f: finite %s  fwd %.1e  dqkv %.1e" %
          (B, N, C, sc, bool(torch.isfinite(out).all() and torch.isfinite(dq).all()), rel(out, ref), rel(dq, qd.grad)))
