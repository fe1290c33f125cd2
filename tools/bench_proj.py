"""Visual-feature injection kernels at the shipped shape: B shapes x 3 views, N points, C x H x W feature maps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
import torch.nn.functional as F
from dvm import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4995
C, H, W = 384, 256, 256
dev = torch.device("cuda", 0)
pts = (torch.randn(3 * B, N, 3) * torch.tensor([0.2, 0.5, 0.15])).to(dev)
f = torch.randn(3 * B, C, H, W, device=dev)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps


t_proj = timed(lambda: ops.proj2img(pts))
img, pc_min, grid, off = ops.proj2img(pts)
t_i2p = timed(lambda: ops.i2p(pts, f, pc_min, grid, off, normalize=True))


def torch_i2p():
    dense = F.interpolate(f, size=(224, 224), mode='bicubic').reshape(3 * B, C, -1).permute(0, 2, 1)
    idx = torch.floor((pts[:, :, :2] - pc_min[:, None]) / grid[:, None, None]) + 1 + off[:, None]
    flat = (idx[:, :, 0] * 224 + idx[:, :, 1]).long()
    return F.normalize(torch.gather(dense, 1, flat.unsqueeze(-1).expand(-1, -1, C)), dim=-1)


t_ref = timed(torch_i2p, 5)
alg = 3 * B * N * (16 * C * 4 + C * 4)
print("proj2img (3B=%d views, N=%d): %.1f us" % (3 * B, N, t_proj * 1e6))
print("i2p fused: %.1f us (%.0f GB/s of tap+output bytes); torch interpolate+gather+normalize: %.1f us" %
      (t_i2p * 1e6, alg / t_i2p / 1e9, t_ref * 1e6))
