"""BASELINE configs[4] — "DINO-ViT multi-view render -> feature injection enabled, N = 4096": Uni3FC.forward(x, None,
upsampler) per shape = 3 depth renderings (HIP) -> ViT-S/14 + JBU x16 (random init; PyTorch-ROCm GEMMs + the HIP adaptive
convolution) -> back-projection (HIP) -> LG-Net.  `bench_visual.py [B N reps]` prints shapes/s and the split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from models.model import Uni3FC
from models.image_backbone import load_upsampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
net = Uni3FC(k=40).cuda().eval(); up = load_upsampler()
x = (torch.rand(B, 3, N) - 0.5).cuda()


def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


with torch.no_grad():
    t_all = timeit(lambda: net(x, None, up))
    dino = net.visual_features(x, up)
    t_vis = timeit(lambda: net.visual_features(x, up))
    imgs = torch.cat([net.proj2img(p)[0] for p in net.views(x)])
    t_up = timeit(lambda: up(imgs))
    t_vit = timeit(lambda: up.model(imgs))
    t_net = timeit(lambda: net(x, dino, None))
print("config 5 (B=%d shapes, N=%d): %.2f ms per batch = %.1f shapes/s | visual features %.2f ms (image backbone %.2f ms of which ViT %.2f ms) | LG-Net %.2f ms"
      % (B, N, t_all * 1e3, B / t_all, t_vis * 1e3, t_up * 1e3, t_vit * 1e3, t_net * 1e3))
