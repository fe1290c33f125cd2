"""FPS kernel forms: per-launch time and identical indices.  usage: python tools/gpu/r5_fps.py  (the multi-wave form it was written for was measured slower and removed: profiles/notes_train.md)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch  # noqa: E402
from dvm import ops  # noqa: E402
for B, N in ((128, 2048), (1024, 2048), (16, 2048), (128, 1000), (128, 500), (64, 77)):
    g = torch.Generator().manual_seed(B + N)
    x = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    x[0, : N // 2] = x[0, N // 2: 2 * (N // 2)]          # duplicated points: ties
    st = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).cuda()
    out = ops.fps(x, N // 2, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = ops.fps(x, N // 2, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("B %5d N %5d: %.3f ms per launch, %.3f us per step, checksum %d" % (B, N, dt * 1e3, dt * 1e6 / (N // 2), int((out.long() * torch.arange(1, N // 2 + 1, device="cuda")).sum())))
