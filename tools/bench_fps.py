import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dv-matcher_amd"))
import torch
from dvm import ops
for (B, N) in ((1, 4995), (16, 2048), (8, 1024), (2, 12000)):
    v = torch.rand(B, N, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(2): ops.fps(v, N // 2, st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.fps(v, N // 2, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("T=%s fps B=%d N=%d: %.3f ms (%.2f us/step)" % (os.environ.get("DVM_FPS_THREADS", "auto"), B, N, dt * 1e3, dt * 1e6 / (N // 2)))
