import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dv-matcher_amd"))
import torch
from dvm import ops
for (B, N, M) in ((1, 4995, 4995), (8, 2048, 2048), (1, 4995, 2200)):
    f1 = torch.randn(B, N, 128, device="cuda"); f2 = torch.randn(B, M, 128, device="cuda")
    for _ in range(3): ops.argmin_exact(f1, f2)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ops.argmin_exact(f1, f2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("argmin_exact B=%d %dx%d: %.3f ms  (%.1f TFLOP/s of 3*N*M*d)" % (B, N, M, dt * 1e3, 3 * B * N * M * 128 / dt / 1e12))
