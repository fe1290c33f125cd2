"""The backward GEMMs of the training step, layer by layer at R = 32768 rows (two network calls of 8 x 2048 points merged): measured time
of the input-gradient GEMM (dX = dY W: dvm_linear_f32 with the roles swapped) and of the weight-gradient kernel (dW = dY^T X) against the
two bounds of each — the fp32 matrix pipe (157 TFLOP/s) and the bytes every operand crosses HBM at least once (4.5 TB/s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops, _lib
R, reps = 32768, 20
lib = _lib.load()
def timeit(f):
    best = 1e9
    for _ in range(3):
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    return best
tot = [0, 0, 0, 0]
layers = [("conv 1152->384", 1152, 384, 1, False), ("conv0 384->64", 384, 64, 1, True), ("qkv64 64->192", 64, 192, 4, True), ("ff64a 64->256", 64, 256, 4, True),
          ("ff64b 256->64", 256, 64, 4, True), ("qkv128 128->384", 128, 384, 3, True), ("ff128a 128->512", 128, 512, 3, True), ("ff128b 512->128", 512, 128, 3, True),
          ("conv1/2 256->512", 256, 512, 2, True), ("conv3/4 768->128", 768, 128, 2, True), ("conv5 256->128", 256, 128, 1, True), ("conv6 512->128", 512, 128, 1, True)]
print("%-18s %5s | dgrad us (mfma bound, hbm bound) | wgrad us (mfma bound, hbm bound)" % ("layer", "count"))
for name, K, Co, count, need_dx in layers:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(Co, K, device="cuda") / K ** 0.5; dy = torch.randn(R, Co, device="cuda")
    dx = torch.empty(R, K, device="cuda"); dW = torch.zeros(Co, K, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    td = timeit(lambda: lib.dvm_linear_f32(w.data_ptr(), dy.data_ptr(), 1, K, Co, R, 1, None, None, None, None, 1.0, dx.data_ptr(), s)) if need_dx else 0.0
    tw = timeit(lambda: lib.dvm_linear_wgrad_f32(dy.data_ptr(), x.data_ptr(), R, Co, K, dW.data_ptr(), s))
    fl = 2.0 * R * K * Co
    bd = 4.0 * (R * Co + R * K + Co * K)
    print("%-18s %5d | %6.0f (%5.0f, %5.0f) | %6.0f (%5.0f, %5.0f)" % (name, count, td * 1e6, fl / 157e12 * 1e6, bd / 4.5e12 * 1e6, tw * 1e6, fl / 157e12 * 1e6, bd / 4.5e12 * 1e6))
    tot[0] += td * count; tot[1] += tw * count; tot[2] += (fl / 157e12) * count * (2 if need_dx else 1); tot[3] += (bd / 4.5e12) * count * (2 if need_dx else 1)
print("per step: dgrad %.2f ms, wgrad %.2f ms; bounds: fp32 matrix pipe %.2f ms, HBM once %.2f ms" % (tot[0] * 1e3, tot[1] * 1e3, tot[2] * 1e3, tot[3] * 1e3))
