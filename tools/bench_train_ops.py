"""Point-major training operators at LG-Net's layer shapes: forward GEMM, dX GEMM, dW (dvm_linear_wgrad_f32) against
torch's library GEMMs on the same operands, and the fused BatchNorm forward / backward (point-major and channel-major
kernels) against the bytes they must move.  `bench_train_ops.py [B N reps]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
LAYERS = [("conv 1152->384", 1152, 384, 1), ("conv0 384->64", 384, 64, 1), ("qkv64 64->192", 64, 192, 4), ("ff64a 64->256", 64, 256, 4),
          ("ff64b 256->64", 256, 64, 4), ("sa pv 64->80", 64, 80, 4), ("sa trans 64->64", 64, 64, 4), ("conv1 256->512", 256, 512, 2),
          ("conv3 768->128", 768, 128, 2), ("conv5 256->128", 256, 128, 1), ("qkv128 128->384", 128, 384, 3),
          ("ff128a 128->512", 128, 512, 3), ("ff128b 512->128", 512, 128, 3), ("conv6 512->128", 512, 128, 1)]
BNS = [(384, 1), (64, 13), (128, 9), (512, 2)]


def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


torch.manual_seed(0)
R = B * N
tot = [0.0] * 6
for name, K, Co, cnt in LAYERS:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(Co, K, device="cuda") / K ** 0.5
    gy = torch.randn(R, Co, device="cuda")
    t_f = timeit(lambda: ops.linear(x, w))
    t_dx = timeit(lambda: ops.linear(w.view(1, Co, K), gy, channel_major=True))
    t_dw = timeit(lambda: ops.linear_wgrad(gy, x))
    l_f = timeit(lambda: x @ w.t())
    l_dx = timeit(lambda: gy @ w)
    l_dw = timeit(lambda: gy.t() @ x)
    fl = 2.0 * R * K * Co
    tot = [a + cnt * b for a, b in zip(tot, (t_f, t_dx, t_dw, l_f, l_dx, l_dw))]
    print("%-18s fwd %6.1f us (%5.1f TF) dX %6.1f us (%5.1f TF) dW %6.1f us (%5.1f TF) | torch fwd %6.1f dX %6.1f dW %6.1f us"
          % (name, t_f * 1e6, fl / t_f / 1e12, t_dx * 1e6, fl / t_dx / 1e12, t_dw * 1e6, fl / t_dw / 1e12, l_f * 1e6, l_dx * 1e6, l_dw * 1e6))
print("per step (layer counts applied): ours fwd %.0f dX %.0f dW %.0f us | torch fwd %.0f dX %.0f dW %.0f us" % tuple(t * 1e6 for t in tot))

tb = [0.0] * 4
for C, cnt in BNS:
    x = torch.randn(1, R, C, device="cuda"); g = torch.rand(C, device="cuda") + 0.5; b = torch.randn(C, device="cuda")
    dy = torch.randn(1, R, C, device="cuda")
    y, m, i = ops.bn_act_train_fwd_pm(x, None, g, b, 1e-5, 0.2, 0.1)
    t_f = timeit(lambda: ops.bn_act_train_fwd_pm(x, None, g, b, 1e-5, 0.2, 0.1))
    t_b = timeit(lambda: ops.bn_act_train_bwd_pm(dy, y, x, None, g, m, i, 0.2))
    xc = x.view(B, N, C).transpose(1, 2).contiguous(); dyc = dy.view(B, N, C).transpose(1, 2).contiguous()
    yc, mc, ic = ops.bn_act_train_fwd(xc, None, g, b, 1e-5, 0.2, 0.1)
    c_f = timeit(lambda: ops.bn_act_train_fwd(xc, None, g, b, 1e-5, 0.2, 0.1))
    c_b = timeit(lambda: ops.bn_act_train_bwd(dyc, yc, xc, None, g, mc, ic, 0.2))
    byt = R * C * 4
    tb = [a + cnt * v for a, v in zip(tb, (t_f, t_b, c_f, c_b))]
    print("BN C=%4d  pm fwd %6.1f us (%4.2f TB/s of 3 passes) bwd %6.1f us (%4.2f TB/s of 7) | cm fwd %6.1f bwd %6.1f us"
          % (C, t_f * 1e6, 3 * byt / t_f / 1e12, t_b * 1e6, 7 * byt / t_b / 1e12, c_f * 1e6, c_b * 1e6))
print("per step: pm fwd %.0f bwd %.0f us | cm fwd %.0f bwd %.0f us" % tuple(t * 1e6 for t in tb))
