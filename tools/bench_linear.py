"""dvm_linear_f32 (fp32 matrix-core 1x1 conv, the reference's chain) vs torch's library GEMM at LG-Net's layer shapes:
`bench_linear.py [B N reps]`.  Prints time, algorithmic TFLOP/s (2*M*K*Co / t) and the fraction of the 157.3 TFLOP/s
fp32 matrix peak."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
LAYERS = [("conv 1152->384", 1152, 384), ("conv0 384->64", 384, 64), ("qkv64 64->192", 64, 192), ("ff64a 64->256", 64, 256),
          ("ff64b 256->64", 256, 64), ("sa v 64->64", 64, 64), ("conv1 256->512", 256, 512), ("conv3 768->128", 768, 128),
          ("conv5 256->128", 256, 128), ("qkv128 128->384", 128, 384), ("ff128a 128->512", 128, 512), ("ff128b 512->128", 512, 128),
          ("conv6 512->128", 512, 128)]


def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


torch.manual_seed(0)
tot = [0.0, 0.0, 0.0, 0.0]
for name, K, Co in LAYERS:
    x = torch.randn(B * N, K, device="cuda"); w = torch.randn(Co, K, device="cuda") / K ** 0.5
    xc = x.view(B, N, K).transpose(1, 2).contiguous()
    al, be = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda")
    t_pm = timeit(lambda: ops.linear(x, w, bn=(al, be), slope=0.2))
    t_cm = timeit(lambda: ops.linear(xc, w, channel_major=True))
    t_lib = timeit(lambda: torch.nn.functional.leaky_relu(torch.nn.functional.linear(x, w) * al + be, 0.2))
    t_bmm = timeit(lambda: torch.bmm(w.unsqueeze(0).expand(B, -1, -1), xc))
    fl = 2.0 * B * N * K * Co
    tot = [a + b for a, b in zip(tot, (t_pm, t_cm, t_lib, t_bmm))]
    print("%-18s point-major+epilogue %7.1f us (%5.1f TF, %.2f of peak) | channel-major %7.1f us (%5.1f TF) | torch linear+bn+act %7.1f us | torch bmm %7.1f us"
          % (name, t_pm * 1e6, fl / t_pm / 1e12, fl / t_pm / 157.3e12, t_cm * 1e6, fl / t_cm / 1e12, t_lib * 1e6, t_bmm * 1e6))
print("sum over layers: ours pm %.1f us, ours cm %.1f us, torch pm(+bn,act) %.1f us, torch bmm %.1f us" % tuple(t * 1e6 for t in tot))
