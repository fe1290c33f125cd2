"""Tile-configuration sweep of dvm_linear_f32 (DVM_LINEAR_CFG forces an entry of the configuration table)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
B, N, reps = 8, 2048, 20
def timeit(f):   # best of three runs of `reps` calls (the clock ramps between configurations)
    best = 1e9
    for _ in range(3):
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    return best
for name, K, Co in [("conv 1152->384", 1152, 384), ("conv0 384->64", 384, 64), ("qkv64 64->192", 64, 192), ("ff64a 64->256", 64, 256), ("ff64b 256->64", 256, 64), ("sa v 64->64", 64, 64), ("conv1 256->512", 256, 512), ("conv5 256->128", 256, 128), ("qkv128 128->384", 128, 384), ("ff128a 128->512", 128, 512), ("ff128b 512->128", 512, 128), ("conv6 512->128", 512, 128)]:
    x = torch.randn(B * N, K, device="cuda"); w = torch.randn(Co, K, device="cuda") / K ** 0.5
    xc = x.view(B, N, K).transpose(1, 2).contiguous()
    out = [name]
    for cm, ncfg in ((False, 8), (True, 4)):
        for c in list(range(ncfg)) + [-1]:
            if c >= 0: os.environ["DVM_LINEAR_CFG"] = str(c)
            else: os.environ.pop("DVM_LINEAR_CFG", None)
            t = timeit((lambda: ops.linear(xc, w, channel_major=True)) if cm else (lambda: ops.linear(x, w)))
            out.append("%s%s:%.0f" % ("cm" if cm else "pm", c if c >= 0 else "auto", t * 1e6))
    print(" ".join(out))
