"""Which operator gives different bits when replayed from a HIP graph than when run eagerly?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
from models.model import Uni3FC
g = torch.Generator().manual_seed(0)
B, N = 8, 2048
xt64 = torch.randn(B, N, 64, generator=g).cuda(); xt128 = torch.randn(B, N, 128, generator=g).cuda()
coor = torch.rand(B, 3, N, generator=g).cuda(); d1152 = torch.randn(B, N, 1152, generator=g).cuda()
w = torch.randn(384, 1152, generator=g).cuda() / 34; w64 = torch.randn(192, 64, generator=g).cuda() / 8
p16 = torch.randn(B, N, 16, generator=g).cuda() * 0.5
idx = ops.knn_neg(xt64, xt64, 40)
qkv = ops.linear(xt64, w64)
cases = {
    "posenc": lambda: ops.pos_encoding(coor),
    "linear 1152->384": lambda: ops.linear(d1152, w),
    "linear 64->192": lambda: ops.linear(xt64, w64),
    "knn64": lambda: ops.knn_neg(xt64, xt64, 40).float(),
    "knn128": lambda: ops.knn_neg(xt128, xt128, 40).float(),
    "n2p core": lambda: ops.n2p_core_fwd(qkv, idx, 4)[0],
    "sa pm": lambda: ops.sa_attention_pm(p16, xt64),
}
for name, f in cases.items():
    eager = f().clone()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): f()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = f()
    gr.replay(); torch.cuda.synchronize()
    print("%-18s graph == eager: %s  (max |diff| %.3g)" % (name, bool(torch.equal(out, eager)), float((out - eager).abs().max())))
