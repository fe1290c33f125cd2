"""kernel-level look at the self-kNN (run under rocprofv3 --kernel-trace --stats): `prof_knn.py [kind] [C]`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "randn"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 2048, C, generator=g)
if kind == "clustered": x = 0.02 * x + torch.randn(8, 1, C, generator=g)
x = x.cuda()
for _ in range(10): ops.knn_neg(x, x, 40)
torch.cuda.synchronize()
