"""cProfile of the host side of LG-Net's training forward + backward (B = 2, N = 1024: host-bound shape)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import nn_ops
from models.model import Uni3FC
net = Uni3FC(k=40).cuda().train()
x = torch.rand(2, 3, 1024).cuda(); d = torch.randn(2, 1024, 1152).cuda()
nn_ops.fuse_grad_accumulation(True)
def step():
    f, c = net(x, d)
    (f.square().mean()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
