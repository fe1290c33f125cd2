import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from models.model import Uni3FC
net = Uni3FC(k=40).cuda().eval()
x = torch.rand(8, 3, 2048).cuda(); dino = torch.randn(8, 2048, 1152).cuda()
with torch.no_grad():
    for _ in range(3): net(x, dino, None)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net(x, dino, None); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=90))
