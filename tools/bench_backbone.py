"""Uni3FC forward (eval) throughput: B shapes of N points with supplied 1152-d visual features."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from models.model import Uni3FC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
net = Uni3FC(k=40).cuda().eval()
x = torch.rand(B, 3, N).cuda(); dino = torch.randn(B, N, 1152).cuda()
with torch.no_grad():
    for _ in range(3): net(x, dino, None)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): net(x, dino, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print("Uni3FC eval forward: B=%d N=%d: %.2f ms (%.1f shapes/s)" % (B, N, dt * 1e3, B / dt))
if len(sys.argv) > 4 and sys.argv[4] == "graph":
    # the same forward captured into a HIP graph (launch-bound at B = 1: ~250 launches per forward)
    sx, sd = x.clone(), dino.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3): net(sx, sd, None)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = net(sx, sd, None)
    with torch.no_grad():          # (the same no-autograd path as the captured forward)
        ref = net(x, dino, None)[0]
    g.replay(); torch.cuda.synchronize()
    print("graph replay equals eager:", bool(torch.equal(out[0], ref)), " max |diff| %.3g, points differing by > 1e-3: %.4f"
          % (float((out[0] - ref).abs().max()), float(((out[0] - ref).abs().amax(-1) > 1e-3).float().mean())))
    t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print("  as a HIP graph: %.2f ms (%.1f shapes/s)" % (dt * 1e3, B / dt))
