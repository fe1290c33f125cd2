"""LG-Net alone, training forward + backward (no criterion): `calls` network calls of B shapes of N points per step, native
node (DVM_NATIVE_TRAIN=1, default) or autograd path (=0).  usage: bench_train_net.py [B N calls reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from models.model import Uni3FC, join_side_streams
from dvm import nn_ops
from dvm.dist import FlatGradBucket
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.manual_seed(0)
net = Uni3FC(k=40).cuda().train()
bucket = FlatGradBucket(list(net.parameters()), attach=True)
nn_ops.fuse_grad_accumulation(True)
g = torch.Generator().manual_seed(1)
xs = [(torch.rand(B, 3, N, generator=g) - 0.5).cuda() for _ in range(calls)]
ds = [torch.randn(B, N, 1152, generator=g).cuda() for _ in range(calls)]
gf = torch.randn(B, N, 128, generator=g).cuda()


CONC = os.environ.get("CONC", "0") == "1"   # experiment: every call on its own stream (BatchNorm running statistics race: timing only)
streams = [torch.cuda.Stream() for _ in range(calls)]


def step():
    if os.environ.get("CONC", "0") == "2" and calls == 2:      # the product's form of the same idea: Uni3FC.forward_pair
        (fa, _), (fb, _) = net.forward_pair(xs[0], ds[0], xs[1], ds[1])
        torch.autograd.backward([fa, fb], [gf, gf])
    elif CONC:
        cur = torch.cuda.current_stream()
        fs = []
        for st, x, d in zip(streams, xs, ds):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                fs.append(net(x, d, None)[0])
        for st, f in zip(streams, fs):
            with torch.cuda.stream(st):
                f.backward(gf)
        for st in streams:
            cur.wait_stream(st)
    else:
        fs = [net(x, d, None)[0] for x, d in zip(xs, ds)]
        torch.autograd.backward(fs, [gf] * calls)
    join_side_streams(torch.device("cuda", 0))
    bucket.zero()


for _ in range(3):
    step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps):
    step()
th = time.perf_counter() - t
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print("LG-Net fwd+bwd: %d call(s) x B=%d x N=%d: %.2f ms per step (host enqueue %.2f ms), native=%s" % (calls, B, N, dt * 1e3, th / reps * 1e3, os.environ.get("DVM_NATIVE_TRAIN", "1")))
