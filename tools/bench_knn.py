"""Feature-space self-kNN (dvm_knn_neg_f32, k = 40): the fp16-sweep path (a is b) against the dense N x N path (b a copy of a:
the library cannot know they are equal) — same indices, time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch
from dvm import ops

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

g = torch.Generator().manual_seed(0)
for kind in ("randn", "relu", "clustered", "duplicates"):
    for (B, N, C) in ((8, 2048, 128), (8, 2048, 64), (1, 4995, 128), (2, 700, 64)):
        x = torch.randn(B, N, C, generator=g)
        if kind == "relu": x = 0.3 * torch.relu(x)
        if kind == "clustered": x = 0.02 * x + torch.randn(B, 1, C, generator=g)
        if kind == "duplicates": x[:, N // 2:] = x[:, :N - N // 2].clone()
        x = x.cuda()
        y = x.clone()
        a, b = ops.knn_neg(x, x, 40), ops.knn_neg(x, y, 40)
        same = bool(torch.equal(a, b))
        bad = int((a != b).any(-1).sum())
        t1, t2 = timeit(lambda: ops.knn_neg(x, x, 40)), timeit(lambda: ops.knn_neg(x, y, 40))
        print("%-10s B=%d N=%d C=%d: equal %s (%d rows differ)   sweep path %.1f us   dense path %.1f us" % (kind, B, N, C, same, bad, t1 * 1e6, t2 * 1e6))
