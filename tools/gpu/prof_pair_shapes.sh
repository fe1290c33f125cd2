# the fused pair forward at the shapes of the configs, with the kernel trace of the asymmetric case
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python3 $R/tools/bench_pair.py 2>&1 | grep pair_forward
cat > /tmp/pshape.py <<'PY'
import sys, os
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(R, "dv-matcher_amd")); sys.path.insert(0, R)
import numpy as np, torch
from dvm import ops
dev = torch.device("cuda", 0)
wl = ops.deformer_weight_list(dict(np.load(os.path.join(R, "tests", "golden", "deformer_scape_r_weights.npz"))), dev)
B, N, M = 64, 4995, 2200
g = torch.Generator().manual_seed(0)
f1, f2 = torch.randn(B, N, 128, generator=g).to(dev), torch.randn(B, M, 128, generator=g).to(dev)
v1, v2 = torch.rand(B, N, 3, generator=g).to(dev), torch.rand(B, M, 3, generator=g).to(dev)
s = torch.zeros(B, dtype=torch.int32, device=dev)
outs = None
import time
for _ in range(3): outs = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s, s, with_map=False, out=outs)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): outs = ops.pair_forward(wl, f1, f2, v1, v2, 100.0, s, s, with_map=False, out=outs)
torch.cuda.synchronize(); print("pair_forward B=64 4995x2200: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
PY
rocprofv3 --kernel-trace --stats -d /tmp/ps -o x --output-format csv -- python3 /tmp/pshape.py > /tmp/ps.log 2>&1
tail -1 /tmp/ps.log; python3 $R/tools/kstats.py /tmp/ps "" 14
