# round 5: after the second form's removal: parity with the route forced, flag rates of the coarse screen on the trained-like set, the hard map
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
{
python - <<'PY'
import os, sys, subprocess
sys.path.insert(0, "dv-matcher_amd")
code = """
import sys, torch, time
sys.path.insert(0, 'dv-matcher_amd')
from dvm import ops
g = torch.Generator().manual_seed(3)
f1, f2 = torch.randn(64, 2048, 128, generator=g).cuda(), torch.randn(64, 2048, 128, generator=g).cuda()
if sys.argv[1] == 'trained': f1, f2 = 0.3 * torch.relu(f1), 0.3 * torch.relu(f2)
for alpha in (33.0, 50.0, 70.0, 100.0, 150.0, 250.0):
    for _ in range(2): ops.softcorr(f1, f2, alpha, topk=10, variant=3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ops.softcorr(f1, f2, alpha, topk=10, variant=3)
    torch.cuda.synchronize(); print('%s alpha %g: %.3f ms per call (64 pairs, one direction)' % (sys.argv[1], alpha, (time.perf_counter() - t) / 5 * 1e3), flush=True)
"""
for kind in ("randn", "trained"):
    for route in ("", "3", "1"):
        env = dict(os.environ, DVM_K1_FLAG_DEBUG="1", DVM_K1_ROUTE_DEBUG="1")
        if route: env["DVM_K1_ROUTE"] = route
        r = subprocess.run([sys.executable, "-c", code, kind], env=env, capture_output=True, text=True)
        print("==", kind, "route", route or "probe")
        lines = (r.stdout + r.stderr).splitlines()
        seen = set()
        for ln in lines:
            if ("ms per call" in ln) or (("pass B" in ln or "K1 routes" in ln) and ln not in seen):
                print("  ", ln); seen.add(ln)
PY
} > gpurun_out/r5/coarse4.txt 2>&1
cat gpurun_out/r5/coarse4.txt
timeout 2400 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r5/coarse4_tests.txt
