# round 5: the gate behind the coarse screen: forced coarse on inputs it serves badly must now cost about coarse + lean; full parity file; bench
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
{
python - <<'PY'
import os, sys, subprocess
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
    for route in ("", "3"):
        env = dict(os.environ)
        if route: env["DVM_K1_ROUTE"] = route
        r = subprocess.run([sys.executable, "-c", code, kind], env=env, capture_output=True, text=True)
        print("==", kind, "route", route or "probe")
        for ln in (r.stdout + r.stderr).splitlines():
            if "ms per call" in ln: print("  ", ln)
PY
} > gpurun_out/r5/coarse5.txt 2>&1
cat gpurun_out/r5/coarse5.txt
timeout 2400 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r5/coarse5_tests.txt
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r5/bench_coarse5.json 2> gpurun_out/r5/bench_coarse5.err; tail -c 300 gpurun_out/r5/bench_coarse5.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_coarse5.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f step %.2f ms  sweep in step %.3f ms alone %s kernel %s check %s' % (d['value'], d['ms_per_step'], r.get('launch_ms', 0), r.get('standalone'), r['kernel'][:60], d.get('check')))
for k in r['kernels']: print(k['kernel'], round(k['launch_ms'],3))"
