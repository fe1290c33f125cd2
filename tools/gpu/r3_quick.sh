# quick check after a kernel change: parity + stress suites, then the bench line with its per-kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for i in 1 2 3; do python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_M or single_row" 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_network.py -m gpu -q -x -s -k "noise_against_float64" 2>&1 | grep -vE "^$|warning|Warning" | tail -40
python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r3/quick_bench.json 2> gpurun_out/r3/quick_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3/quick_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("pairs/s %.0f  ms/step %.2f  sweep in-step %.2f ms  alone %.2f ms  frac %.3f / %.3f  check %s" % (d["value"], d["ms_per_step"], r["launch_ms"], r["standalone"]["launch_ms"], r["frac"], r["standalone"]["frac"], d["check"]["ok"]))
for k in r["kernels"]: print("  %-28s %.3f ms/launch  %.2f ms/step  frac %.3f" % (k["kernel"],k["launch_ms"],k["ms_per_step"],k["frac"]))
PY
for a in 2 4 8; do DVM_MLP_AHEAD=$a python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'mlp' in x['kernel']][0]
print('MLP weights $a k-steps ahead: %.3f ms/launch, step %.2f ms' % (k['launch_ms'], d['ms_per_step']))"; done
