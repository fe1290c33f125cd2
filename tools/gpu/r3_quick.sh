# quick check after a K1 change: softcorr / pair parity + a bench line with kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 > gpurun_out/r3/quick_bench.json 2> gpurun_out/r3/quick_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3/quick_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("pairs/s %.0f  ms/step %.2f  sweep in-step %.2f ms  alone %.2f ms  frac %.3f  check %s" % (d["value"], d["ms_per_step"], r["launch_ms"], r.get("standalone", {}).get("launch_ms"), r["frac"], d.get("check", {}).get("ok")))
for k in r.get("kernels", []):
    print("  %-28s %.3f ms/launch  %.2f ms/step  frac %.3f" % (k["kernel"], k["launch_ms"], k.get("ms_per_step", 0), k.get("frac", 0)))
PY
