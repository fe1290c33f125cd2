# round 3: the second form of the K1 sweep: parity tests + bench + cycle stamps under DVM_K1_SWEEP forms (0 = first form)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
FORMS=${FORMS:-"1 2 0"}
for f in $FORMS; do
  export DVM_K1_SWEEP=$f
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "softcorr or argmin or pair" 2>&1 | tail -8 > gpurun_out/r3/k1_form${f}_tests.txt
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > gpurun_out/r3/k1_form${f}_bench.json 2> gpurun_out/r3/k1_form${f}_bench.err
  echo "form $f: $(tail -1 gpurun_out/r3/k1_form${f}_tests.txt)"
  grep -E "FAILED|Error" gpurun_out/r3/k1_form${f}_tests.txt | head -5
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3/k1_form${f}_bench.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("   pairs/s %.0f  ms/step %.2f  sweep in-step %.2f ms  alone %.2f ms  frac %.3f  check %s" % (d["value"], d["ms_per_step"], r["launch_ms"], r["standalone"]["launch_ms"], r["frac"], d["check"]))
except Exception as e:
    print("   bench failed:", e, open("gpurun_out/r3/k1_form${f}_bench.err").read()[-600:])
PY
  if [ "$f" != "0" ]; then DVM_K1_STAMPS=1 python tools/run_softcorr.py 256 1 3 100 2>&1 | grep -E "K1 stamps|ms/call|flagged|equal" | tail -4; fi
done 2>&1 | tee gpurun_out/r3/k1_forms_summary.txt
