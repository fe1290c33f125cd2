# round 3: the second form of the K1 sweep.  (1) the ubench that decides whether vector instructions hide in the gaps of a
# matrix chain, (2) parity tests + bench under every DVM_K1_SWEEP form, one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
./tools/ubench/mfma_valu_gap > gpurun_out/r3/ubench_gap.txt 2>&1
for f in 1 2 3 0; do
  export DVM_K1_SWEEP=$f
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "softcorr or argmin or pair" 2>&1 | tail -4 > gpurun_out/r3/k1_form${f}_tests.txt
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > gpurun_out/r3/k1_form${f}_bench.json 2> gpurun_out/r3/k1_form${f}_bench.err
  echo "form $f: $(tail -1 gpurun_out/r3/k1_form${f}_tests.txt)"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/k1_form${f}_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("   pairs/s %.0f  ms/step %.2f  sweep in-step %.2f ms  alone %.2f ms  frac %.3f  check %s" % (d["value"], d["ms_per_step"], r["launch_ms"], r["standalone"]["launch_ms"], r["frac"], d["check"]))
PY
done
tail -60 gpurun_out/r3/ubench_gap.txt
