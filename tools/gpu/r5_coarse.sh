# round 5: the coarse screen (one fp16 plane) against the second form: K1 alone, 256 pairs one direction, then the softcorr parity tests with the route forced
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
{
for route in 2 3; do
  echo "== DVM_K1_ROUTE=$route"
  DVM_K1_ROUTE=$route DVM_K1_FLAG_DEBUG=1 timeout 300 python tools/run_softcorr.py 256 5 3 100 2>&1 | grep -E "ms/call|equal|pass B" | tail -4
done
echo "== stamps, QB=2"
DVM_K1_ROUTE=3 DVM_K1_STAMPS=1 timeout 300 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps" | tail -1
echo "== QB=1"
DVM_K1_ROUTE=3 DVM_K1_COARSE_QB=1 timeout 300 python tools/run_softcorr.py 256 5 3 100 2>&1 | grep -E "ms/call|equal" | tail -2
DVM_K1_ROUTE=3 DVM_K1_COARSE_QB=1 DVM_K1_STAMPS=1 timeout 300 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps" | tail -1
echo "== trained-like (scale irrelevant: randn), alpha 33"
DVM_K1_ROUTE=3 DVM_K1_FLAG_DEBUG=1 timeout 300 python tools/run_softcorr.py 256 3 3 33 2>&1 | grep -E "ms/call|equal|pass B" | tail -3
} > gpurun_out/r5/coarse1.txt 2>&1
cat gpurun_out/r5/coarse1.txt
DVM_K1_ROUTE=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "softcorr and not refine_forms and not probe_routes" 2>&1 | tail -15 | tee gpurun_out/r5/coarse1_tests.txt
