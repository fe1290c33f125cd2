# round 5: coarse screen, keys per tile 64 vs 128 (DVM_K1_COARSE_KT): kernel time + stamps, then the K1 parity slice with the route forced
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
export DVM_K1_ROUTE=3
for kt in 64 128; do
  echo "== DVM_K1_COARSE_KT=$kt"
  rm -rf /tmp/prof_c
  DVM_K1_COARSE_KT=$kt rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o k1 --output-format csv -- python3 tools/run_softcorr.py 256 20 3 100 2>&1 | grep -E "ms/call|equal"
  python3 tools/kstats.py /tmp/prof_c "" 8 2>/dev/null | grep -E "coarse"
  DVM_K1_COARSE_KT=$kt DVM_K1_STAMPS=1 timeout 300 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps" | tail -1
done > gpurun_out/r5/coarse7.txt 2>&1
cat gpurun_out/r5/coarse7.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(softcorr or argmin or pair_forward) and not probe_routes" 2>&1 | tail -3
