# round 5: coarse screen A/B (DVM_K1_COARSE_PACE, DVM_K1_COARSE_QB): kernel time from rocprofv3's kernel trace, K1 alone, 256 pairs one direction
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
export DVM_K1_ROUTE=3
for cfg in "DVM_K1_COARSE_PACE=1" "DVM_K1_COARSE_PACE=0" "DVM_K1_COARSE_PACE=1 DVM_K1_COARSE_QB=1" "DVM_K1_COARSE_PACE=0 DVM_K1_COARSE_QB=1"; do
  echo "== $cfg"
  rm -rf /tmp/prof_c
  env $cfg rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o k1 --output-format csv -- python3 tools/run_softcorr.py 256 20 3 100 2>&1 | grep -E "ms/call|equal"
  python3 tools/kstats.py /tmp/prof_c "" 8 2>/dev/null | grep -E "coarse|refine|exact_rows|rownorm|total"
  env $cfg DVM_K1_STAMPS=1 timeout 300 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps" | tail -1
done > gpurun_out/r5/coarse3.txt 2>&1
cat gpurun_out/r5/coarse3.txt
