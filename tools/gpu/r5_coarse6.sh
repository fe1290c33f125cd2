# round 5: coarse screen forms (DVM_K1_COARSE_FORM: 0 = 8 waves x 2 blocks paced, 1 = unpaced, 2 = 8 x 1, 3 = 16 waves x 1 block): kernel time + stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
export DVM_K1_ROUTE=3
for form in 4 6; do
  echo "== DVM_K1_COARSE_FORM=$form"
  rm -rf /tmp/prof_c
  DVM_K1_COARSE_FORM=$form rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o k1 --output-format csv -- python3 tools/run_softcorr.py 256 20 3 100 2>&1 | grep -E "ms/call|equal"
  python3 tools/kstats.py /tmp/prof_c "" 8 2>/dev/null | grep -E "coarse"
  DVM_K1_COARSE_FORM=$form DVM_K1_STAMPS=1 timeout 300 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps" | tail -1
done > gpurun_out/r5/coarse6.txt 2>&1
cat gpurun_out/r5/coarse6.txt
