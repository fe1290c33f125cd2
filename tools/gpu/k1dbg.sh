cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 6 8 9; do
  rm -rf /tmp/p$d
  DVM_K1_DEBUG=$d rocprofv3 --kernel-trace --stats -d /tmp/p$d -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_softcorr.py 256 3 3 100 > /tmp/p$d.log 2>&1
  f=$(find /tmp/p$d -name "*kernel_stats.csv" | head -1)
  echo "dbg=$d: $(grep sweep $f | awk -F, '{print $2, $4}')"
done
tail -3 /tmp/p0.log
