: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
X_HX=1 timeout 900 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py -x -q -k "softcorr or k1 or pair_forward_full" 2>&1 | tail -2
for x in 0 1; do
  rm -rf /tmp/px$x
  X_HX=$x rocprofv3 --kernel-trace --stats -d /tmp/px$x -o b --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check --pairs 64 > /tmp/bx.log 2>&1
  echo "X_HX=$x pairs 64"; python3 tools/kstats.py $(find /tmp/px$x -name "*kernel_stats.csv" | head -1) exact_rows 3
  rm -rf /tmp/px$x
  X_HX=$x rocprofv3 --kernel-trace --stats -d /tmp/px$x -o b --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check --pairs 512 > /tmp/bx.log 2>&1
  echo "X_HX=$x pairs 512"; python3 tools/kstats.py $(find /tmp/px$x -name "*kernel_stats.csv" | head -1) exact_rows 3
done
for x in 0 1; do for p in 32 64; do
  X_HX=$x timeout 300 python bench.py --pairs $p --steps 40 --warmup 6 --cpu-sample 0 --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('X_HX=$x pairs %4d: %7.3f ms/step' % ($p, d['ms_per_step']))"
done; done
