# round 5: coarse screen vs second form, per-kernel times (rocprofv3 kernel trace), K1 alone: 256 pairs, one direction
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
for route in 2 3; do
  export DVM_K1_ROUTE=$route
  DVM_K1_FLAG_DEBUG=1 timeout 300 python tools/run_softcorr.py 256 5 3 100 2>&1 | grep -E "ms/call|equal|pass B" | tail -3
  rocprofv3 --kernel-trace --stats -d gpurun_out/r5/prof_route$route -o k1 -- python3 tools/run_softcorr.py 256 20 3 100 > /dev/null 2>&1
  python3 tools/kstats.py gpurun_out/r5/prof_route$route 2>/dev/null | head -14
done > gpurun_out/r5/coarse2.txt 2>&1
unset DVM_K1_ROUTE
cat gpurun_out/r5/coarse2.txt
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r5/bench_coarse.json 2> gpurun_out/r5/bench_coarse.err; tail -c 600 gpurun_out/r5/bench_coarse.err; python3 -c "
import json
d=json.loads(open('gpurun_out/r5/bench_coarse.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f step %.2f ms  sweep in step %.3f ms alone %s kernel %s check %s' % (d['value'], d['ms_per_step'], r.get('launch_ms', 0), r.get('standalone'), r['kernel'][:60], d.get('check')))"
