# round 5: pair-forward parity + the bench at the strong-scaling proxy sizes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_state.py -m gpu -q -x -k "pair or abi or thread or stream" 2>&1 | tail -3
for p in 64 128 512; do
  python bench.py --pairs $p --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pairs %4d: %8.0f pairs/s  %7.3f ms/step  sweep %.3f ms  graph_cached %8.0f pairs/s  check %s' % ($p, d['value'], d['ms_per_step'], r['launch_ms'], d['graph_cached']['value'], d['check']['ok']))"
done | tee gpurun_out/r5/pairs_after_side2.txt
