: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py tests/test_gpu_criterion_native.py -x -q 2>&1 | tail -2
for p in 32 64 512; do
  timeout 300 python bench.py --pairs $p --steps 40 --warmup 6 --cpu-sample 0 --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs %4d: %7.3f ms/step' % ($p, d['ms_per_step']))"
done
