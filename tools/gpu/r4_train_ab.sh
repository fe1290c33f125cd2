cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/train
run() { timeout 600 python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 step %.2f ms  pairs/s %.1f  host %.2f ms' % (d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step']))" | tee -a gpurun_out/train/ab.txt; }
for i in 1 2; do
DVM_FLAT_GRADS=0 run flat0
DVM_FLAT_GRADS=1 run flat1
done
