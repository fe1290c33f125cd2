# the directional (N != M) criterion node: its tests, the partial-shape training bench before / after (criterion.native_train is the switch)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s
timeout 1500 python -m pytest tests/test_gpu_criterion_native.py -m gpu -x -q 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_network.py -m gpu -x -q -k "partial or sharded or training_step" 2>&1 | tail -4
timeout 600 python bench.py --workload partial --steps 10 --warmup 3 > gpurun_out/r5s/bench_partial.json 2> gpurun_out/r5s/bench_partial.err; tail -c 300 gpurun_out/r5s/bench_partial.err
python3 -c "
import json; d=json.load(open('gpurun_out/r5s/bench_partial.json')); print('partial: %.2f ms/step, host %.2f ms, %s' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], d['last_losses']))"
timeout 600 python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train: %.2f ms/step, host %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"
