# the training step after a change of its native nodes: the tests that pin them, the train bench, launch counts, the callers of what is left of ATen
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s
timeout 2400 python -m pytest tests/test_gpu_criterion_native.py tests/test_gpu_train_native.py tests/test_gpu_train_pm.py tests/test_gpu_ddp.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_network.py -m gpu -x -q -k "training_step or backbone_backward" 2>&1 | tail -5
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > gpurun_out/r5s/bench_train.json 2> gpurun_out/r5s/bench_train.err; tail -c 300 gpurun_out/r5s/bench_train.err; cut -c1-330 gpurun_out/r5s/bench_train.json
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5s/prof -o train --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps 4 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r5s/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r5s/kernel_stats_train.csv 2>/dev/null; rm -rf gpurun_out/r5s/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5s/kernel_stats_train.csv')))
steps=6
tot=sum(int(r['Calls']) for r in rows); nat=sum(int(r['Calls']) for r in rows if 'at::' in r['Name'])
print("launches per step %.0f, at::native %.0f, rocclr copy/fill %.0f, kernel ms per step %.2f" % (tot/steps, nat/steps, sum(int(r['Calls']) for r in rows if 'rocclr' in r['Name'])/steps, sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps))
for r in sorted(rows,key=lambda r:-int(r['Calls']))[:22]: print(r['Calls'], '%.1f'%(float(r['TotalDurationNs'])/1e3/int(r['Calls'])), r['Name'][:110])
PY
timeout 900 python tools/gpu/r5_fills.py 3 2>&1 | tail -45
