# grid kernels (xyz kNN, ring, influence, Chamfer) after a change: geometry parity + bench line with the kernel table + kernel stats
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s %.0f  ms/step %.2f  check %s' % (d['value'], d['ms_per_step'], d.get('check',{}).get('ok')))
for k in d['roofline']['kernels']: print('  %-28s %.3f ms/launch' % (k['kernel'], k['launch_ms']))"
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/p_b -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_b.log 2>&1
head -16 $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) | cut -c1-150
