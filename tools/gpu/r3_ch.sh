# grid Chamfer after a change: geometry parity + bench line with the kernel table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -k "chamfer or deformer or pair or grid" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s %.0f  ms/step %.2f  check %s' % (d['value'], d['ms_per_step'], d.get('check',{}).get('ok')))
for k in d['roofline']['kernels']: print('  %-28s %.3f ms/launch' % (k['kernel'], k['launch_ms']))"
