cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "softcorr or pair_forward" 2>&1 | tail -2
for f in ${FORMS:-0 3 0 3}; do
DVM_K1_REFINE=$f python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'refine' in x['kernel']][0]
print('DVM_K1_REFINE=$f refine %.3f ms  step %.2f ms  check %s' % (k['launch_ms'], d['ms_per_step'], d.get('check',{}).get('ok')))"
done
