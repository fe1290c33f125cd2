# pass B of K1, form 4 (64-byte pieces per four lanes + LDS transpose): pieces in flight per lane
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in ${WINS:-2 3 4 8}; do
DVM_K1_REFINE=14 DVM_K1_REFINE_WIN=$w python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'refine' in x['kernel']][0]
print('DVM_K1_REFINE=14 WIN=$w refine %.3f ms  step %.2f ms  check %s' % (k['launch_ms'], d['ms_per_step'], d.get('check',{}).get('ok')))"
done
