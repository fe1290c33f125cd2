cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in 2 4 8; do
DVM_MLP_AHEAD=$t python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'mlp' in x['kernel']][0]
print('DVM_MLP_AHEAD=$t mlp %.3f ms  step %.2f ms  pairs/s %.0f check %s' % (k['launch_ms'], d['ms_per_step'], d['value'], d.get('check',{}).get('ok')))"
done
