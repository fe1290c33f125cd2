# Deformer MLP, ablations (WRONG results, timing only): what the kernel costs without its weight stream / its LDS operand reads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in ${ABL:-0 1 2 3}; do
DVM_MLP_ABLATE=$a python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'mlp' in x['kernel']][0]
print('DVM_MLP_ABLATE=$a mlp %.3f ms  step %.2f ms' % (k['launch_ms'], d['ms_per_step']))"
done
