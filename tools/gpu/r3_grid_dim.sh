# grid resolution sweep (DVM_GRID_DIM forces one resolution for every grid): bench line + the grid kernels' launch times
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in ${DIMS:-0 8 10 11 13 14}; do
( if [ "$t" != "0" ]; then export DVM_GRID_DIM=$t; fi; python bench.py --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kernel']:x['launch_ms'] for x in d['roofline']['kernels']}
print('DVM_GRID_DIM=$t  step %.2f ms  pairs/s %.0f  chamfer %.3f  knn %.3f  check %s' % (d['ms_per_step'], d['value'], k['grid_chamfer_kernel'], k['grid_knn_self_kernel'], d.get('check',{}).get('ok')))" )
done
