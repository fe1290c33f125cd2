cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in ${SCANS:-32 8 16 24 65}; do
DVM_CHAMFER_SCAN_MIN=$t python bench.py --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'chamfer' in x['kernel']][0]
print('DVM_CHAMFER_SCAN_MIN=$t chamfer %.3f ms  step %.2f ms  pairs/s %.0f check %s' % (k['launch_ms'], d['ms_per_step'], d['value'], d.get('check',{}).get('ok')))"
done
