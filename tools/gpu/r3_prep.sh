# the pair path's one-pass preparation (row norms + absmax + fp16 planes): rows per trip (DVM_PREP_ROWS) and against the two-pass
# form (DVM_K1_FUSED_PREP=0), same box; planes checked against the host definition
cd /tmp && export TMPDIR=/tmp
for cfg in "1 1" "1 2" "1 4" "0 2" "1 1" "1 2" "1 4" "0 2"; do
set -- $cfg
rm -rf /tmp/p_b; DVM_K1_FUSED_PREP=$1 DVM_PREP_ROWS=$2 rocprofv3 --kernel-trace --stats -d /tmp/p_b -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_b.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/p_b/**/*kernel_stats.csv', recursive=True)[0]
t = {}
for r in csv.DictReader(open(f)):
    for k in ('rownorm_split', 'rownorm2_k128', 'split_planes_kernel'):
        if k in r['Name']: t[k] = float(r['AverageNs']) / 1e3
print('FUSED=$1 ROWS=$2', ' '.join('%s %.0f us' % kv for kv in sorted(t.items())))
PY
done
cd $GRAFT_REPO_ROOT; for r in 1 2 4; do echo "rows $r: $(DVM_PREP_ROWS=$r python tools/check_planes.py 2>&1 | tail -1)"; done
