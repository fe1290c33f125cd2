# kernel times of the pair forward per (feature set, alpha): rocprofv3 trace of tools/bench_alpha.py split by launch order
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pa; rocprofv3 --kernel-trace -d /tmp/pa -o x --output-format csv -- python3 $R/tools/bench_alpha.py > /tmp/pa.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pa/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# segment: 8 cases x 7 calls each; use the sweep kernel occurrences as call markers
calls = [i for i, r in enumerate(rows) if 'softcorr_sweep_f16_kernel' in r['Kernel_Name']]
names = ['randn a10', 'randn a31', 'randn a33', 'randn a100', 'trained a10', 'trained a31', 'trained a33', 'trained a100']
for c in range(8):
    a = calls[c * 7 + 3]; b = calls[c * 7 + 4]     # one steady-state call: from its sweep to the next sweep
    acc = collections.Counter()
    for r in rows[a:b]:
        acc[r['Kernel_Name'][:60]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    top = ', '.join('%s %.0f' % (k.split('(')[0].split('::')[-1][:28], v) for k, v in acc.most_common(5))
    print('%-13s %s' % (names[c], top))
PY
