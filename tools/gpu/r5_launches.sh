# launches PER STEP of the training bench, without the one-time launches (model upload, optimizer state): two kernel-stat runs of 4 and 12 steps, differenced
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5s; mkdir -p $O
for K in 4 12; do
  rm -rf /tmp/pl$K
  rocprofv3 --kernel-trace --stats -d /tmp/pl$K -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps $K --warmup 2 > /tmp/pl$K.log 2>&1
  cp $(find /tmp/pl$K -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_steps$K.csv
done
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r5s/train_launches.txt
import csv, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5s/"
def load(k):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(O + "kernel_stats_train_steps%d.csv" % k))}
a, b = load(4), load(12)
names = set(a) | set(b)
per = {n: ((b.get(n, (0, 0))[0] - a.get(n, (0, 0))[0]) / 8.0, (b.get(n, (0, 0))[1] - a.get(n, (0, 0))[1]) / 8e3) for n in names}
tot = sum(v[0] for v in per.values()); nat = sum(v[0] for n, v in per.items() if "at::" in n); roc = sum(v[0] for n, v in per.items() if "rocclr" in n)
print("per training step (B = 8, N = 2048; (12-step run - 4-step run) / 8): %.0f launches, of them at::native %.0f, rocclr copy / fill %.0f; kernel time %.2f ms"
      % (tot, nat, roc, sum(v[1] for v in per.values()) / 1e3))
for n, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:40]:
    if v[0] > 0: print("%6.1f launches  %8.1f us  %s" % (v[0], v[1], n[:120]))
PY
