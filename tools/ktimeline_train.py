"""One TRAINING step from a rocprofv3 kernel trace: where the wall time of the step goes — the union of the kernels' intervals (GPU busy),
the idle gaps between them (nothing on any queue: launch latency, host, dependencies), the busy time per queue, and the step in 0.5-ms slices
(kernels running, dominant kernel).  A step = the kernels between two optimizer launches (multi_tensor_apply).
usage: ktimeline_train.py <dir-or-csv> [step-index] [first-kernel-substring]   (with a substring: a step BEGINS at that kernel — the pair bench)"""
import collections, csv, glob, os, sys
p = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:
    st = [i for i, r in enumerate(rows) if sys.argv[3].lower() in r["Kernel_Name"].lower()]
    st = [i for j, i in enumerate(st) if j == 0 or i - st[j - 1] > 20]
    a, b = st[which - 1], st[which]
else:
    adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
    ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] - i > 20]      # the last optimizer launch of every step
    a, b = ends[which - 1] + 1, ends[which] + 1
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows[a:b]]
t0, t1 = ks[0][0], max(k[1] for k in ks)
print("step of %d kernels, span %.3f ms, sum of kernel times %.3f ms" % (len(ks), (t1 - t0) / 1e6, sum(k[1] - k[0] for k in ks) / 1e6))
ev = sorted([(k[0], 1) for k in ks] + [(k[1], -1) for k in ks])
busy = over2 = 0; depth = 0; last = t0; gaps = []
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: over2 += t - last
    if depth == 0 and t - last > 0: gaps.append((t - last, last))
    depth += d; last = t
print("GPU busy (union) %.3f ms, two or more kernels at once %.3f ms, idle %.3f ms in %d gaps" % (busy / 1e6, over2 / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _ in gaps: hist["<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else ">=100us"] += g
print("idle by gap length (ms):", {k: round(v / 1e6, 3) for k, v in hist.items()})
for g, at in sorted(gaps, reverse=True)[:12]:
    before = max((k for k in ks if k[1] <= at + 1), key=lambda k: k[1], default=None)
    after = min((k for k in ks if k[0] >= at + g - 1), key=lambda k: k[0], default=None)
    print("  gap %7.1f us at %7.3f ms  after %-45s before %s" % (g / 1e3, (at - t0) / 1e6, before[3][:45] if before else "-", after[3][:60] if after else "-"))
q = collections.Counter()
for k in ks: q[k[2]] += k[1] - k[0]
print("busy per queue (ms):", {k: round(v / 1e6, 3) for k, v in q.items()})
sl = 500000
for s in range(t0, t1, sl):
    inside = [(min(k[1], s + sl) - max(k[0], s), k[3]) for k in ks if k[1] > s and k[0] < s + sl]
    tot = sum(x[0] for x in inside)
    top = collections.Counter()
    for d, n in inside: top[n.split("(")[0][-48:]] += d
    n1, d1 = top.most_common(1)[0] if top else ("-", 0)
    print("%6.1f ms  kernel time in slice %5.0f us  (%d kernels)  top: %-48s %4.0f us" % ((s - t0) / 1e6, tot / 1e3, len(inside), n1, d1 / 1e3))
