"""One step's kernels from a rocprofv3 kernel trace in start order: offset from the step's first kernel, duration, queue —
to read the critical path and the overlap of the streams.  usage: ktimeline.py <dir-or-csv> <first-kernel-substring> [step-index] [min-us]"""
import csv, glob, os, sys
p = sys.argv[1]; first = sys.argv[2].lower(); which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"].lower()]
# a step begins at each occurrence of the marker kernel that follows a gap of other kernels
steps = [s for j, s in enumerate(starts) if j == 0 or s - starts[j - 1] > 20]
a = steps[which]; b = steps[which + 1] if which + 1 < len(steps) and which != -1 else len(rows)
t0 = int(rows[a]["Start_Timestamp"]); busy = {}
end = max(int(r["End_Timestamp"]) for r in rows[a:b])
print("step of %d kernels, span %.3f ms" % (b - a, (end - t0) / 1e6))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r["Queue_Id"]; busy[q] = busy.get(q, 0) + (e - s)
    if (e - s) / 1e3 >= min_us:
        print("%9.3f ms  +%8.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, r["Kernel_Name"][:100]))
print({q: round(v / 1e6, 3) for q, v in busy.items()})
