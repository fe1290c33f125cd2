"""Print a rocprofv3 `--kernel-trace --stats` kernel_stats.csv: calls, average and total time per kernel, optionally
only kernels matching a substring.  usage: kstats.py <dir-or-csv> [substring] [top]"""
import csv, glob, os, sys
p = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms, %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in [r for r in rows if sub.lower() in r["Name"].lower()][:top]:
    print("%5.1f%% %6d calls  avg %9.1f us  total %9.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / 1e3, r["Name"][:90]))
