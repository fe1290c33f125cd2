"""Group a rocprofv3 kernel trace by (kernel, grid): calls and average duration per launch shape — what `--stats` hides when
one kernel runs at several sizes.  usage: ktrace.py <dir-or-csv> [substring] [top]"""
import collections, csv, glob, os, sys
p = sys.argv[1]
sub = sys.argv[2].lower() if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"].lower():
        continue
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
    k = (r["Kernel_Name"][:70], grid, wg)
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (name, grid, wg), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%6d calls  avg %8.1f us  total %9.1f us  grid %-16s wg %4d  %s" % (n, t / n, t, "x".join(map(str, grid)), wg, name))
