"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per launch of every counter for kernels matching a substring.
usage: pmc_summary.py <dir> <kernel-substring>"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-36s mean/launch %.5g  (n=%d)" % (k, sum(v) / len(v), len(v)))
