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
    m = sum(v) / len(v)
    # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 B: doubled, as
    # /opt/skills/guides/MI355X_MICROARCH.md prescribes (the same correction tools/k1_traffic.py applies)
    extra = "  = %.4g GB fetched (x 1024 x 2)" % (m * 2048 / 1e9) if k == "FETCH_SIZE" else "  = %.4g GB written (x 1024)" % (m * 1024 / 1e9) if k == "WRITE_SIZE" else ""
    print("%-36s mean/launch %.5g  (n=%d)%s" % (k, m, len(v), extra))
if "TCC_HIT_sum" in acc and "TCC_MISS_sum" in acc:
    h, ms = sum(acc["TCC_HIT_sum"]) / len(acc["TCC_HIT_sum"]), sum(acc["TCC_MISS_sum"]) / len(acc["TCC_MISS_sum"])
    print("%-36s %.3f" % ("L2 hit rate (TCC_HIT / (HIT + MISS))", h / (h + ms)))
