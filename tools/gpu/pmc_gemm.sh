# SQ counters of dvm_linear_f32 at LG-Net's layer shapes (separate --pmc passes, kernel-trace only)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/a --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_linear.py 8 2048 3 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_linear.py 8 2048 3 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $OUT/c --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_linear.py 8 2048 3 > $OUT/c.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_mfma_kernel<false" in r["Kernel_Name"]:
            acc[(r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda k: -int(k[0]))[:4]:
    print("grid", k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print("   %-28s %.5g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
