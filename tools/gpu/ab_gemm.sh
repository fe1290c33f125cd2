# A/B of two builds of the library on the GEMM shapes, kernel times from rocprofv3 (csrc/libdvm_old.so vs libdvm_hip.so)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cp $R/dv-matcher_amd/csrc/libdvm_hip.so $R/dv-matcher_amd/csrc/libdvm_new.so
for v in old new; do
cp $R/dv-matcher_amd/csrc/libdvm_$v.so $R/dv-matcher_amd/csrc/libdvm_hip.so
rm -rf /tmp/pg_$v
rocprofv3 --kernel-trace -d /tmp/pg_$v -o x --output-format csv -- python3 $R/tools/bench_linear.py 8 2048 20 > /tmp/pg_$v.log 2>&1
echo "== $v"; python3 $R/tools/ktrace.py /tmp/pg_$v "linear_mfma_kernel" 16
done
