# A/B of two builds on the training operators (kernel times by launch shape): csrc/libdvm_old.so vs libdvm_hip.so
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cp $R/dv-matcher_amd/csrc/libdvm_hip.so $R/dv-matcher_amd/csrc/libdvm_new.so
for v in old new; do
cp $R/dv-matcher_amd/csrc/libdvm_$v.so $R/dv-matcher_amd/csrc/libdvm_hip.so
rm -rf /tmp/pt_$v
rocprofv3 --kernel-trace -d /tmp/pt_$v -o x --output-format csv -- python3 $R/tools/bench_train_ops.py 8 2048 10 > /tmp/pt_$v.log 2>&1
echo "== $v"; python3 $R/tools/ktrace.py /tmp/pt_$v "${1:-wgrad}" 14
done
