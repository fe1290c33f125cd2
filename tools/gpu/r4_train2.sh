# round 4: native training node: tests + kernel statistics of the step, native vs autograd path
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4t2; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_train_native.py -m gpu -q --timeout 900 > $O/native.log 2>&1; echo "rc=$?" >> $O/native.log
tail -12 $O/native.log
python -m pytest tests/test_gpu_network.py -m gpu -q --timeout 900 > $O/train_tests.log 2>&1; echo "rc=$?" >> $O/train_tests.log
tail -8 $O/train_tests.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/p_nat -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 4 --warmup 2 --batch 8 --points 2048 > /tmp/p_nat.log 2>&1
cp $(find /tmp/p_nat -name "*kernel_stats.csv" | head -1) $R/$O/kstats_native.csv
cp $(find /tmp/p_nat -name "*kernel_trace.csv" | head -1) $R/$O/ktrace_native.csv
DVM_NATIVE_TRAIN=0 rocprofv3 --kernel-trace --stats -d /tmp/p_py -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 4 --warmup 2 --batch 8 --points 2048 > /tmp/p_py.log 2>&1
cp $(find /tmp/p_py -name "*kernel_stats.csv" | head -1) $R/$O/kstats_python.csv
cd $R
python3 tools/kstats.py $O/kstats_native.csv "" 45
python3 tools/kstats.py $O/kstats_python.csv "" 12
gzip $O/ktrace_native.csv
