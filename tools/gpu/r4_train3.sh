cd $GRAFT_REPO_ROOT; O=gpurun_out/r4t3; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_train_native.py -m gpu -q --timeout 900 > $O/native.log 2>&1; echo "rc=$?" >> $O/native.log
tail -6 $O/native.log
python tools/bench_train_net.py 8 2048 2
DVM_STEP_BREAKDOWN=1 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 > $O/breakdown.log 2>&1; grep "host ms" $O/breakdown.log; tail -1 $O/breakdown.log | cut -c1-200
DVM_FUSED_ADAM=0 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/p_nat -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 4 --warmup 2 --batch 8 --points 2048 > /tmp/p_nat.log 2>&1
cp $(find /tmp/p_nat -name "*kernel_stats.csv" | head -1) $R/$O/kstats_native.csv
cd $R; python3 tools/kstats.py $O/kstats_native.csv "" 70 | cut -c1-150
