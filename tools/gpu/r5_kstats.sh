# round 5: kernel stats of the bench (rocprofv3 kernel trace)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r5
rm -rf /tmp/p_bench
rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o x --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r5/kernel_stats_bench_steps3_pairs512.csv
python3 $R/tools/kstats.py $R/gpurun_out/r5/kernel_stats_bench_steps3_pairs512.csv "" 40
