cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/p_nat -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 4 --warmup 2 --batch 8 --points 2048 > /tmp/p_nat.log 2>&1
mkdir -p $R/gpurun_out/r4ts; cp $(find /tmp/p_nat -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4ts/kstats.csv
cd $R; python3 tools/kstats.py gpurun_out/r4ts/kstats.csv "" 60 | cut -c1-150
DVM_STEP_BREAKDOWN=1 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | grep "host ms"
