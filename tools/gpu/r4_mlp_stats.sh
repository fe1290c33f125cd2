# kernel stats of the bench with the persistent MLP and with the kernel it replaces (serialised by the profiler)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mlp; mkdir -p $O
for v in 1 0; do
export DVM_MLP_PERSIST=$v
rocprofv3 --kernel-trace --stats -d /tmp/p_bench$v -o x --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_bench$v.log 2>&1
cp $(find /tmp/p_bench$v -name "*kernel_stats.csv" | head -1) $O/kstats_persist$v.csv
echo "== DVM_MLP_PERSIST=$v"; python $R/tools/kstats.py $O/kstats_persist$v.csv "" 40 | grep -i "mlp\|assemble\|pack\|split\|TOTAL\|total" 
done
