# round-4 evidence under gpurun_out/r4p/ (copied into profiles/r4_* by hand): bench line, kernel stats of the bench, PMC passes
# (separate runs, kernel-trace only) for the K1 sweep (HBM bytes + SQ rows at --pairs 512), pass B, the MLP, the grid kernels, FPS;
# the training workloads through bench.py and the kernel stats of the training step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_pairs512.json 2> $O/bench.err
python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python bench.py --workload partial --steps 10 --warmup 3 > $O/bench_partial.json 2> $O/bench_partial.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o x --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_steps3_pairs512.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_train -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 3 --warmup 1 --batch 8 --points 2048 > /tmp/p_train.log 2>&1
cp $(find /tmp/p_train -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_B8_N2048.csv
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc/fetch --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc/write --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc/tcc --output-format csv -- $B > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc/sqa --output-format csv -- $B > $O/pmc_sqa.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/pmc/sqb --output-format csv -- $B > $O/pmc_sqb.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT -d $O/pmc/sqc --output-format csv -- $B > $O/pmc_sqc.log 2>&1
cd $R
python tools/k1_traffic.py $O/pmc $O/bench_pairs512.json $O/k1_traffic.json
for k in softcorr_sweep2 softcorr_refine mlp_f16x2 pool_kernel; do echo "== $k"; python tools/pmc_summary.py $O/pmc $k; done > $O/pmc_summary.txt 2>&1
for k in grid_knn_self grid_ring grid_infl grid_chamfer fps_kernel; do echo "== $k"; python tools/pmc_summary.py $O/pmc $k; done > $O/pmc_grid.txt 2>&1
find $O/pmc -name "*.csv" -size +2M -delete
# the bench line again, now that the traffic file of THIS source exists (it is picked up from profiles/ only: copy first)
cp $O/k1_traffic.json $R/profiles/r4_k1_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_pairs512_with_traffic.json 2>> $O/bench.err
cat $O/pmc_grid.txt | head -60; cut -c1-400 $O/bench_pairs512_with_traffic.json; python tools/kstats.py $O/kernel_stats_bench_steps3_pairs512.csv "" 14
