# round-5 evidence under gpurun_out/r5p/ (copied into profiles/r5_* by hand): bench lines of the three workloads, kernel stats of the
# bench and of the training step, PMC passes (separate runs, kernel-trace only) for the K1 screen (HBM bytes + SQ rows at --pairs 512),
# pass B, the MLP, pooling; the strong-scaling proxy (small resident batches on one GPU).
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_pairs512.json 2> $O/bench.err
python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python bench.py --workload partial --steps 10 --warmup 3 > $O/bench_partial.json 2> $O/bench_partial.err
# strong-scaling proxy: what one rank of an 8-GPU strong-scaling run does alone (512 / 8 = 64 pairs), and the points between
for p in 32 64 128 256 512; do
  python bench.py --pairs $p --steps 20 --warmup 5 --cpu-sample 0 --no-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pairs %4d: %8.0f pairs/s  %7.3f ms/step  sweep %.3f ms  graph_cached %8.0f pairs/s' % ($p, d['value'], d['ms_per_step'], r['launch_ms'], d['graph_cached']['value']))"
done > $O/scaling_proxy.txt 2>&1
for b in 1 2 4 8; do
  python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch $b --points 2048 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('train batch %d: %7.1f pairs/s  %7.2f ms/step  host enqueue %.2f ms' % ($b, d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))"
done >> $O/scaling_proxy.txt 2>&1
for b in 1 2 8; do   # the same step captured into one HIP graph and replayed (train_driver --graph): no host work per step
  python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch $b --points 2048 --graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('train batch %d, captured step: %7.1f pairs/s  %7.2f ms/step' % ($b, d['value'], d['ms_per_step']))"
done >> $O/scaling_proxy.txt 2>&1
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
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc/grbm --output-format csv -- $B > $O/pmc_grbm.log 2>&1
cd $R
python tools/k1_traffic.py $O/pmc $O/bench_pairs512.json $O/k1_traffic.json
for k in softcorr_coarse softcorr_refine mlp_f16x2p pool_kernel grid_chamfer fps_kernel; do echo "== $k"; python tools/pmc_summary.py $O/pmc $k; done > $O/pmc_summary.txt 2>&1
find $O/pmc -name "*.csv" -size +2M -delete
# the bench line again, now that the traffic file of THIS source exists (it is picked up from profiles/ only: copy first)
cp $O/k1_traffic.json $R/profiles/r5_k1_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_pairs512_with_traffic.json 2>> $O/bench.err
cat $O/scaling_proxy.txt; cut -c1-400 $O/bench_pairs512_with_traffic.json; python tools/kstats.py $O/kernel_stats_bench_steps3_pairs512.csv "" 14; python tools/kstats.py $O/kernel_stats_train_B8_N2048.csv "" 25
