# round 6: the K1 screen's PMC passes (HBM bytes, L2 hit rate, SQ rows) at --pairs 512 after a change of its source -> profiles/r6_k1_traffic.json
# with provenance, then the bench line that picks it up; the K1 parity slice first.
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; rm -rf "$O"; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py -x -q -k "softcorr or k1 or pair_forward_full or argmin" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_pairs512.json 2> $O/bench.err
cd /tmp
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
cp $O/k1_traffic.json $R/profiles/r6_k1_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_pairs512_with_traffic.json 2>> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench_pairs512_with_traffic.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step','single_call')}); print({k: r[k] for k in r if k not in ('kernels',)})"
head -20 $O/pmc_summary.txt
