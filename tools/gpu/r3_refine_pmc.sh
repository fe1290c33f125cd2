# what bounds pass B (softcorr_refine_kernel): texture-address / L1 counters per form (DVM_K1_REFINE)
# (a pass with TA_ADDR_STALLED_BY_TC_CYCLES / TA_DATA_STALLED_BY_TC_CYCLES / TA_FLAT_READ_WAVEFRONTS hung the profiler: not collected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ref; mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check"
for f in ${FORMS:-2}; do
export DVM_K1_REFINE=$f
timeout 200 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE -d $O/f$f/ta --output-format csv -- $B > $O/f${f}_ta.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O/f$f/tcp --output-format csv -- $B > $O/f${f}_tcp.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM -d $O/f$f/sq --output-format csv -- $B > $O/f${f}_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/f$f/tcc --output-format csv -- $B > $O/f${f}_tcc.log 2>&1
echo "== DVM_K1_REFINE=$f"; cd $R; python tools/pmc_summary.py $O/f$f softcorr_refine; cd /tmp
done > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
