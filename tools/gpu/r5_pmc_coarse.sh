# round 5: SQ counters of the coarse screen alone (K1 only, 256 pairs one direction; separate --pmc passes, kernel trace only)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/pmc_coarse; rm -rf $O; mkdir -p $O
export DVM_K1_ROUTE=3
B="python3 $R/tools/run_softcorr.py 256 4 3 100"
cd $R
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/a --output-format csv -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $O/b --output-format csv -- $B > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $O/c --output-format csv -- $B > $O/c.log 2>&1
for k in softcorr_coarse softcorr_refine; do echo "== $k"; python3 tools/pmc_summary.py $O $k; done > $R/gpurun_out/r5/pmc_coarse.txt 2>&1
find $O -name "*.csv" -size +1M -delete
cat $R/gpurun_out/r5/pmc_coarse.txt; tail -3 $O/c.log
