cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1b
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $OUT/a --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_softcorr.py 256 2 3 100 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_softcorr.py 256 2 3 100 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/c --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/run_softcorr.py 256 2 3 100 > $OUT/c.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT softcorr_sweep_f16; tail -2 $OUT/a.log
