# SQ counters of the K1 sweep (separate --pmc pass, kernel-trace only), bench at --pairs 256
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/a --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --pairs 256 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --pairs 256 > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT softcorr_sweep_f16
