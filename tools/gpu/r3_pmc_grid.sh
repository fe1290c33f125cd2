# what bounds the grid kernels (xyz kNN, node ring, influence search, Chamfer): SQ counters per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3grid; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/a --output-format csv -- $B > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/b --output-format csv -- $B > $O/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS TA_BUSY_avr -d $O/c --output-format csv -- $B > $O/c.log 2>&1
cd $R
for k in grid_knn_self grid_ring grid_infl grid_chamfer fps_kernel softcorr_exact_rows; do echo "== $k"; python tools/pmc_summary.py $O $k; done > $O/summary.txt 2>&1
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
