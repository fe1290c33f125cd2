cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vis -o vis -- python3 $GRAFT_REPO_ROOT/tools/bench_visual.py 8 4096 2 > $GRAFT_REPO_ROOT/gpurun_out/r2_prof_vis.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/r2_prof_vis.log
