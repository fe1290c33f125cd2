cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_linear.py -q 2>&1 | tail -15 > gpurun_out/r2_t_linear.log
python -m pytest tests/test_gpu_network.py -q -s -rA 2>&1 > gpurun_out/r2_t_network.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bb -o bb -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py 8 2048 10 > $GRAFT_REPO_ROOT/gpurun_out/r2_prof_bb.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r2_t_linear.log; grep -n "worst grad\|flipped neighbour\|e2e maps\|passed\|failed\|AssertionError" gpurun_out/r2_t_network.log | head -40
