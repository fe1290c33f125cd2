cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_linear.py -q 2>&1 | tail -15 > gpurun_out/r2_t_linear.log
python -m pytest tests/test_gpu_network.py -q -s -rA 2>&1 > gpurun_out/r2_t_network.log
python tools/bench_linear.py 8 2048 20 > gpurun_out/r2_bench_linear.log 2>&1
python tools/bench_backbone.py 8 2048 10 graph > gpurun_out/r2_bench_backbone.log 2>&1
python tools/bench_backbone.py 1 4995 10 graph >> gpurun_out/r2_bench_backbone.log 2>&1
tail -3 gpurun_out/r2_t_linear.log; grep -n "teacher-forced:\|worst grad\|flipped neighbour\|e2e maps\|passed\|failed\|AssertionError" gpurun_out/r2_t_network.log | head -40
cat gpurun_out/r2_bench_linear.log gpurun_out/r2_bench_backbone.log
