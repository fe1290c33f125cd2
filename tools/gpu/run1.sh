cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_linear.py -x -q 2>&1 | tail -15 > gpurun_out/r2_t_linear.log
python -m pytest tests/test_gpu_network.py -q -s 2>&1 | tail -80 > gpurun_out/r2_t_network.log
python tools/bench_linear.py 8 2048 20 > gpurun_out/r2_bench_linear.log 2>&1
python tools/bench_backbone.py 8 2048 10 > gpurun_out/r2_bench_backbone.log 2>&1
python tools/bench_backbone.py 1 4995 10 >> gpurun_out/r2_bench_backbone.log 2>&1
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_network.py --deselect tests/test_gpu_linear.py 2>&1 | tail -15 > gpurun_out/r2_t_rest.log
tail -5 gpurun_out/r2_t_linear.log gpurun_out/r2_t_rest.log; tail -30 gpurun_out/r2_t_network.log
