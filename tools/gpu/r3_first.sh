# round 3, first call: the -m gpu suite, smoke, the default bench line (new launcher / check / kernels[] legs)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r3/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r3/smoke.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
python bench.py --gpus 2 --backend gloo --pairs 256 --steps 5 --warmup 2 > gpurun_out/r3/bench_gloo2.json 2> gpurun_out/r3/bench_gloo2.err
tail -3 gpurun_out/r3/pytest_gpu.txt; cat gpurun_out/r3/smoke.txt; head -c 600 gpurun_out/r3/bench_default.json
