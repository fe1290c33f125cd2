cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_linear.py -q 2>&1 | tail -5
python tools/bench_linear.py 8 2048 20 2>&1 | grep -v amdgpu.ids
python tools/bench_linear.py 1 4995 20 2>&1 | grep -v amdgpu.ids
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids
