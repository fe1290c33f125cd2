# backbone after a change: linear + network parity (native forward == Python path), inference forward timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_linear.py tests/test_gpu_network.py -x -q 2>&1 | tail -3
python tools/bench_native_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/bench_native_fwd.txt
python tools/bench_backbone.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/bench_backbone.txt | tail -3
