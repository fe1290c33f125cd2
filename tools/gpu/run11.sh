cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_network.py tests/test_gpu_stress.py -q -x 2>&1 | tail -6
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids
DVM_KNN_FUSED=0 python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids
python tools/bench_backbone.py 1 4995 10 2>&1 | grep -v amdgpu.ids
DVM_KNN_FUSED=0 python tools/bench_backbone.py 1 4995 10 2>&1 | grep -v amdgpu.ids
python tools/bench_ops.py 2>&1 | grep -i knn | head
