cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_network.py tests/test_gpu_backbone.py tests/test_gpu_image_backbone.py -q -x 2>&1 | tail -4
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu.ids | tail -1
