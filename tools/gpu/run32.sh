cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_image_backbone.py -q -x 2>&1 | tail -3
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu.ids
