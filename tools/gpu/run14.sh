cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_linear.py -q 2>&1 | tail -3
python tools/bench_linear_cfg.py 2>&1 | grep -v amdgpu.ids
