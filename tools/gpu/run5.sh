cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/bench_linear_cfg.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_linear.py -q --durations=5 2>&1 | tail -12
