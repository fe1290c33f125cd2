cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_train_native.py tests/test_gpu_train_pm.py -m gpu -x -q 2>&1 | tail -15
