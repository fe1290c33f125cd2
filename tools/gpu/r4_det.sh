cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train_native.py tests/test_gpu_train_pm.py tests/test_gpu_backward.py -m gpu -q --timeout 900 2>&1 | tail -8
python tools/bench_train_net.py 8 2048 2
DVM_DETERMINISTIC=1 python tools/bench_train_net.py 8 2048 2
