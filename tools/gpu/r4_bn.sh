cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train_native.py tests/test_gpu_train_pm.py tests/test_gpu_network.py -m gpu -q --timeout 900 2>&1 | grep -v amdgpu | tail -4
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c60-200
python tools/bench_train_net.py 8 2048 2
