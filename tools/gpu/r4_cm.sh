cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_network.py tests/test_gpu_ddp.py tests/test_gpu_backward.py -m gpu -q --timeout 900 2>&1 | grep -v amdgpu | tail -4
DVM_CRIT_MERGE=0 python tools/bench_criterion.py 8 2048
python tools/bench_criterion.py 8 2048
DVM_CRIT_MERGE=0 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c60-200
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c60-200
