cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train_native.py -m gpu -q --timeout 900 2>&1 | grep -v amdgpu | tail -4
python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | cut -c1-200
python bench.py --workload partial --steps 10 --warmup 3 2>/dev/null | cut -c1-200
python dv-matcher_amd/train_driver.py --steps 8 --warmup 2 --batch 2 --points 1024 2>/dev/null | tail -1 | cut -c60-200
DVM_PAIR_CALLS=0 python dv-matcher_amd/train_driver.py --steps 8 --warmup 2 --batch 2 --points 1024 2>/dev/null | tail -1 | cut -c60-200
