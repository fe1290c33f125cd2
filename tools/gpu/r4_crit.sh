cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_network.py tests/test_gpu_ddp.py -m gpu -q --timeout 900 -k "criterion or training_step or sharded" 2>&1 | tail -3
DVM_CRIT_STREAMS=0 python tools/bench_criterion.py 8 2048
python tools/bench_criterion.py 8 2048
DVM_CRIT_STREAMS=0 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c1-200
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c1-200
DVM_CRIT_STREAMS=0 python dv-matcher_amd/train_driver.py --partial --steps 10 --warmup 3 --batch 2 --points 4995 --points-target 2200 2>/dev/null | tail -1 | cut -c1-200
python dv-matcher_amd/train_driver.py --partial --steps 10 --warmup 3 --batch 2 --points 4995 --points-target 2200 2>/dev/null | tail -1 | cut -c1-200
