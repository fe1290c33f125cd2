# training-path tests + step timing (B = 8 x 2048, B = 2 x 1024 eager and graph)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_pm.py tests/test_gpu_network.py tests/test_gpu_ddp.py -q -x 2>&1 | tail -3
cd dv-matcher_amd
python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c60-250
python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 2>&1 | tail -1 | cut -c60-250
python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 --graph 2>&1 | tail -1 | cut -c60-250
cd ..; python tools/bench_backbone.py 8 2048 10 2>&1 | tail -1; python tools/bench_backbone.py 1 4995 10 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_stress.py -q -x 2>&1 | tail -2
