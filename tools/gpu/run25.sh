cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_pm.py -q -x 2>&1 | tail -3
(cd dv-matcher_amd; for fz in 0 1; do DVM_FUSE_GRAD_ACC=$fz timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c60-250; done
timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 --graph 2>&1 | tail -1 | cut -c60-250)
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py -q -x 2>&1 | tail -3
