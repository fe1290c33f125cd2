cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_pm.py -q -x 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py -q -x 2>&1 | tail -8
cd dv-matcher_amd
for l in cm pm; do
DVM_TRAIN_LAYOUT=$l timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1
DVM_TRAIN_LAYOUT=$l timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 --graph 2>&1 | tail -1
done
