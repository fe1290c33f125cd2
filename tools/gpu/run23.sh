cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_pm.py -q -x 2>&1 | tail -3
for bs in 0 1; do
echo "branch streams $bs"
DVM_BRANCH_STREAMS=$bs python tools/bench_backbone.py 8 2048 2>&1 | tail -1
DVM_BRANCH_STREAMS=$bs python tools/bench_backbone.py 1 4995 2>&1 | tail -1
(cd dv-matcher_amd; for ts in 0 1; do DVM_BRANCH_STREAMS=$bs DVM_TWO_STREAMS=$ts timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c60-200; done)
done
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py tests/test_gpu_backbone.py -q -x 2>&1 | tail -3
