cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_pm.py -q -x 2>&1 | tail -3
timeout 300 python tools/bench_train_ops.py 2>&1 | tail -6
cd dv-matcher_amd
for ts in 0 1; do for l in cm pm; do
echo "layout $l two_streams $ts"
DVM_TWO_STREAMS=$ts DVM_TRAIN_LAYOUT=$l timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c1-200
done; done
