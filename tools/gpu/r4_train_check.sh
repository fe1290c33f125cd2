cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/train
timeout 1500 python -m pytest tests/test_gpu_train_native.py tests/test_gpu_network.py tests/test_gpu_ddp.py -m gpu -q --timeout 900 -x 2>&1 | tail -3
for w in "--batch 8 --points 2048" "--partial --batch 2 --points 4995 --points-target 2200"; do
echo "== $w"
DVM_STEP_BREAKDOWN=1 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 $w 2>&1 | grep -v "amdgpu\|Warning\|run_backward" | tail -2 | cut -c1-330
done
