cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/train
for w in "--batch 8 --points 2048" "--partial --batch 2 --points 4995 --points-target 2200"; do
echo "== $w"
DVM_STEP_BREAKDOWN=1 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 $w 2>&1 | grep -v amdgpu | tail -4 | cut -c1-900
done
