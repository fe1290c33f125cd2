# training-path kernels after a change: criterion / dist-loss / training-step parity, bench line of the training workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_stress.py tests/test_gpu_train_pm.py -x -q 2>&1 | tail -2
python bench.py --workload train 2>/dev/null | cut -c1-420
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_t -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/dv-matcher_amd/train_driver.py --steps 2 --warmup 1 --batch 8 --points 2048 > /tmp/p_t.log 2>&1
grep -E "dist_loss|topk_wave|knn_scores" $(find /tmp/p_t -name "*kernel_stats.csv" | head -1) | cut -c1-200
