cd $GRAFT_REPO_ROOT
for q in 1 2 3 4; do for pc in 0 1; do
echo "GPU_MAX_HW_QUEUES=$q DVM_PAIR_CALLS=$pc"
GPU_MAX_HW_QUEUES=$q DVM_PAIR_CALLS=$pc python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c60-200
done; done
for q in 1 2 3; do echo "GPU_MAX_HW_QUEUES=$q bench"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-check 2>/dev/null | cut -c1-160; done
GPU_MAX_HW_QUEUES=2 python tools/bench_train_net.py 8 2048 2
GPU_MAX_HW_QUEUES=2 DVM_CRIT_STREAMS=0 DVM_PAIR_CALLS=0 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c60-200
