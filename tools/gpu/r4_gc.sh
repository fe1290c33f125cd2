cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ddp.py -m gpu -q --timeout 900 -k "graph_cache or partial_mode or hip_graph or full_loop" 2>&1 | tail -3
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>/dev/null | tail -1 | cut -c1-200
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 --graph-cache 2>/dev/null | tail -1 | cut -c1-200
