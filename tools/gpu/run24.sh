cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c60-330
timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 16 --points 2048 2>&1 | tail -1 | cut -c60-330
timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 2>&1 | tail -1 | cut -c60-330
timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 --graph 2>&1 | tail -1 | cut -c60-330
