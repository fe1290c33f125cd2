cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 --graph 2>&1 | tail -3 | cut -c1-400
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c1-200
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 --graph 2>&1 | tail -1 | cut -c1-200
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 2 --points 1024 2>&1 | tail -1 | cut -c1-200
