cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
python train_driver.py --partial --steps 6 --warmup 2 --batch 2 --points 4995 --points-target 2200 2>&1 | tail -1 | cut -c60-330
python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 | cut -c60-330
cd ..; timeout 600 python -m pytest tests/test_gpu_ddp.py -q -x 2>&1 | tail -2
