cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 2>&1 | tail -1
