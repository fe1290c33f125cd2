# the pair forward across the alpha schedule on both synthetic feature sets, the eval forward, config 5 (end of round 4)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4p
ALPHAS=10,33,100 ITERS=10 python tools/bench_alpha.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4p/bench_alpha.txt
(python tools/bench_backbone.py 8 2048 50; python tools/bench_backbone.py 1 4995 50) 2>&1 | grep -v amdgpu | tee gpurun_out/r4p/backbone.txt
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r4p/visual.txt
