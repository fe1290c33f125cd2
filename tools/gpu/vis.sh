# config 5: image-backbone tests, bench, kernel stats
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_image_backbone.py -q -x 2>&1 | tail -3
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_vis -o v --output-format csv -- python3 tools/bench_visual.py 8 4096 2 > gpurun_out/vis.log 2>&1
python3 tools/ktrace.py gpurun_out/prof_vis "adaptive_conv7" 6
