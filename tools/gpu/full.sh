# the whole -m gpu suite + smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/full/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/full/gpu_tests.log
tail -6 gpurun_out/full/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
