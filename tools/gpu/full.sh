# the -m gpu suite (with the slowest tests listed) + __graft_entry__.smoke()
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -45 | tee gpurun_out/r3/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r3/smoke.txt
