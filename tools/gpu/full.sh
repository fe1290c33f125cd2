# the whole GPU suite + the bench line (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/full/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; tail -c 300 gpurun_out/full/bench.err; cut -c1-900 gpurun_out/full/bench.json
