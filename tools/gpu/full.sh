# the whole GPU suite + the bench line (what the driver runs at round end)
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/full/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; tail -c 300 gpurun_out/full/bench.err; cut -c1-1500 gpurun_out/full/bench.json
python3 -c "
import json; d=json.load(open('gpurun_out/full/bench.json')); print(d['cpu_baseline']); print(d['check'])"
