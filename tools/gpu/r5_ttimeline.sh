# the training step as a kernel timeline: busy / idle / overlap (tools/ktimeline_train.py)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; rm -rf /tmp/ptl
rocprofv3 --kernel-trace -d /tmp/ptl -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload train --steps 5 --warmup 3 > /tmp/ptl.log 2>&1
tail -c 300 /tmp/ptl.log
python3 $GRAFT_REPO_ROOT/tools/ktimeline_train.py /tmp/ptl -2 | tee $GRAFT_REPO_ROOT/gpurun_out/r5s/timeline_train.txt
