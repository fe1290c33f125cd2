# round 5: one bench step at 64 resident pairs (one rank's share of an 8-GPU strong-scaling run) as a kernel timeline
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
rm -rf /tmp/prof64
rocprofv3 --kernel-trace --stats -d /tmp/prof64 -o b --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check --pairs 64 > /tmp/bench64.log 2>&1
python3 tools/ktimeline.py /tmp/prof64 sample_absmax 4 8 | tee gpurun_out/r5/timeline_pairs64.txt
