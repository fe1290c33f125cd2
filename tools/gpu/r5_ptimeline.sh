# the pair bench step (512 pairs) as a kernel timeline: busy / idle / overlap
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5s; rm -rf /tmp/ppl
rocprofv3 --kernel-trace -d /tmp/ppl -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check > /tmp/ppl.log 2>&1
tail -c 200 /tmp/ppl.log
python3 $GRAFT_REPO_ROOT/tools/ktimeline_train.py /tmp/ppl -2 sample_absmax | tee $GRAFT_REPO_ROOT/gpurun_out/r5s/timeline_pairs512.txt
python3 $GRAFT_REPO_ROOT/tools/ktimeline.py /tmp/ppl sample_absmax 4 100 | tail -50
