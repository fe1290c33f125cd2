# round 6: bench steps at 64 / 32 resident pairs in the pipelined form as kernel timelines (queues = streams)
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O
for p in 64 32; do
  rm -rf /tmp/prof$p
  rocprofv3 --kernel-trace --stats -d /tmp/prof$p -o b --output-format csv -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check --pairs $p > /tmp/bench$p.log 2>&1
  python3 tools/ktimeline.py /tmp/prof$p sample_absmax 4 5 > $O/timeline_pairs$p.txt
  python3 tools/ktimeline.py /tmp/prof$p sample_absmax 5 5 >> $O/timeline_pairs$p.txt
done
cat $O/timeline_pairs64.txt
