cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pk -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py 8 2048 5 > /tmp/pk.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pk knn; python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pk topk; grep Uni3FC /tmp/pk.log
rocprofv3 --kernel-trace --stats -d /tmp/pk2 -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_backbone.py 1 4995 5 > /tmp/pk2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/pk2 knn; grep Uni3FC /tmp/pk2.log
