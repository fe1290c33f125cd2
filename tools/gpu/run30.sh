cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o b --output-format csv -- python3 bench.py --steps 3 --warmup 2 --cpu-sample 0 --pairs 512 > gpurun_out/bench_prof.log 2>&1
python3 tools/ktimeline.py gpurun_out/prof_bench rownorm2 2 40
