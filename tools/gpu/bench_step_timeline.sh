# parity suites, bench line and the kernel timeline of one bench step (tools/ktimeline.py)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_abi_state.py -q -x 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pairs 512 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('pairs/s', round(r['value']), 'ms', round(r['ms_per_step'],2), 'in-step launch', round(r['roofline']['launch_ms'],3), 'standalone', round(r['roofline']['standalone']['launch_ms'],3))"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench2 -o b --output-format csv -- python3 bench.py --steps 3 --warmup 2 --cpu-sample 0 --pairs 512 > gpurun_out/bench_prof.log 2>&1
python3 tools/ktimeline.py gpurun_out/prof_bench2 rownorm2 2 100
