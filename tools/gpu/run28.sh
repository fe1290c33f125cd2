cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -q -x 2>&1 | tail -4
python tools/run_softcorr.py 256 5 3 100 2>&1 | head -3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k1 -o k1 --output-format csv -- python3 tools/run_softcorr.py 256 3 3 100 > gpurun_out/k1.log 2>&1
python3 tools/kstats.py gpurun_out/prof_k1 "" 6
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pairs 512 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('pairs/s', round(r['value']), 'ms', round(r['ms_per_step'],2), 'roofline', r['roofline'])"
