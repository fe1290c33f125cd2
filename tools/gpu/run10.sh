cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -q -x 2>&1 | tail -5
python bench.py --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('pairs/s', r['value'], 'frac', r['roofline']['frac'], 'launch_ms', r['roofline']['launch_ms'], 'standalone', r['roofline']['standalone'])"
python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids | tail -12
