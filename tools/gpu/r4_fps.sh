cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_dataset.py -m gpu -q --timeout 900 -k "fps or graph or pair_direction or dataset" 2>&1 | grep -v amdgpu | tail -3
python tools/bench_fps.py 2>&1 | grep -v amdgpu | tail -5
python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), j['check']['ok'], [ (k['kernel'][:8], round(k['launch_ms'],2)) for k in j['roofline']['kernels']])"
