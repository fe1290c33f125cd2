cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_abi_state.py tests/test_gpu_backbone.py tests/test_dataset.py -m gpu -q --timeout 900 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): print(ln.strip()); continue
    j=json.loads(ln); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],3), 'check', j['check'], 'cached', round(j['graph_cached']['value']), [ (k['kernel'][:8], round(k['launch_ms'],2)) for k in r['kernels']], r['traffic_source'])"
python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | cut -c1-200
python bench.py --workload partial --steps 10 --warmup 3 2>/dev/null | cut -c1-200
