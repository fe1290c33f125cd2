cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_backbone.py -m gpu -q --timeout 900 -k "deformer or mlp or pair or criterion" 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), j['check']['ok'], 'cached', round(j['graph_cached']['value']))"; done
