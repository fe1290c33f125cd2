cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi_state.py tests/test_gpu_stress.py -m gpu -q --timeout 900 -k "pair or graph or smoke or two_streams or warp" 2>&1 | grep -v amdgpu | tail -3
for m in 0 1 0 1; do echo "DVM_WARP_FUSED=$m"; DVM_WARP_FUSED=$m python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],3), j['check'], 'cached', round(j['graph_cached']['value']))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_m -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_m.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/p_m "dg_" 4 | cut -c1-140
