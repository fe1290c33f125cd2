# Deformer MLP: persistent form vs the kernel it replaces (same box), tests, stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mlp; rm -f gpurun_out/mlp/ab.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "deformer or pair_forward or mlp or Deformer" -x > gpurun_out/mlp/tests.log 2>&1; tail -3 gpurun_out/mlp/tests.log
run() { timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['roofline']['kernels'] if 'mlp' in x['kernel']][0]
print('$1 mlp %.3f ms  step %.2f ms (median %.2f)  pairs/s %.0f cached %.0f check %s' % (k['launch_ms'], d['ms_per_step'], d['median_ms_per_step'], d['value'], d.get('graph_cached',{}).get('value',0), d.get('check',{}).get('ok')))" | tee -a gpurun_out/mlp/ab.txt; }
DVM_MLP_PERSIST=0 run old
for b in 1 2 4 8; do DVM_MLP_BPW=$b run bpw$b; done
DVM_MLP_PERSIST=0 run old
DVM_MLP_BPW=4 run bpw4
DVM_MLP_BPW=4 DVM_MLP_STAMPS=1 python bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-check 2>&1 | grep "MLP stamps" | tail -1 | tee gpurun_out/mlp/stamps_persist.txt
