# K1 second form: 8-wave workgroups (one per compute unit) vs 4-wave workgroups (two per compute unit)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k1w; rm -f gpurun_out/k1w/ab.txt
DVM_K1_WAVES=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "softcorr or argmin or pair_forward" -x > gpurun_out/k1w/tests.log 2>&1; tail -3 gpurun_out/k1w/tests.log
run() { timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 sweep %.3f ms (alone %s) frac %.3f  step %.2f ms (median %.2f)  pairs/s %.0f cached %.0f check %s' % (r['kernels'][0]['launch_ms'], r.get('alone',{}).get('launch_ms'), r['frac'], d['ms_per_step'], d['median_ms_per_step'], d['value'], d.get('graph_cached',{}).get('value',0), d.get('check',{}).get('ok')))" | tee -a gpurun_out/k1w/ab.txt; }
for w in 8 4 8 4; do DVM_K1_WAVES=$w run waves$w; done
for w in 8 4; do DVM_K1_WAVES=$w DVM_K1_STAMPS=1 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check 2>&1 | grep "K1 stamps" | tail -2 | tee -a gpurun_out/k1w/stamps.txt; done
