# K1 pass A, third form (split roles: DVM_K1_SWEEP=5) against the second form: parity tests, bench, alone time, stamps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k1r; rm -f gpurun_out/k1r/ab.txt
DVM_K1_SWEEP=5 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "softcorr or argmin or pair_forward" -x > gpurun_out/k1r/tests.log 2>&1; tail -3 gpurun_out/k1r/tests.log
for f in ${FORMS:-2 5 2 5}; do
DVM_K1_SWEEP=$f timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('form $f: sweep in step %.3f ms frac %.3f | alone %.3f ms frac %.3f | step %.2f ms (median %.2f)  pairs/s %.0f cached %.0f check %s' % (r.get('launch_ms',0), r['frac'], r['standalone']['launch_ms'], r['standalone']['frac'], d['ms_per_step'], d['median_ms_per_step'], d['value'], d.get('graph_cached',{}).get('value',0), d.get('check',{}).get('ok')))" | tee -a gpurun_out/k1r/ab.txt
done
DVM_K1_SWEEP=5 DVM_K1_STAMPS=1 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check 2>&1 | grep "K1 stamps" | tail -2 | tee gpurun_out/k1r/stamps.txt
