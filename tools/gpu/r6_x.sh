: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r6x; mkdir -p $O
X_K1_W=4 X_K1_FS=1 timeout 900 python -m pytest tests/test_gpu_k1_routes.py tests/test_gpu_parity.py -x -q -k "softcorr or k1 or pair_forward_full or argmin" 2>&1 | tail -3
for w in 8 4; do for fs in 0 1; do
  X_K1_W=$w X_K1_FS=$fs timeout 300 python bench.py --pairs 512 --steps 20 --warmup 5 --cpu-sample 0 --no-check 2>>$O/err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['single_call']
print('W $w fullstores $fs: %8.0f pairs/s %7.3f ms/step | one-call %7.3f ms | sweep in step %.3f ms, alone %.3f ms' % (d['value'], d['ms_per_step'], s['ms_per_step'], r['launch_ms'], r['standalone']['launch_ms']))"
done; done > $O/k1.txt 2>&1
cat $O/k1.txt; tail -3 $O/err.txt
