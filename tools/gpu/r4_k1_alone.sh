cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/k1w
for cfg in "DVM_K1_WAVES=4" "DVM_K1_WAVES=4 DVM_K1_LDS_PAD=8192"; do
echo "== $cfg" | tee -a gpurun_out/k1w/alone.txt
env $cfg DVM_K1_STAMPS=1 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-check 2>&1 | grep "K1 stamps" | tail -2 | tee -a gpurun_out/k1w/alone.txt
env $cfg python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('step %.2f ms  sweep in step %.3f ms alone %s' % (d['ms_per_step'], r.get('launch_ms', 0), r.get('alone')))" | tee -a gpurun_out/k1w/alone.txt
done
