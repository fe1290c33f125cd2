# round 5: pass B behind the coarse screen (lists of 16, evaluated candidates compacted): window 2 vs 3, bench + parity of the K1 tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
for w in 2 3; do
  DVM_K1_REFINE_WIN=$w timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('win $w: value %.0f step %.2f ms  sweep in step %.3f alone %.3f  passB %.3f ok %s' % (d['value'], d['ms_per_step'], r['launch_ms'], r['standalone']['launch_ms'], r['kernels'][0]['launch_ms'], d['check']['ok']))"
done 2>&1 | tee gpurun_out/r5/refine_win.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "softcorr or pair_forward" 2>&1 | tail -4
DVM_K1_ROUTE=3 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "softcorr and not probe_routes" 2>&1 | tail -4
