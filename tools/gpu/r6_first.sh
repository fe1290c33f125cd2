# round 6, first pass: the suite, the bench line (pipelined form + the one-call form inside it), the strong-scaling proxy
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r6a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
for p in 32 64 128 256 512; do
  timeout 300 python bench.py --pairs $p --steps 20 --warmup 5 --cpu-sample 0 --no-check 2>>$O/proxy.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['single_call']
print('pairs %4d: pipelined %8.0f pairs/s %7.3f ms/step (median %7.3f) | one-call %8.0f pairs/s %7.3f ms | same bits %s | sweep %.3f ms | graph_cached %8.0f pairs/s' % ($p, d['value'], d['ms_per_step'], d['median_ms_per_step'], s['value'], s['ms_per_step'], s['bit_identical_to_pipelined'], r['launch_ms'], d['graph_cached']['value']))"
done > $O/scaling_proxy.txt 2>&1
cat $O/scaling_proxy.txt; python3 -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']
print({k: d[k] for k in ('value','ms_per_step','median_ms_per_step','single_call','check','cpu_baseline')})
print({k: r[k] for k in r if k not in ('kernels','traffic_source')})"
