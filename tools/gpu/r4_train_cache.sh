cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in "" "--graph-cache"; do
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 $a 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step'], d.get('graph_cache'))"
done
