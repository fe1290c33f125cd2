cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for e in "DVM_PAIR_CALLS=merged" "DVM_PAIR_CALLS=1" "DVM_PAIR_CALLS=0" "DVM_PAIR_CALLS=1 DVM_CRIT_STREAMS=0" "DVM_CRIT_STREAMS=0"; do
env $e python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --partial --batch 2 --points 4995 --points-target 2200 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],2), round(d['value'],1), round(d['host_enqueue_ms_per_step'],2))"
done
