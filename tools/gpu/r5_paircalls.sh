# the two network calls of a training step: merged into one batch (default), side by side on two streams, one after the other
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in merged 1 0; do
  for b in 8 2; do
    DVM_PAIR_CALLS=$m timeout 300 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch $b --points 2048 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('DVM_PAIR_CALLS=$m batch $b: %7.2f ms/step  host enqueue %.2f ms' % (d['ms_per_step'], d['host_enqueue_ms_per_step']))"
  done
done
