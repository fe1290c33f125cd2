# what each criterion term costs the training STEP (wall time): the step with one term's weight set to 0 (timing only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, copy, yaml
sys.path.insert(0, "dv-matcher_amd")
import train_driver as t
for name, kv in (("full", {}), ("no_dist", {"w_dist": 0}), ("no_map", {"w_map": 0}), ("no_self_rec", {"w_self_rec": 0})):
    c = copy.deepcopy(t.FULL_CFG)
    c["loss"].update(kv)
    yaml.safe_dump(c, open("/tmp/cfg_%s.yaml" % name, "w"))
PY
for n in full no_dist no_map no_self_rec full; do
python dv-matcher_amd/train_driver.py --config /tmp/cfg_$n.yaml --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))"
done
