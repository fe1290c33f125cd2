cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in 512 1024 2048; do python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pairs $p 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print($p, 'pairs/s', round(r['value']), 'ms', round(r['ms_per_step'],2), 'frac', round(r['roofline']['frac'],3))"; done
