# where the host spends a training step (cProfile, sorted by own time)
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
python -m cProfile -o /tmp/prof.out train_driver.py --steps 10 --warmup 2 --batch 2 --points 1024 > /tmp/log.txt 2>&1
tail -1 /tmp/log.txt | cut -c60-250
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('tottime').print_stats(45)
PY
