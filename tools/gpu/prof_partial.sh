: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pp -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --partial --steps 4 --warmup 1 --batch 2 --points 4995 --points-target 2200 > /tmp/pp.log 2>&1
python3 $R/tools/kstats.py /tmp/pp "" 22
