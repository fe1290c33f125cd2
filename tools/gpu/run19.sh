cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
export DVM_TRAIN_LAYOUT=pm
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pm -o pm --output-format csv -- python3 train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 > $GRAFT_REPO_ROOT/gpurun_out/pm.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/prof_pm "" 45
