cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
export DVM_TRAIN_LAYOUT=pm
for ts in 0 1; do
echo "graph two_streams $ts"
DVM_TWO_STREAMS=$ts timeout 300 python train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 --graph 2>&1 | tail -1 | cut -c1-200
done
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_pm2 -o pm --output-format csv -- python3 train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 > $GRAFT_REPO_ROOT/gpurun_out/pm.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/ktrace.py $GRAFT_REPO_ROOT/gpurun_out/prof_pm2 "bn_pm" 45
python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/prof_pm2 "" 12
