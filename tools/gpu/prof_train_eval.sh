# kernel stats of the training step and of the eval forward
: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_train -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 4 --warmup 1 --batch 8 --points 2048 > /tmp/p_train.log 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/p_train "" 24
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_bb -o x --output-format csv -- python3 $R/tools/bench_backbone.py 8 2048 5 > /tmp/p_bb.log 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/p_bb "" 16
