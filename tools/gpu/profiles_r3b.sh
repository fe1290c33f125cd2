# round-3 evidence, second half (gpurun_out/r3q/ -> profiles/r3_* by hand): the other workloads through bench.py (--workload train |
# partial), backbone eval forward (native call), config 5, and per-kernel stats of the training step and the backbone forward
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; rm -rf $O; mkdir -p $O
cd $R
python bench.py --workload train > $O/bench_train.json 2> $O/bench_train.err
python bench.py --workload partial > $O/bench_partial.json 2> $O/bench_partial.err
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids > $O/backbone.txt
python tools/bench_backbone.py 1 4995 10 2>&1 | grep -v amdgpu.ids >> $O/backbone.txt
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu.ids > $O/visual.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_train -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 2 --warmup 1 --batch 8 --points 2048 > /tmp/p_train.log 2>&1
cp $(find /tmp/p_train -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_B8_N2048.csv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_bb -o x --output-format csv -- python3 $R/tools/bench_backbone.py 8 2048 5 > /tmp/p_bb.log 2>&1
cp $(find /tmp/p_bb -name "*kernel_stats.csv" | head -1) $O/kernel_stats_backbone_eval_B8_N2048.csv
cat $O/backbone.txt $O/visual.txt; cut -c1-600 $O/bench_train.json $O/bench_partial.json; head -14 $O/kernel_stats_backbone_eval_B8_N2048.csv | cut -c1-200
