# round-2 evidence: bench line, per-kernel stats (bench, training step, backbone eval, config 5), alpha sweep
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_pairs512.json 2> $O/bench.err
python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids > $O/bench_alpha.txt
python tools/bench_backbone.py 8 2048 10 2>&1 | grep -v amdgpu.ids > $O/backbone.txt
python tools/bench_backbone.py 1 4995 10 2>&1 | grep -v amdgpu.ids >> $O/backbone.txt
python tools/bench_visual.py 8 4096 3 2>&1 | grep -v amdgpu.ids > $O/visual.txt
python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 2>&1 | tail -1 > $O/train_B8_N2048.json
python dv-matcher_amd/train_driver.py --partial --steps 6 --warmup 2 --batch 2 --points 4995 --points-target 2200 2>&1 | tail -1 > $O/train_partial_4995x2200.json
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_bench -o x --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > /tmp/p_bench.log 2>&1
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_steps3_pairs512.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_train -o x --output-format csv -- python3 $R/dv-matcher_amd/train_driver.py --steps 2 --warmup 1 --batch 8 --points 2048 > /tmp/p_train.log 2>&1
cp $(find /tmp/p_train -name "*kernel_stats.csv" | head -1) $O/kernel_stats_train_B8_N2048.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_bb -o x --output-format csv -- python3 $R/tools/bench_backbone.py 8 2048 5 > /tmp/p_bb.log 2>&1
cp $(find /tmp/p_bb -name "*kernel_stats.csv" | head -1) $O/kernel_stats_backbone_eval_B8_N2048.csv
rocprofv3 --kernel-trace --stats -d /tmp/p_vis -o x --output-format csv -- python3 $R/tools/bench_visual.py 8 4096 2 > /tmp/p_vis.log 2>&1
cp $(find /tmp/p_vis -name "*kernel_stats.csv" | head -1) $O/kernel_stats_config5_B8_N4096.csv
cat $O/bench_alpha.txt $O/backbone.txt $O/visual.txt; cut -c1-300 $O/train_B8_N2048.json $O/train_partial_4995x2200.json; cut -c1-200 $O/bench_pairs512.json
