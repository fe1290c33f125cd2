# shader clock / power while the bench runs (rocm-smi samples), and idle
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/clk
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -20 > gpurun_out/clk/idle.txt
python bench.py --steps 400 --warmup 5 --cpu-sample 0 --no-check > gpurun_out/clk/bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -4; sleep 1; done > gpurun_out/clk/load.txt
wait $BP
cat gpurun_out/clk/idle.txt; echo ==; cat gpurun_out/clk/load.txt; cut -c1-200 gpurun_out/clk/bench.json
