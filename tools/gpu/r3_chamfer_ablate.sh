# grid Chamfer, deferred form: what the retry kernel's time is made of (WRONG results under ablation; timing only)
cd /tmp && export TMPDIR=/tmp
for a in ${ABL:-0 1 2 3 4}; do
rm -rf /tmp/p_b; DVM_CHAMFER_DEFER=1 DVM_CHAMFER_ABLATE=$a rocprofv3 --kernel-trace --stats -d /tmp/p_b -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-check > /tmp/p_b.log 2>&1
echo "ABLATE=$a"; grep -i "chamfer" $(find /tmp/p_b -name "*kernel_stats.csv" | head -1) | cut -d, -f1,4 
done
