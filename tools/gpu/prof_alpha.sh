# the pair forward across the alpha schedule on both synthetic feature sets, second vs first form of the sweep
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(echo "== DVM_K1_SWEEP=2 (default: second form)"; python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids
 echo "== DVM_K1_SWEEP=0 (first form)"; DVM_K1_SWEEP=0 python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r3/bench_alpha.txt
