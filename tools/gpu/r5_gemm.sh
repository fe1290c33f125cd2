: ${GRAFT_REPO_ROOT:?}   # (the recipes rm -rf / write under it)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s
timeout 600 python tools/bench_gemm_bwd.py 2>&1 | tee gpurun_out/r5s/gemm_bwd.txt
