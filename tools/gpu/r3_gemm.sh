# conv GEMM with LDS-DMA staging: bit-exactness under every forced configuration, then the configuration sweep
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for c in 4 5 6 7; do echo "DVM_LINEAR_CFG=$c: $(DVM_LINEAR_CFG=$c python -m pytest tests/test_gpu_linear.py -x -q 2>&1 | tail -1)"; done
python tools/bench_linear_cfg.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/linear_cfg.txt
