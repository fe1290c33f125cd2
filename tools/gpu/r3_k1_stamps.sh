# round 3: where a wave of the second sweep form spends its cycles (s_memtime stamps per phase, diagnostic build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for f in 1 2; do
  DVM_K1_SWEEP=$f DVM_K1_STAMPS=1 python tools/run_softcorr.py 256 2 3 100 2>&1 | grep -E "K1 stamps|ms/call|flagged|equal" | tail -6
done > gpurun_out/r3/k1_stamps.txt 2>&1
cat gpurun_out/r3/k1_stamps.txt
DVM_K1_SWEEP=2 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "clustered" 2>&1 | tail -40 > gpurun_out/r3/k1_clustered.txt
cat gpurun_out/r3/k1_clustered.txt
