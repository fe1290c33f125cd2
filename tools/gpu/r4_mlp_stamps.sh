# Deformer MLP: where a wave's cycles go (s_memtime stamps per phase, diagnostic build of the kernels) and the shader clock they ran at
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mlp; rm -f gpurun_out/mlp/stamps.txt
for cfg in "DVM_MLP_PERSIST=0" "DVM_MLP_BPW=1" "DVM_MLP_BPW=4" "DVM_MLP_BPW=0"; do
echo "== $cfg" | tee -a gpurun_out/mlp/stamps.txt
env $cfg DVM_MLP_STAMPS=1 python bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-check 2>&1 | grep "MLP stamps" | tail -2 | tee -a gpurun_out/mlp/stamps.txt
done
