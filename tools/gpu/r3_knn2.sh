# dense kNN path after a change to the score kernel: parity (kNN, backbone, network), time per call with and without the DMA form
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_stress.py -x -q -k "knn or n2p or dist_loss" 2>&1 | tail -2
for f in 0 1; do echo "DVM_KNN_SCORES_DMA=$f"; DVM_KNN_SCORES_DMA=$f timeout 300 python tools/bench_knn.py 2>&1 | grep -v amdgpu.ids | grep "randn"; done
python tools/bench_backbone.py 2>&1 | grep -v amdgpu.ids | tail -1
