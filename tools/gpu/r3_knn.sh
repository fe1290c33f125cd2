# the opt-in fp16-sweep self-kNN: equality with the dense path, timings of both
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_backbone.py -x -q -k "knn" 2>&1 | tail -2
DVM_KNN_F16=1 timeout 300 python tools/bench_knn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/knn_f16.txt
