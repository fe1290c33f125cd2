# the probe's thresholds: the pair forward with every (direction, pair) forced through one pass-A kernel, against the probe's
# own choice, across alpha on both synthetic feature sets; the probe's fractions printed once per point
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export ALPHAS=${ALPHAS:-33,50,70,100,150}
(for r in 0 1 2; do echo "== DVM_K1_ROUTE=$r"; DVM_K1_ROUTE=$r python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids; done
 echo "== routed by the probe"; python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids
 echo "== the probe's fractions"; ITERS=1 DVM_K1_ROUTE_DEBUG=1 python tools/bench_alpha.py 2>&1 | grep -v amdgpu.ids | uniq) | tee gpurun_out/r3/route_calib.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "softcorr or pair_forward" 2>&1 | tail -3
