# round 4: the native training node — its own tests, the training-step parity tests that must stay green, the training benches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4t; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_train_native.py -m gpu -q -x --timeout 900 > $O/native.log 2>&1; echo "rc=$?" >> $O/native.log
tail -25 $O/native.log
python -m pytest tests/test_gpu_network.py tests/test_gpu_ddp.py tests/test_gpu_train_pm.py tests/test_gpu_backward.py -m gpu -q --timeout 900 > $O/train_tests.log 2>&1; echo "rc=$?" >> $O/train_tests.log
tail -8 $O/train_tests.log
python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; cut -c1-500 $O/bench_train.json; tail -3 $O/bench_train.err
DVM_NATIVE_TRAIN=0 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_py.json 2>/dev/null; cut -c1-300 $O/bench_train_py.json
python bench.py --workload partial --steps 10 --warmup 3 > $O/bench_partial.json 2> $O/bench_partial.err; cut -c1-300 $O/bench_partial.json
DVM_STEP_BREAKDOWN=1 python dv-matcher_amd/train_driver.py --steps 10 --warmup 3 --batch 8 --points 2048 > $O/breakdown.log 2>&1; grep "host ms" $O/breakdown.log
