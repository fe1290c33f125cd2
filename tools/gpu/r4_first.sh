# round 4, first GPU call: the new tests, the whole -m gpu suite, bench lines (pair / train / partial) as this round's baseline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_abi_state.py tests/test_gpu_ddp.py::test_rccl_branch_runs_at_world_size_one \
  "tests/test_gpu_parity.py::test_softcorr_contract_size_low_alpha_vs_oracle" "tests/test_gpu_parity.py::test_pair_forward_contract_size_low_alpha_vs_oracle" \
  tests/test_gpu_parity.py::test_softcorr_non_finite_features_keep_columns_in_range tests/test_gpu_parity.py::test_partial_criterion_contract_size_vs_oracle \
  tests/test_gpu_image_backbone.py::test_uni3fc_config5_full_size_properties -m gpu -q -x --timeout 900 > $O/new_tests.log 2>&1
echo "new tests rc=$?" >> $O/new_tests.log
python -m pytest tests -m gpu -q --timeout 900 > $O/full.log 2>&1
echo "full rc=$?" >> $O/full.log
python bench.py --steps 20 --warmup 5 > $O/bench_pair.json 2> $O/bench_pair.err
python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python bench.py --workload partial --steps 10 --warmup 3 > $O/bench_partial.json 2> $O/bench_partial.err
tail -5 $O/new_tests.log; tail -5 $O/full.log; cut -c1-300 $O/bench_pair.json; cut -c1-400 $O/bench_train.json
