cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_parity.py::test_pair_forward_graph_cache_is_bit_identical tests/test_gpu_backbone.py::test_criterion_graph_cache_equals_rebuild tests/test_gpu_train_native.py tests/test_gpu_ddp.py::test_bench_single_gpu_line_and_check -m gpu -q --timeout 900 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 > $O/bench_pair.json 2> $O/bench_pair.err; python -c "
import json; j=json.load(open('$O/bench_pair.json')); print(j['value'], j['ms_per_step'], j['median_ms_per_step'], j['graph_cached'], j['check'])"
python tools/bench_train_net.py 8 2048 2
python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | cut -c1-260
