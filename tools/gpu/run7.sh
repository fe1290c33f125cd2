cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ddp.py tests/test_gpu_abi_state.py tests/test_gpu_backbone.py -q -x -s 2>&1 | tail -25
python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['frac'], r['roofline']['algorithmic']['frac'], r['roofline']['standalone'])"
