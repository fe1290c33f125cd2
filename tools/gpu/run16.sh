cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_network.py -q -x -k full_size --durations=3 2>&1 | tail -12
