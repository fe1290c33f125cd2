cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python tools/gpu/r5_fills.py 3 2>&1 | tail -70
