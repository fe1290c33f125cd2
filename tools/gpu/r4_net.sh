cd $GRAFT_REPO_ROOT
python tools/bench_train_net.py 8 2048 2
CONC=1 python tools/bench_train_net.py 8 2048 2
CONC=1 python tools/bench_train_net.py 4 2048 4
python tools/bench_train_net.py 4 2048 4
