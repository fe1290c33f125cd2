# A/B of two builds of the library on one box: put the baseline at csrc/libdvm_old.so, the candidate at csrc/libdvm_hip.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp dv-matcher_amd/csrc/libdvm_hip.so dv-matcher_amd/csrc/libdvm_new.so
for v in old new old new; do
cp dv-matcher_amd/csrc/libdvm_$v.so dv-matcher_amd/csrc/libdvm_hip.so
echo "== $v"
python tools/run_softcorr.py 256 10 3 100 2>&1 | grep -E "ms/call|equal"
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --pairs 512 2>&1 | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('pairs/s', round(r['value']), 'ms', round(r['ms_per_step'],2), 'in-step launch', round(r['roofline']['launch_ms'],3), 'standalone', round(r['roofline']['standalone']['launch_ms'],3))"
done
