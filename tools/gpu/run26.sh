cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_pm.py -q -x -k fusion 2>&1 | grep -E "assert|Error|worst" | head
cd dv-matcher_amd
cat > /tmp/syncdbg.py <<'PY'
import sys, runpy, torch, warnings
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
sys.argv = ["train_driver.py", "--steps", "2", "--warmup", "1", "--batch", "8", "--points", "2048"]
runpy.run_path("train_driver.py", run_name="__main__")
PY
timeout 300 python /tmp/syncdbg.py 2>&1 | grep -B1 -A3 "synchroniz" | grep -v "^--" | sort | uniq -c | sort -rn | head -30
