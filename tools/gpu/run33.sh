cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT/dv-matcher_amd
cat > /tmp/syncdbg.py <<'PY'
import sys, runpy, torch, warnings
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
sys.argv = ["train_driver.py", "--partial", "--steps", "2", "--warmup", "1", "--batch", "2", "--points", "4995", "--points-target", "2200"]
runpy.run_path("train_driver.py", run_name="__main__")
PY
timeout 300 python /tmp/syncdbg.py 2>&1 | grep -A1 "synchroniz" | grep -v "^--\|synchroniz" | sort | uniq -c | sort -rn | head -20
