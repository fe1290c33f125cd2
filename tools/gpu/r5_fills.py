"""Where do the fill / copy launches of a training step come from?  Runs train_driver's timing mode under torch.profiler (with Python stacks)
and prints the callers of aten::zeros / zero_ / fill_ / copy_ / clone / cat.  usage: python tools/gpu/r5_fills.py [steps]"""
import io
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import train_driver  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    with contextlib.redirect_stdout(io.StringIO()):
        train_driver.main(["--steps", str(steps), "--warmup", "2", "--batch", "8", "--points", "2048"])
want = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone", "aten::cat", "aten::zeros_like", "aten::full", "aten::sum",
        "aten::mul", "aten::add", "aten::add_", "aten::bmm", "aten::select", "aten::index_select", "aten::index_add_")
import collections  # noqa: E402
cnt = collections.Counter()
for e in prof.events():
    if e.name not in want:
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 3:
        chain.append(q.name[:60])
        q = q.cpu_parent
    if chain and chain[0] in want:      # (an op called by another listed op: counted with its caller)
        continue
    cnt[(e.name, " <- ".join(chain))] += 1
for (name, chain), c in cnt.most_common(50):
    print("%5d  %-18s %s" % (c, name, chain))
