"""Which ATen calls does a training step still make, and from where?  (<= 600 launches per step, host <= 3 ms.)
Runs train_driver's timing mode under a TorchDispatchMode that counts every operator reaching the dispatcher from Python (the
autograd engine's own thread is not seen) by the innermost frame of this repository.
python tools/train_smallops.py [driver args...]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))


def main():
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    import train_driver
    by = collections.Counter()

    class Count(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            try:
                f = sys._getframe(1)
            except ValueError:      # called from the autograd engine's thread
                f = None
            where = "(autograd engine)"
            while f is not None:
                fn = f.f_code.co_filename
                if "dv-matcher_amd" in fn and "tools/" not in fn:
                    where = "%s:%d %s" % (fn.split("dv-matcher_amd/")[-1], f.f_lineno, f.f_code.co_name)
                    break
                f = f.f_back
            by[(str(func), where)] += 1
            return func(*args, **(kwargs or {}))

    steps, warm = 3, 1
    argv = ["--steps", str(steps), "--warmup", str(warm), "--batch", "8", "--points", "2048"] + sys.argv[1:]
    with Count():
        train_driver.main(argv)
    n = steps + warm
    tot = collections.Counter()
    for (name, where), c in by.items():
        tot[name] += c
    print("operators reaching the dispatcher from Python, per step (%d steps; set-up calls inflate the counts slightly)" % n)
    for name, c in tot.most_common(30):
        print("%8.1f  %s" % (c / n, name))
    print("\nby call site (>= 1 per step):")
    for (name, where), c in by.most_common(200):
        if c / n >= 1:
            print("%8.1f  %-34s %s" % (c / n, name, where))


if __name__ == "__main__":
    main()
