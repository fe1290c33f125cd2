"""test_driver.py at a given size: the T_<a>_<b>.txt maps it writes equal the CPU oracle's exact arg-min on the features
it saved (the N = 300 version of this is tests/test_gpu_backbone.py::test_inference_driver_writes_reference_outputs)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd")); sys.path.insert(0, ROOT)
import numpy as np
import scipy.io
import test_driver
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4995
out = tempfile.mkdtemp()
test_driver.main(["--synthetic", "1", "--points", str(N), "--out", out])
f1 = scipy.io.loadmat(os.path.join(out, "feature", "usefeature_s000a.mat"))["uphi"].astype(np.float32)
f2 = scipy.io.loadmat(os.path.join(out, "feature", "usefeature_s000b.mat"))["uphi"].astype(np.float32)
T12 = np.loadtxt(os.path.join(out, "T", "T_s000a_s000b.txt"), dtype=np.int64)
T21 = np.loadtxt(os.path.join(out, "T", "T_s000b_s000a.txt"), dtype=np.int64)
o12, _ = O.argmin_exact(f1, f2)
o21, _ = O.argmin_exact(f2, f1)
print("N=%d maps equal the oracle: %s %s" % (N, np.array_equal(T12, o12 + 1), np.array_equal(T21, o21 + 1)))
