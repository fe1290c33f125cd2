import os, sys, ctypes, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dv-matcher_amd"))
shutil.copy(os.path.join(ROOT, "dv-matcher_amd/csrc/libdvm_dbg.so"), os.path.join(ROOT, "dv-matcher_amd/csrc/libdvm_hip.so"))
import torch
from dvm import ops, _lib
lib = _lib.load()
lib.dvm_debug_counters.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
g = torch.Generator().manual_seed(0)
B = 16
f1 = torch.randn(B, 2048, 128, generator=g).cuda(); f2 = torch.randn(B, 2048, 128, generator=g).cuda()
out = (ctypes.c_ulonglong * 4)()
lib.dvm_debug_counters(out)
for alpha in (100.0, 40.0, 10.0):
    ops.softcorr(f1, f2, alpha); lib.dvm_debug_counters(out)
    waves = B * 8 * 8
    print("alpha", alpha, "loop iterations per wave-subtile-epilogue:", out[0] / max(out[1], 1), " flagged per lane per epilogue:", out[2] / max(out[1], 1) / 64,
          " epilogues per wave:", out[1] / waves)
