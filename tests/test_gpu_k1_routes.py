"""-m gpu: pass A of the soft-correspondence kernel is chosen per launch by a probe (csrc/dvm_softcorr_f16.hip::k1_probe_kernel:
coarse one-plane screen / lean first form / full first form); the results must not depend on the choice.  The policy is read once
per process, so each forced route runs the K1 slice of the parity suite in a child process:
  DVM_K1_ROUTE=3   the coarse screen for every launch at alpha >= 32 (csrc/dvm_softcorr_coarse.hip), whatever the probe says — on
                   clustered or flat inputs most rows then fail pass B's certification and take the exact-rows kernel
  DVM_K1_ROUTE=1   the lean first form for every such launch
  DVM_K1_ROUTE=0   the full first form
(reference: models/loss.py:110-114, 1339-1347, 91-95; test.py:19-23)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

K1 = "softcorr or argmin or pair_forward or pair_direction"
SKIP = "not probe_routes"   # (that test reads the unforced policy)


@pytest.mark.parametrize("route", ["3", "1", "0"], ids=["coarse", "lean", "full"])
def test_k1_parity_slice_with_the_route_forced(route):
    e = dict(os.environ, DVM_K1_ROUTE=route)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                        "-k", "(%s) and %s" % (K1, SKIP), "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
