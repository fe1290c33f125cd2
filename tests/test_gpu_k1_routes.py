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


@pytest.mark.parametrize("route,slot", [("3", 2), ("1", 1), ("0", 0), (None, None)], ids=["coarse", "lean", "full", "probe"])
def test_k1_last_routes_reports_what_ran(route, slot):
    """dvm_k1_last_routes (what bench.py prices its roofline line from): with a route forced every (direction, pair) entry is counted
    for that kernel; routed by the probe, random features at alpha 100 take the coarse screen, trained-like features at alpha 33 the
    full first form, and alpha < 32 is the full first form whatever is asked.  (Child process: the policy is read once per process.)"""
    code = (
        "import ctypes, os, sys, torch\n"
        "sys.path.insert(0, os.path.join(%r, 'dv-matcher_amd'))\n"
        "from dvm import ops, _lib\n"
        "lib = _lib.load()\n"
        "def routes():\n"
        "    c = (ctypes.c_int * 5)()\n"
        "    ops.check(lib.dvm_k1_last_routes(c), 'dvm_k1_last_routes')\n"
        "    return list(c)\n"
        "g = torch.Generator().manual_seed(5)\n"
        "f1, f2 = torch.randn(4, 2048, 128, generator=g).cuda(), torch.randn(4, 2048, 128, generator=g).cuda()\n"
        "ops.softcorr(f1, f2, 100.0, topk=10, variant=3); print('R', routes())\n"
        "ops.softcorr(0.3 * torch.relu(f1), 0.3 * torch.relu(f2), 33.0, topk=10, variant=3); print('T', routes())\n"
        "ops.softcorr(f1, f2, 10.0, topk=10, variant=3); print('L', routes())\n" % ROOT)
    env = dict(os.environ)
    env.pop("DVM_K1_ROUTE", None), env.pop("DVM_K1_ROUTE_P", None)
    if route is not None:
        env["DVM_K1_ROUTE"] = route
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    got = {ln[0]: eval(ln[2:]) for ln in res.stdout.splitlines() if ln[:2] in ("R ", "T ", "L ")}
    assert set(got) == {"R", "T", "L"}, res.stdout
    for k in got:
        assert got[k][4] == 4 and sum(got[k][:3]) == 4, got     # one direction x 4 pairs, every entry in exactly one column
    assert got["L"][0] == 4, got                                  # alpha < 32: the full first form
    if route is None:
        assert got["R"][2] == 4 and got["T"][0] == 4, got         # the probe's choice on the two synthetic feature sets
    else:
        assert got["R"][slot] == 4 and got["T"][slot] == 4, got
