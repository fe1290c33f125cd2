"""Sanitizer builds on the CPU (SURVEY §5 / VERDICT r1 #9): the C oracle and the HOST side of libdvm_hip (argument
validation, workspace arenas, launch geometry, context registries) compiled with AddressSanitizer + UBSan and driven by
small harnesses.  GPU sanitizers are not available on this pool; nothing here launches a kernel."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def _run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, **kw)


def test_oracle_under_asan_ubsan():
    b = _run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    assert b.returncode == 0, b.stderr[-2000:]
    r = _run([os.path.join(ROOT, "oracle", "san", "build", "san_oracle")], env=ENV)
    assert r.returncode == 0 and "san_oracle: ok" in r.stdout and "runtime error" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, \
        (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_side_of_the_hip_library_under_asan_ubsan():
    """Builds every source of dv-matcher_amd/csrc with -fsanitize=address,undefined on the host code (device code compiled
    as usual) and runs csrc/san/san_host: workspace queries at the shipped shapes, every rejection path of the entry
    points, the context registry.  (First found: pointer arithmetic on the null base used to SIZE a workspace.)"""
    b = _run(["make", "-C", os.path.join(ROOT, "dv-matcher_amd", "csrc"), "-s", "-j8", "san"])
    assert b.returncode == 0, b.stderr[-2000:]
    r = _run([os.path.join(ROOT, "dv-matcher_amd", "csrc", "build_san", "san_host")], env=dict(ENV, ASAN_OPTIONS=ENV["ASAN_OPTIONS"] + ":detect_leaks=0"))
    assert r.returncode == 0 and "san_host: ok" in r.stdout and "runtime error" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, \
        (r.stdout[-500:], r.stderr[-3000:])
