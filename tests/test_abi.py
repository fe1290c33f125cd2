"""CPU-side checks of the C ABI: the library loads without a GPU and exports every symbol
include/dvm.h declares; the ctypes table covers the header one to one."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dvm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = header_symbols()
    assert "dvm_softcorr_fwd_f32" in syms and "dvm_pair_direction_fwd_f32" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    from dvm import _lib
    assert os.path.exists(_lib.SO_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.SO_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), "libdvm_hip.so does not export %s" % s


def test_ctypes_table_matches_header():
    from dvm import _lib
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_loads_and_reports_without_gpu():
    from dvm import _lib
    lib = _lib.load()
    assert lib.dvm_abi_version() == 1
    assert lib.dvm_softcorr_workspace_bytes(2, 100, 50, 128) >= (2 * 100 + 2 * 50) * 4
    assert lib.dvm_pair_direction_workspace_bytes(1, 256, 256) > 0


def test_no_cpu_fallback():
    import torch
    from dvm import ops
    from dvm._lib import DvmError
    f = torch.randn(1, 8, 128)
    with pytest.raises(DvmError):
        ops.softcorr(f, f, 10.0)
    with pytest.raises(DvmError):
        ops.knn_cdist(torch.rand(1, 8, 3), torch.rand(1, 8, 3), 3)


def test_argument_validation_without_gpu():
    """Shape/argument errors are reported before anything touches a device."""
    from dvm import _lib
    lib = _lib.load()
    rc = lib.dvm_softcorr_fwd_f32(None, None, 1, 8, 8, 128, -1.0, 10, None, None, None, None, 0, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.dvm_last_error()
    one = ctypes.c_void_p(16)
    rc = lib.dvm_softcorr_fwd_f32(one, one, 1, 8, 8, 130, -1.0, 10, one, one, None, None, 0, None, 0, None)
    assert rc == -1 and b"d=130" in lib.dvm_last_error()
    rc = lib.dvm_softcorr_fwd_f32(one, one, 1, 8, 8, 128, -1.0, 10, one, one, None, None, 0, None, 0, None)
    assert rc == -3 and b"workspace" in lib.dvm_last_error()
    rc = lib.dvm_knn_cdist_f32(one, one, 1, 8, 8, 3, 17, one, None, 0, None)
    assert rc == -1
