"""The C-ABI library loads and exports every symbol include/lattigo_ring.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lattigo_ring.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    syms = _declared_symbols()
    assert len(syms) >= 50
    lib = ctypes.CDLL(pkg._native.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_table_matches_header(pkg):
    assert sorted(pkg._native.SYMBOLS) == _declared_symbols()
    pkg._native.lib()  # sets argtypes for every symbol; AttributeError if one is absent


def test_build_info_and_error_string(pkg):
    lib = pkg._native.lib()
    assert b"gfx950" in lib.lr_build_info()
    assert lib.lr_last_error_string() is not None


def test_argument_errors_do_not_need_a_device(pkg):
    lib = pkg._native.lib()
    h = ctypes.c_void_p()
    mod = (ctypes.c_uint64 * 1)(1099512938497)
    # ring/ring_context.go:71-73: invalid degree; :141-146: modulus does not allow NTT
    assert lib.lr_context_create(12, mod, 1, 0, ctypes.byref(h)) == 1
    bad = (ctypes.c_uint64 * 1)(1099512938499)
    assert lib.lr_context_create(1 << 12, bad, 1, 0, ctypes.byref(h)) == 2
    assert b"does not allow NTT" in lib.lr_last_error_string()
    assert lib.lr_context_create(1 << 12, None, 1, 0, ctypes.byref(h)) == 4


def test_product_refuses_to_run_without_its_library(pkg, monkeypatch):
    """no CPU fallback: with the shared library missing, the first call of the binding raises and names the build step; and with the
    library present but no device, creating a context reports the HIP error instead of computing anywhere else"""
    nat = pkg._native
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", nat.LIB_PATH + ".missing")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.ring.NewContextWithParams(16, [pkg.params.Qi60()[-1]])
    monkeypatch.undo()
    if nat.device_count() == 0:
        with pytest.raises(nat.LatticeRingError) as e:
            pkg.ring.NewContextWithParams(16, [pkg.params.Qi60()[-1]])
        assert e.value.code == 5                                                 # LR_ERR_HIP: no device to create the context on
