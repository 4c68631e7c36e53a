"""CPU check: libmm_hip.so loads and exports every symbol include/mm_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from openmm_amd import lib as L
    names = declared_symbols()
    assert len(names) >= 10
    dll = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(dll, n)]
    assert not missing, missing
    # the Python binding covers the whole header too
    unbound = [n for n in names if n not in L.SIGNATURES and n != "mmh_last_error"]
    assert not unbound, unbound


def test_bad_arguments_return_codes_without_gpu():
    from openmm_amd import lib as L
    lib = L.load_library()
    assert lib.mmh_ctx_create(None, 4096, None, 2, 0, 0) == 2  # ERR_NULL_POINTER
    h = ctypes.c_void_p()
    q = (ctypes.c_uint64 * 2)(1152921504606830593, 562949953216513)
    assert lib.mmh_ctx_create(ctypes.byref(h), 1000, q, 2, 0, 0) == 1  # ERR_CONFIG: N not built
    bad = (ctypes.c_uint64 * 2)(1152921504606830595, 562949953216513)
    assert lib.mmh_ctx_create(ctypes.byref(h), 4096, bad, 2, 0, 0) == 7  # ERR_INVALID_ARG
    assert b"mmh_ctx_create" in lib.mmh_last_error()
