"""GPU test of the allocation-free entry point mmh_batch_decode_noalloc (include/mm_hip.h): through the C ABI,
bit for bit against mmh_batch_decode, on two calls in a row, the second with a smaller input (the static
scratch array keeps the first call's contents), plus the size limit.  Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_binding import A, OracleCtx, uniform_mod

pytestmark = pytest.mark.gpu
P32 = 4293918721
N = 4096


def test_batch_decode_noalloc_matches_batch_decode():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    from openmm_amd import lib as L
    lib = L.load_library()
    g = openmm_amd.Context(N, (60, 49), P32)
    o = OracleCtx(N, (60, 49), P32)
    limit = C.c_size_t()
    L.check(lib.mmh_batch_decode_noalloc_max(g.handle, C.byref(limit)))
    assert limit.value == (1 << 20) // N
    s = L._stream_ptr()
    for n in (limit.value, 3):   # full size first, then smaller: stale words of call 1 stay in the array
        pts_h = uniform_mod(0x4D264D + n, (n, N), P32)
        pts = L.to_device(pts_h)
        keep = pts.clone()
        ref = torch.empty_like(pts)
        got = torch.full_like(pts, -1)
        L.check(lib.mmh_batch_decode(g.handle, L.dptr(pts), n, L.dptr(ref), s))
        L.check(lib.mmh_batch_decode_noalloc(g.handle, L.dptr(pts), n, L.dptr(got), s))
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        assert torch.equal(pts, keep)            # the input is not modified
        dec = np.zeros(N, np.uint64)             # and both agree with the oracle
        o.lib.mmo_batch_decode(o.p, A(pts_h[n - 1]), A(dec))
        assert np.array_equal(L.to_host(got[n - 1]), dec)
    too_many = torch.zeros((limit.value + 1, N), dtype=torch.int64, device="cuda")
    out = torch.empty_like(too_many)
    rc = lib.mmh_batch_decode_noalloc(g.handle, L.dptr(too_many), limit.value + 1, L.dptr(out), s)
    assert rc == 6 and b"static scratch" in lib.mmh_last_error()   # ERR_OUT_BOUND, nothing enqueued
