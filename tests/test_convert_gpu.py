"""GPU parity of the modulus-conversion kernels vs the oracle (bit for bit), plus the
reference tests' reconstruction identities through the Truncation mirror at full size."""
import threading

import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64, uniform_mod

pytestmark = pytest.mark.gpu
SEED = 0x4D264D
PRIMES = {32: 4293918721, 37: 137438822401, 40: 1099480956929, 41: 2198100901889}


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()


def host(t):
    a = t.cpu().numpy()
    return a.view(np.uint64) if a.dtype == np.int64 else a


def run_pre_post(mm, kind, party, x, y, **kw):
    """one conversion's pre and post kernels on the GPU with a given COT output y"""
    import torch
    from openmm_amd.lib import check, dptr, load_library
    lib = load_library()
    n, skew = len(x), kw.get("skew", 0)   # skew 1: every array starts one element into its allocation (8 mod 16)
    pad = np.zeros(skew, np.uint64)
    d_in, d_y = dev(np.concatenate([pad, x]))[skew:], dev(np.concatenate([pad, y]))[skew:]
    tmp = torch.empty(n + skew, dtype=torch.int64, device="cuda")[skew:]
    msb = torch.empty(n + skew, dtype=torch.uint8, device="cuda")[skew:]
    corr = torch.zeros(n + skew, dtype=torch.int64, device="cuda")[skew:]
    out = torch.empty((n + skew, 2) if kind == "r2r128" else n + skew, dtype=torch.int64, device="cuda")[skew:]
    if kind == "r2f":
        check(lib.mmh_r2f_pre(party, dptr(d_in), n, kw["bw"], kw["p"], dptr(tmp), dptr(msb), dptr(corr), None))
        check(lib.mmh_r2f_post(party, dptr(tmp), dptr(msb), dptr(d_y), n, kw["bw"], kw["p"], dptr(out), None))
    elif kind == "f2r":
        check(lib.mmh_f2r_pre(party, dptr(d_in), n, kw["p"], kw["shift"], dptr(tmp), dptr(msb), dptr(corr), None))
        if kw["shift"]:
            check(lib.mmh_f2r_trunc_post(party, dptr(tmp), dptr(msb), dptr(d_y), n, kw["p"], kw["bw"],
                                         kw["shift"], dptr(out), None))
        else:
            check(lib.mmh_f2r_post(party, dptr(tmp), dptr(msb), dptr(d_y), n, kw["p"], kw["bw"], dptr(out), None))
    else:
        check(lib.mmh_r2r_pre(party, dptr(d_in), n, kw["bw_a"], dptr(tmp), dptr(msb), dptr(corr), None))
        fn = lib.mmh_r2r128_post if kind == "r2r128" else lib.mmh_r2r_post
        check(fn(party, dptr(tmp), dptr(msb), dptr(d_y), n, kw["bw_a"], kw["bw_b"], dptr(out), None))
    torch.cuda.synchronize()
    return host(tmp), host(msb), host(corr), host(out)


def edge_mix(n, hi, seed):
    """random values in [0,hi) with the boundary values forced in"""
    x = uniform_mod(seed, n, hi) if hi < (1 << 64) else splitmix64(seed, n)
    edges = [0, 1, hi - 1, hi // 2, hi // 2 - 1, hi // 2 + 1, hi // 4, hi // 4 - 1, hi // 4 + 1, 3 * (hi // 4)]
    x[:len(edges)] = np.array([e % (1 << 64) for e in edges], dtype=np.uint64)
    return x


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw", [32, 37, 40, 41])
def test_ring_to_field_kernels(mm, party, bw):
    o, p, n = load(), PRIMES[bw], 20000
    x = edge_mix(n, 1 << bw, SEED + bw)
    y = edge_mix(n, p, SEED + 1)
    tmp, msb, corr, out = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_r2f_pre(party, A(x), n, bw, p, A(tmp), A(msb), A(corr))
    o.mmo_r2f_post(party, A(tmp), A(msb), A(y), n, bw, p, A(out))
    g = run_pre_post(mm, "r2f", party, x, y, bw=bw, p=p)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
    if party == 1:
        assert np.array_equal(g[2], corr)


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw,shift", [(32, 0), (37, 0), (32, 12), (41, 12), (64, 0)])
def test_field_to_ring_kernels(mm, party, bw, shift):
    o, n = load(), 20000
    p = PRIMES.get(bw, PRIMES[41])
    x = edge_mix(n, p, SEED + bw)
    y = edge_mix(n, 1 << bw, SEED + 2)
    tmp, msb, corr, out = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_f2r_pre(party, A(x), n, p, shift, A(tmp), A(msb), A(corr))
    if shift:
        o.mmo_f2r_trunc_post(party, A(tmp), A(msb), A(y), n, p, bw, shift, A(out))
    else:
        o.mmo_f2r_post(party, A(tmp), A(msb), A(y), n, p, bw, A(out))
    g = run_pre_post(mm, "f2r", party, x, y, bw=bw, p=p, shift=shift)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
    if party == 1:
        assert np.array_equal(g[2], corr)


@pytest.mark.parametrize("n", [1, 2, 255, 513, 19999])
@pytest.mark.parametrize("skew", [0, 1])
def test_conversion_kernels_odd_lengths_and_alignment(mm, n, skew):
    """the kernels take two elements per lane when their arrays are 16-byte aligned and one otherwise: odd lengths
    (the last lane does one) and arrays at 8 mod 16 give the same bits"""
    o, p, bw = load(), PRIMES[37], 37
    for party in (1, 2):
        x, y = edge_mix(max(n, 10), 1 << bw, SEED + 50)[:n], edge_mix(max(n, 10), p, SEED + 51)[:n]
        tmp, msb, corr, out = (np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64),
                               np.zeros(n, np.uint64))
        o.mmo_r2f_pre(party, A(x), n, bw, p, A(tmp), A(msb), A(corr))
        o.mmo_r2f_post(party, A(tmp), A(msb), A(y), n, bw, p, A(out))
        g = run_pre_post(mm, "r2f", party, x, y, bw=bw, p=p, skew=skew)
        assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
        assert party == 2 or np.array_equal(g[2], corr)
        for shift in (0, 12):
            o.mmo_f2r_pre(party, A(y), n, p, shift, A(tmp), A(msb), A(corr))
            if shift:
                o.mmo_f2r_trunc_post(party, A(tmp), A(msb), A(x), n, p, bw, shift, A(out))
            else:
                o.mmo_f2r_post(party, A(tmp), A(msb), A(x), n, p, bw, A(out))
            g = run_pre_post(mm, "f2r", party, y, x, bw=bw, p=p, shift=shift, skew=skew)
            assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
            assert party == 2 or np.array_equal(g[2], corr)


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw_a,bw_b", [(32, 64), (37, 50), (16, 17), (60, 64), (64, 128), (64, 70), (40, 104), (56, 64)])
def test_ring_extension_kernels(mm, party, bw_a, bw_b):
    o, n = load(), 20000
    x = edge_mix(n, 1 << bw_a, SEED + bw_a)
    y = edge_mix(n, 1 << (bw_b - bw_a), SEED + 3)
    tmp, msb, corr = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64)
    o.mmo_r2r_pre(party, A(x), n, bw_a, A(tmp), A(msb), A(corr))
    out128 = np.zeros(2 * n, np.uint64)
    o.mmo_r2r128_post(party, A(tmp), A(msb), A(y), n, bw_a, bw_b, A(out128))
    g = run_pre_post(mm, "r2r128", party, x, y, bw_a=bw_a, bw_b=bw_b)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb)
    assert np.array_equal(g[3].reshape(-1), out128)
    if party == 1:
        assert np.array_equal(g[2], corr)
    if bw_b <= 64:
        out = np.zeros(n, np.uint64)
        o.mmo_r2r_post(party, A(tmp), A(msb), A(y), n, bw_a, bw_b, A(out))
        assert np.array_equal(run_pre_post(mm, "r2r", party, x, y, bw_a=bw_a, bw_b=bw_b)[3], out)


def test_bole_cross_term(mm):
    from openmm_amd.truncation import bole_cross_term
    o, n, p = load(), 30000, PRIMES[32]
    out, x, y = splitmix64(SEED, n), splitmix64(SEED + 1, n), splitmix64(SEED + 2, n)
    out[:4] = [0, p - 1, (1 << 64) - 1, (1 << 64) - p]
    y[:4] = [0, 1, p, (1 << 64) - 1]
    ref = out.copy()
    o.mmo_bole_cross_term(A(ref), A(x), A(y), n, p)
    d = dev(out)
    bole_cross_term(d, dev(x), dev(y), p)
    assert np.array_equal(host(d), ref)


def _two_party(fn_alice, fn_bob):
    res = {}
    ta = threading.Thread(target=lambda: res.__setitem__(1, fn_alice()))
    tb = threading.Thread(target=lambda: res.__setitem__(2, fn_bob()))
    ta.start(), tb.start()
    ta.join(60), tb.join(60)
    assert 1 in res and 2 in res
    return res[1], res[2]


def test_conversions_reconstruct_at_reference_size(mm):
    """N = 10^7 as in tests/ring_to_field_test.cpp:171: ring->field, field->ring, extension"""
    import torch
    from openmm_amd.truncation import ALICE, BOB, IdealCOT, Truncation
    n, bw, p = 10_000_000, 32, PRIMES[32]
    cot = IdealCOT(SEED)
    alice, bob = Truncation(ALICE, cot), Truncation(BOB, cot)
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    v = torch.randint(-(1 << (bw - 3)), 1 << (bw - 3), (n,), dtype=torch.int64, device="cuda", generator=gen)
    sa = torch.randint(0, 1 << bw, (n,), dtype=torch.int64, device="cuda", generator=gen)
    sb = (v - sa) & ((1 << bw) - 1)
    fa, fb = _two_party(lambda: alice.ring_to_field(n, sa, bw, p), lambda: bob.ring_to_field(n, sb, bw, p))
    assert torch.equal((fa % p + fb % p) % p, v % p)          # tests/ring_to_field_test.cpp:128-143
    fa, fb = fa % p, fb % p
    ra, rb = _two_party(lambda: alice.field_to_ring(n, fa, p, bw), lambda: bob.field_to_ring(n, fb, p, bw))
    assert torch.equal((ra + rb) & ((1 << bw) - 1), v & ((1 << bw) - 1))   # tests/field_to_ring_test.cpp:121-135
    ta, tb = _two_party(lambda: alice.field_to_ring_with_truncate(n, fa, p, bw, 12),
                        lambda: bob.field_to_ring_with_truncate(n, fb, p, bw, 12))
    rec = (ta + tb) & ((1 << bw) - 1)
    rec = torch.where(rec >= (1 << (bw - 1)), rec - (1 << bw), rec)
    assert int((rec - (v >> 12)).abs().max()) <= 1                          # tests/bole_test.cpp:286-289
    ea, eb = _two_party(lambda: alice.ring_to_ring(n, sa, bw, 64), lambda: bob.ring_to_ring(n, sb, bw, 64))
    assert torch.equal(ea + eb, v)                                          # tests/ring_extension_test.cpp:100-119
