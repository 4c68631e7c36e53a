"""CPU tests of the oracle itself: parameter selection, NTT identities, BFV identities,
conversion reconstruction.  The reference ships no golden vectors for this path
(parity unpinned), so the oracle is pinned by the algebraic
acceptance identities the reference's own tests use."""
import numpy as np
import pytest

from tests.oracle_binding import A, U, OracleCtx, load, splitmix64, uniform_mod

SEED = 0x4D264D
Q0, Q1 = 1152921504606830593, 562949953216513
P32 = 4293918721  # SCI/src/utils/constants.h:46


def test_parameter_selection():
    lib = load()
    c = OracleCtx(4096, (60, 49), 1 << 37)
    # values derived in SURVEY.md section 8 from SEAL's documented selection rule
    assert c.q == [Q0, Q1]
    assert c.psi == [116777451583545, 75260022318]
    assert c.gamma == 2305843009213489153
    c8 = OracleCtx(8192, (60, 60, 60, 49), 0)
    assert c8.q == [1152921504606683137, 1152921504606748673, 1152921504606830593, 562949952847873]
    for q, psi in zip(c.q, c.psi):
        assert lib.mmo_powmod(psi, 4096, q) == q - 1


def test_ntt_matches_schoolbook_small():
    lib = load()
    N = 256
    pr = np.zeros(1, dtype=np.uint64)
    assert lib.mmo_get_primes(2 * N, 50, 1, A(pr)) == 0
    q = int(pr[0])
    psi = lib.mmo_min_primitive_root(2 * N, q)
    w, iw = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
    lib.mmo_ntt_tables(N, q, psi, A(w), A(iw))
    a, b = uniform_mod(SEED, N, q), uniform_mod(SEED + 1, N, q)
    c = np.zeros(N, np.uint64)
    lib.mmo_negacyclic_schoolbook(A(a), A(b), A(c), N, q)
    fa, fb = a.copy(), b.copy()
    lib.mmo_ntt_forward(A(fa), N, q, A(w))
    lib.mmo_ntt_forward(A(fb), N, q, A(w))
    # evaluation semantics: out[i] = a(psi^(2*bitrev(i)+1))
    for i in (0, 1, 5, N - 1):
        r = int(f"{i:08b}"[::-1], 2)
        x = pow(psi, 2 * r + 1, q)
        assert int(fa[i]) == sum(int(a[k]) * pow(x, k, q) for k in range(N)) % q
    fc = np.zeros(N, np.uint64)
    lib.mmo_dyadic_mul(A(fa), A(fb), A(fc), N, q)
    lib.mmo_ntt_inverse(A(fc), N, q, A(iw))
    assert np.array_equal(fc, c)


@pytest.mark.parametrize("N,bits", [(4096, (60, 49)), (8192, (60, 60, 60, 49))])
def test_ntt_fast_equals_plain_and_roundtrip(N, bits):
    lib = load()
    c = OracleCtx(N, bits, 0)
    for l, q in enumerate(c.q):
        a = uniform_mod(SEED + l, N, q)
        slow = a.copy()
        lib.mmo_ntt_forward(A(slow), N, q, A(c.table("psi_rev", l)))
        fast = c.ntt_forward(np.stack([a] * c.L)[None])[0, l]
        assert np.array_equal(slow, fast)
        back = slow.copy()
        lib.mmo_ntt_inverse(A(back), N, q, A(c.table("ipsi_rev", l)))
        assert np.array_equal(back, a)
        assert np.array_equal(c.ntt_inverse(np.stack([slow] * c.L)[None])[0, l], a)


def _keys(c, seed):
    """ternary secret (NTT form) from the seeded stream"""
    N, L = c.N, c.L
    s = (splitmix64(seed, N) % np.uint64(3)).astype(np.int64) - 1
    sk = np.stack([np.where(s < 0, np.uint64(q - 1), s.astype(np.uint64)) for q in c.q]).astype(np.uint64)
    return s, c.ntt_forward(sk[None])[0]


def _noise(seed, n):
    r = splitmix64(seed, n)
    return ((r & np.uint64(7)).astype(np.int64) - (r >> np.uint64(61)).astype(np.int64))


@pytest.mark.parametrize("t", [1 << 37, P32])
def test_bfv_encrypt_multiply_modswitch_decrypt(t):
    lib = load()
    c = OracleCtx(4096, (60, 49), t)
    N, L = c.N, c.L
    _, sk = _keys(c, SEED)
    a = np.stack([uniform_mod(SEED + 10 + l, N, q) for l, q in enumerate(c.q)])
    e = _noise(SEED + 20, N)
    m = uniform_mod(SEED + 30, N, t)
    ct = np.zeros((2, L, N), np.uint64)
    lib.mmo_encrypt_symmetric(c.p, 0, A(sk), A(a), A(e), A(m), N, A(ct))
    out = np.zeros(N, np.uint64)
    lib.mmo_decrypt(c.p, 0, A(sk), A(ct), A(out))
    assert np.array_equal(out, m)
    # ct * sparse plaintext == negacyclic product mod t (exact BFV semantics)
    w = np.zeros(N, np.uint64)
    idx = [0, 3, 59, 118, 4095]
    vals = [5, t - 7, 11, t - 1, 2]
    for i, v in zip(idx, vals):
        w[i] = v
    prod = np.zeros((2, L, N), np.uint64)
    lib.mmo_multiply_plain(c.p, 0, A(ct), A(w), N, A(prod))
    lib.mmo_decrypt(c.p, 0, A(sk), A(prod), A(out))
    # expected: signed weights, arithmetic mod t
    exp = np.zeros(N, dtype=object)
    mi = [int(x) for x in m]
    for i, v in zip(idx, vals):
        sv = v - t if v >= (t + 1) // 2 else v
        for k in range(N):
            j = i + k
            if j < N:
                exp[j] += sv * mi[k]
            else:
                exp[j - N] -= sv * mi[k]
    exp = np.array([int(x) % t for x in exp], dtype=np.uint64)
    assert np.array_equal(out, exp)
    # mod-switch to the last level keeps the plaintext
    low = np.zeros((2, 1, N), np.uint64)
    lib.mmo_mod_switch_to_last(c.p, A(prod), A(low), 2)
    lib.mmo_decrypt(c.p, 1, A(sk[:1].copy()), A(low), A(out))
    assert np.array_equal(out, exp)
    # truncate_for_decryption keeps it too (hom_conv2d_ss.cc:413-415)
    m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
    lib.mmo_truncate_masks(c.p, A(m0), A(m1))
    low[0] &= m0[0]
    low[1] &= m1[0]
    lib.mmo_decrypt(c.p, 1, A(sk[:1].copy()), A(low), A(out))
    assert np.array_equal(out, exp)


def test_public_key_encrypt_zero_and_add_sub_plain():
    lib = load()
    t = 1 << 37
    c = OracleCtx(4096, (60, 49), t)
    N, L = c.N, c.L
    _, sk = _keys(c, SEED)
    a_ntt = np.stack([uniform_mod(SEED + 40 + l, N, q) for l, q in enumerate(c.q)])
    pk = np.zeros((2, L, N), np.uint64)
    lib.mmo_make_public_key(c.p, A(sk), A(a_ntt), A(_noise(SEED + 41, N)), A(pk))
    u = (splitmix64(SEED + 42, N) % np.uint64(3)).astype(np.int64) - 1
    z = np.zeros((2, L, N), np.uint64)
    lib.mmo_encrypt_zero_asymmetric(c.p, 0, A(pk), A(u), A(_noise(SEED + 43, N)), A(_noise(SEED + 44, N)), A(z))
    m = uniform_mod(SEED + 45, N, t)
    lib.mmo_add_plain(c.p, 0, A(z), A(m), N)
    out = np.zeros(N, np.uint64)
    lib.mmo_decrypt(c.p, 0, A(sk), A(z), A(out))
    assert np.array_equal(out, m)
    m2 = uniform_mod(SEED + 46, N, t)
    lib.mmo_sub_plain(c.p, 0, A(z), A(m2), N)
    lib.mmo_decrypt(c.p, 0, A(sk), A(z), A(out))
    assert np.array_equal(out, (m - m2) & np.uint64(t - 1))


def test_batch_encoder_roundtrip_and_simd_product():
    lib = load()
    c = OracleCtx(4096, (60, 49), P32)
    N = c.N
    x, y = uniform_mod(SEED, N, P32), uniform_mod(SEED + 1, N, P32)
    px, py, back = np.zeros(N, np.uint64), np.zeros(N, np.uint64), np.zeros(N, np.uint64)
    lib.mmo_batch_encode(c.p, A(x), N, A(px))
    lib.mmo_batch_encode(c.p, A(y), N, A(py))
    lib.mmo_batch_decode(c.p, A(px), A(back))
    assert np.array_equal(back, x)
    prod = np.zeros(N, np.uint64)
    lib.mmo_negacyclic_schoolbook(A(px[:N].copy()), A(py), A(prod), N, P32)
    lib.mmo_batch_decode(c.p, A(prod), A(back))
    exp = np.array([int(a) * int(b) % P32 for a, b in zip(x, y)], dtype=np.uint64)
    assert np.array_equal(back, exp)


def _signed(seed, n, bound):
    return (splitmix64(seed, n) % np.uint64(2 * bound)).astype(np.int64) - bound


@pytest.mark.parametrize("bw", [32, 37, 41])
def test_ring_to_field_reconstructs(bw):
    """tests/ring_to_field_test.cpp:128-143: (outA + outB) mod p is the signed value mod p"""
    lib = load()
    p = {32: 4293918721, 37: 137438822401, 41: 2198100901889}[bw]
    n, mask = 5000, (1 << bw) - 1
    v = _signed(SEED, n, 1 << (bw - 3))
    sa = splitmix64(SEED + 1, n) & np.uint64(mask)
    sb = (v.astype(np.uint64) - sa) & np.uint64(mask)
    outs = {}
    tmp, msb = {}, {}
    corr = np.zeros(n, np.uint64)
    for party, s in ((1, sa), (2, sb)):
        tmp[party], msb[party] = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        lib.mmo_r2f_pre(party, A(s), n, bw, p, A(tmp[party]), A(msb[party]), A(corr))
    x = uniform_mod(SEED + 2, n, p)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(corr), A(msb[2]), n, 0, p, 1, A(y))
    for party, cot in ((1, x), (2, y)):
        outs[party] = np.zeros(n, np.uint64)
        lib.mmo_r2f_post(party, A(tmp[party]), A(msb[party]), A(cot), n, bw, p, A(outs[party]))
    rec = (outs[1] % np.uint64(p) + outs[2] % np.uint64(p)) % np.uint64(p)
    exp = np.array([int(a) % p for a in v], dtype=np.uint64)
    assert np.array_equal(rec, exp)


@pytest.mark.parametrize("bw,shift", [(32, 0), (37, 0), (32, 12), (41, 12)])
def test_field_to_ring_reconstructs(bw, shift):
    """tests/field_to_ring_test.cpp:121-135 (exact); with truncation |err| <= 1 (tests/bole_test.cpp:286-289)"""
    lib = load()
    p = {32: 4293918721, 37: 137438822401, 41: 2198100901889}[bw]
    n, mask = 5000, (1 << bw) - 1
    v = _signed(SEED, n, p // 8)
    sa = uniform_mod(SEED + 1, n, p)
    sb = np.array([(int(a) - int(b)) % p for a, b in zip(v, sa)], dtype=np.uint64)
    tmp, msb = {}, {}
    corr = np.zeros(n, np.uint64)
    for party, s in ((1, sa), (2, sb)):
        tmp[party], msb[party] = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        lib.mmo_f2r_pre(party, A(s), n, p, shift, A(tmp[party]), A(msb[party]), A(corr))
    x = splitmix64(SEED + 2, n) & np.uint64(mask)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(corr), A(msb[2]), n, bw, 0, 0, A(y))
    outs = {}
    for party, cot in ((1, x), (2, y)):
        outs[party] = np.zeros(n, np.uint64)
        if shift:
            lib.mmo_f2r_trunc_post(party, A(tmp[party]), A(msb[party]), A(cot), n, p, bw, shift, A(outs[party]))
        else:
            lib.mmo_f2r_post(party, A(tmp[party]), A(msb[party]), A(cot), n, p, bw, A(outs[party]))
    rec = (outs[1] + outs[2]) & np.uint64(mask)
    rec_s = np.where(rec >= np.uint64(1 << (bw - 1)), rec.astype(np.int64) - (1 << bw), rec.astype(np.int64))
    if shift:
        assert np.max(np.abs(rec_s - (v >> shift))) <= 1
    else:
        assert np.array_equal(rec_s, v)


@pytest.mark.parametrize("bw_a,bw_b", [(32, 64), (37, 50), (64 - 8, 64)])
def test_ring_extension_reconstructs(bw_a, bw_b):
    """tests/ring_extension_test.cpp:100-119"""
    lib = load()
    n, ma, mb = 5000, (1 << bw_a) - 1, (1 << bw_b) - 1 if bw_b < 64 else (1 << 64) - 1
    v = _signed(SEED, n, 1 << (bw_a - 3))
    sa = splitmix64(SEED + 1, n) & np.uint64(ma)
    sb = (v.astype(np.uint64) - sa) & np.uint64(ma)
    tmp, msb = {}, {}
    corr = np.zeros(n, np.uint64)
    for party, s in ((1, sa), (2, sb)):
        tmp[party], msb[party] = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
        lib.mmo_r2r_pre(party, A(s), n, bw_a, A(tmp[party]), A(msb[party]), A(corr))
    sh = bw_b - bw_a
    x = splitmix64(SEED + 2, n) & np.uint64((1 << sh) - 1)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(corr), A(msb[2]), n, sh, 0, 0, A(y))
    outs = {}
    for party, cot in ((1, x), (2, y)):
        outs[party] = np.zeros(n, np.uint64)
        lib.mmo_r2r_post(party, A(tmp[party]), A(msb[party]), A(cot), n, bw_a, bw_b, A(outs[party]))
    rec = (outs[1] + outs[2]) & np.uint64(mb)
    assert np.array_equal(rec, v.astype(np.uint64) & np.uint64(mb))
    # 128-bit variant: 64 -> 64 + s (tests/ring_extension_test.cpp:76)
    o128 = {}
    for party, cot in ((1, x), (2, y)):
        o128[party] = np.zeros(2 * n, np.uint64)
        lib.mmo_r2r128_post(party, A(tmp[party]), A(msb[party]), A(cot), n, bw_a, bw_b, A(o128[party]))
        assert np.array_equal(o128[party][0::2], outs[party])


def test_modulo_mult_is_mulmod():
    lib = load()
    a, b = splitmix64(SEED, 200), splitmix64(SEED + 1, 200)
    for x, y in zip(a, b):
        assert lib.mmo_modulo_mult(int(x), int(y) >> 20, P32) == (int(x) % P32) * (int(y) >> 20) % P32


@pytest.mark.parametrize("N,bits", [(4096, (60, 49)), (8192, (60, 60, 60, 49))])
def test_avx512_restatement_gives_the_same_bits(N, bits):
    """oracle/mm_oracle_avx512.c (the CPU baseline's vector code) against the scalar lazy transforms it restates:
    forward on lazy inputs in [0, 4q), inverse on [0, 2q), every limb, through the batch entry points"""
    from tests.oracle_binding import load as load_oracle
    lib = load_oracle()
    if not lib.mmo_have_avx512():
        pytest.skip("host without AVX-512F/DQ")
    o = OracleCtx(N, bits, 0)
    L = len(o.q)
    qs = np.array(o.q, dtype=np.uint64)[None, :, None]
    raw = splitmix64(0xA5, 3 * L * N).reshape(3, L, N)
    for lazy, fn in ((4, lib.mmo_ntt_forward_batch), (2, lib.mmo_ntt_inverse_batch)):
        a = raw % (qs * np.uint64(lazy))
        a[0, :, :8] = qs[0] * np.uint64(lazy) - np.uint64(1)     # the top of the lazy range
        a[1, :, -8:] = 0
        want, got = a.copy(), a.copy()
        try:
            assert lib.mmo_set_vector_isa(0) == 0
            fn(o.p, A(want), 3, L, 1)
            assert lib.mmo_set_vector_isa(1) == 1
            fn(o.p, A(got), 3, L, 2)
        finally:
            lib.mmo_set_vector_isa(0)
        assert np.array_equal(got, want)
        assert not np.array_equal(got, a)
