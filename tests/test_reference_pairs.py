"""The conversions at the (bit width, prime) pairs of the reference's own test programs
(tests/ring_to_field_test.cpp:24-32, tests/field_to_ring_test.cpp:24-32: bw 16/32/40/48/56/60/64 with
65521 ... 2^64 - 59; tests/ring_extension_test.cpp:145-149: 64 -> 64 + s, s in [1, 64]), inputs drawn the way those
programs draw them, edge values forced in.

CPU part: (i) the C oracle against a second, line-by-line restatement of SCI/src/BuildingBlocks/truncation.cpp in
Python integers (uint64 wrap made explicit), bit for bit, both parties; (ii) the two-party flow over the ideal COT
reconstructs the reference tests' signed value exactly.  These pairs reach what the table primes of constants.h do
not: 2^l > p (16-bit), `p - y` and `x + y` next to 2^64 (2^64 - 59), `1 << (bw - 2)` at bw = 64.
GPU part (marked gpu): the pre/post kernels through the C ABI against the oracle on the same inputs.
"""
import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64, uniform_mod

SEED = 0x4D264D
M64 = (1 << 64) - 1
# tests/ring_to_field_test.cpp:24-32 == tests/field_to_ring_test.cpp:24-32
REF_PAIRS = [(16, 65521), (32, 4293918721), (40, 1099511480321), (48, 281474976694273),
             (56, 72057594037641217), (60, 1152921504606830593), (64, 18446744073709551557)]
EXT_SHIFTS = [1, 8, 32, 63, 64]   # tests/ring_extension_test.cpp:139-149: base_bw = 64, shift_bw in [1, 64]


def bw_mask(bw):
    return M64 if bw == 64 else (1 << bw) - 1


# ------------------------------------------------------------------ second restatement (Python integers)
def secure_add(x, y, p):            # truncation.cpp:533-536
    return (x - (p - y)) & M64 if x >= (p - y) & M64 else (x + y) & M64


def secure_sub(x, y, p):            # truncation.cpp:538-541
    return (x - y) & M64 if x >= y else (x + ((p - y) & M64)) & M64


def py_r2f(party, x, y, bw, p):
    """truncation.cpp:518-581 with the COT output y given: (tmp, msb, corr, out)"""
    mask, msb_mask, big = bw_mask(bw), 1 << (bw - 1), 1 << (bw - 2)
    tmp = (x + big) & mask if party == 1 else x                    # :525-531
    msb = int(bool(tmp & msb_mask))                                # :561-564
    cv = mask % p + 1                                              # :566
    corr = (0 if msb else cv) if party == 1 else 0                 # :546-548
    if party == 1:                                                 # :551-554
        t = secure_sub(0, y, p)
        y = secure_add(cv, t, p) if msb else t
    out = secure_sub(tmp, y, p)                                    # :571 / :576
    if party == 1:
        out = secure_sub(out, big % p, p)                          # :572
    return tmp, msb, corr, out


def py_f2r(party, x, y, p, bw, shift):
    """truncation.cpp:583-637 (shift = 0) and :639-695"""
    mask, big, half = bw_mask(bw), p // 4, p // 2 + 1
    if party == 1:                                                 # :591-598
        tmp = (x - p + big) & M64 if x >= p - big else (x + big) & M64
    else:
        tmp = x
    msb = int(tmp >= half)                                         # :618-621
    cv = p >> shift
    corr = (0 if msb else cv) if party == 1 else 0                 # :603-606 / :661-664
    if party == 1:                                                 # :609-611 / :667-669
        y = ((cv - y) & M64) & mask if msb else ((-y) & M64) & mask
    if shift == 0:                                                 # :626-634
        out = (tmp - y - big) & M64 & mask if party == 1 else (tmp - y) & M64 & mask
    elif party == 1:                                               # :684-692
        out = ((((tmp - big) & M64) >> shift) - y) & M64 & mask
    else:
        out = ((tmp >> shift) - y + 1) & M64 & mask
    return tmp, msb, corr, out


def py_r2r128(party, x, y, bw_a, bw_b):
    """truncation.cpp:383-442; the result as a Python integer below 2^bw_b"""
    mask, msb_mask, big = bw_mask(bw_a), 1 << (bw_a - 1), 1 << (bw_a - 2)
    bmask = (1 << bw_b) - 1
    tmp = (x + big) & mask if party == 1 else x                    # :397-403
    msb = int(bool(tmp & msb_mask))                                # :424-427
    corr = msb ^ 1 if party == 1 else 0                            # :408-411
    if party == 1:
        y = (msb - y) & M64 & bw_mask(bw_b - bw_a)                 # :414-416
    out = (tmp - (y << bw_a) - (big if party == 1 else 0)) & bmask  # :430-438
    return tmp, msb, corr, out


# ------------------------------------------------------------------ inputs
def edge_mix(n, hi, seed):
    x = uniform_mod(seed, n, hi) if hi <= M64 else splitmix64(seed, n)
    edges = [0, 1, 2, hi - 1, hi - 2, hi // 2, hi // 2 - 1, hi // 2 + 1, hi // 4, hi // 4 - 1, hi // 4 + 1,
             3 * (hi // 4), 3 * (hi // 4) - 1, 3 * (hi // 4) + 1]
    x[:len(edges)] = np.array([e & M64 for e in edges], dtype=np.uint64)
    return x


def oracle_r2f(lib, party, x, y, bw, p):
    n = len(x)
    tmp, msb, corr, out = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    lib.mmo_r2f_pre(party, A(x), n, bw, p, A(tmp), A(msb), A(corr))
    lib.mmo_r2f_post(party, A(tmp), A(msb), A(y), n, bw, p, A(out))
    return tmp, msb, corr, out


def oracle_f2r(lib, party, x, y, p, bw, shift):
    n = len(x)
    tmp, msb, corr, out = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    lib.mmo_f2r_pre(party, A(x), n, p, shift, A(tmp), A(msb), A(corr))
    if shift:
        lib.mmo_f2r_trunc_post(party, A(tmp), A(msb), A(y), n, p, bw, shift, A(out))
    else:
        lib.mmo_f2r_post(party, A(tmp), A(msb), A(y), n, p, bw, A(out))
    return tmp, msb, corr, out


def oracle_r2r128(lib, party, x, y, bw_a, bw_b):
    n = len(x)
    tmp, msb, corr, out = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(2 * n, np.uint64)
    lib.mmo_r2r_pre(party, A(x), n, bw_a, A(tmp), A(msb), A(corr))
    lib.mmo_r2r128_post(party, A(tmp), A(msb), A(y), n, bw_a, bw_b, A(out))
    return tmp, msb, corr, out


def same(oracle_out, py_rows, wide=False):
    tmp, msb, corr, out = oracle_out
    for i, (t, m, c, o) in enumerate(py_rows):
        got = int(out[2 * i]) | int(out[2 * i + 1]) << 64 if wide else int(out[i])
        assert (int(tmp[i]), int(msb[i]), int(corr[i]), got) == (t, m, c, o), i


# ------------------------------------------------------------------ CPU: oracle == second restatement
@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_oracle_r2f_equals_python_restatement(party, bw, p):
    n = 600
    x, y = edge_mix(n, 1 << bw, SEED + bw), edge_mix(n, p, SEED + 1)
    x[20:34], y[20:34] = x[:14][::-1].copy(), y[:14].copy()       # edge x against edge y in both alignments
    same(oracle_r2f(load(), party, x, y, bw, p), [py_r2f(party, int(a), int(b), bw, p) for a, b in zip(x, y)])


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("shift", [0, 12])
@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_oracle_f2r_equals_python_restatement(party, bw, p, shift):
    if shift >= bw - 2:
        pytest.skip("shift wider than the ring")
    n = 600
    x, y = edge_mix(n, p, SEED + bw), edge_mix(n, 1 << bw, SEED + 2)
    x[20:34], y[20:34] = x[:14][::-1].copy(), y[:14].copy()
    same(oracle_f2r(load(), party, x, y, p, bw, shift),
         [py_f2r(party, int(a), int(b), p, bw, shift) for a, b in zip(x, y)])


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("s", EXT_SHIFTS)
def test_oracle_r2r128_equals_python_restatement(party, s):
    n, bw_a, bw_b = 600, 64, 64 + s
    x, y = edge_mix(n, 1 << 64, SEED + s), edge_mix(n, 1 << s, SEED + 3)
    same(oracle_r2r128(load(), party, x, y, bw_a, bw_b),
         [py_r2r128(party, int(a), int(b), bw_a, bw_b) for a, b in zip(x, y)], wide=True)


# ------------------------------------------------------------------ CPU: the reference tests' identities
def ref_values(n, bound, seed):
    """signed values in [-bound, bound) with the extremes forced in (ring_to_field_test.cpp:74-82)"""
    v = [int(r) % (2 * bound) - bound for r in splitmix64(seed, n)]
    v[:6] = [0, 1, -1, bound - 1, -bound, -bound + 1]
    return v


@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_ring_to_field_reference_identity(bw, p):
    """tests/ring_to_field_test.cpp:67-143: shares mod 2^bw of a value below min(2^(bw-2), p/4) come out as shares
    mod p of the same signed value"""
    lib, n, mask = load(), 3000, bw_mask(bw)
    v = ref_values(n, min(1 << (bw - 2), p // 4), SEED + bw)
    sb = edge_mix(n, 1 << bw, SEED + 5)
    sa = np.array([(a - int(b)) & mask for a, b in zip(v, sb)], dtype=np.uint64)
    x = edge_mix(n, p, SEED + 6)                                    # the sender's COT output
    pre = {1: oracle_r2f(lib, 1, sa, x, bw, p), 2: None}
    tmp2, msb2 = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    lib.mmo_r2f_pre(2, A(sb), n, bw, p, A(tmp2), A(msb2), None)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(pre[1][2]), A(msb2), n, 0, p, 1, A(y))
    out_b = oracle_r2f(lib, 2, sb, y, bw, p)[3]
    out_a = pre[1][3]
    for a, b, want in zip(out_a, out_b, v):
        got = (int(a) % p + int(b) % p) % p                        # add_mod, :52-57
        got = got - p if got >= (p + 1) // 2 else got              # getSignedVal_Field, :59-65
        assert got == want


@pytest.mark.parametrize("shift", [0, 12])
@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_field_to_ring_reference_identity(bw, p, shift):
    """tests/field_to_ring_test.cpp:60-135 (exact); the truncating variant within one unit (tests/bole_test.cpp:286-289)"""
    if shift and bw < 32:
        pytest.skip("the reference bounds the truncating variant's error at l = 32 only (bole_test.cpp:568): with a "
                    "16-bit prime p >> 12 = 15 drops 6 % of p / 2^12 and the error reaches 2")
    if shift and 64 - shift < bw:
        pytest.skip("ALICE's (tmp - p/4) >> s is a logical shift of a value that may have wrapped below zero in "
                    "uint64_t (truncation.cpp:686): only meaningful while 64 - s >= l; restated as written, compared "
                    "bit for bit above, no identity to check here")
    lib, n, mask = load(), 3000, bw_mask(bw)
    v = [abs(int(r) - (1 << 63)) % (p // 2) - p // 4 for r in splitmix64(SEED + bw, n)]   # :73-75
    v[:5] = [0, 1, -1, p // 2 - 1 - p // 4, -(p // 4)]
    sb = edge_mix(n, p, SEED + 7)
    sa = np.array([(a - int(b)) % p for a, b in zip(v, sb)], dtype=np.uint64)
    x = edge_mix(n, 1 << bw, SEED + 8)
    out_a = oracle_f2r(lib, 1, sa, x, p, bw, shift)
    tmp2, msb2 = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    lib.mmo_f2r_pre(2, A(sb), n, p, shift, A(tmp2), A(msb2), None)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(out_a[2]), A(msb2), n, bw, 0, 0, A(y))
    out_b = oracle_f2r(lib, 2, sb, y, p, bw, shift)[3]
    for a, b, want in zip(out_a[3], out_b, v):
        got = (int(a) + int(b)) & mask
        got = got - (1 << bw) if got >> (bw - 1) else got          # signed_val
        if not shift:
            assert got == want
    if shift:
        # a' + b = v + w p with a' = tmpA - p/4 (signed), w in {0, 1}; the parties output floor(a'/2^s) - share and
        # floor(b/2^s) + 1 - share with shares of w (p >> s), so the sum is off floor(v/2^s) by
        # 1 - floor((a0 + b0 - w (p mod 2^s)) / 2^s) in {0, 1, 2}; every batching prime is 1 mod 2^13, which leaves 2
        # to a0 = b0 = 0.  The reference checks |err| <= 1 against C's truncating division on random data
        # (tests/bole_test.cpp:286-289); both are asserted, the second off the forced edge values.
        for i, (ta, tb, a, b, want) in enumerate(zip(out_a[0], tmp2, out_a[3], out_b, v)):
            a1, b0 = int(ta) - p // 4, int(tb)
            w, rem = divmod(a1 + b0 - want, p)
            assert rem == 0 and w in (0, 1)
            got = (int(a) + int(b)) & mask
            got = got - (1 << bw) if got >> (bw - 1) else got
            assert got == (a1 >> shift) + (b0 >> shift) + 1 - w * (p >> shift)
            assert 0 <= got - (want >> shift) <= 2
            if i >= 20:
                assert abs(got - int(want / (1 << shift))) <= 1 or (a1 | b0) & ((1 << shift) - 1) == 0


@pytest.mark.parametrize("s", EXT_SHIFTS)
def test_ring_extension_reference_identity(s):
    """tests/ring_extension_test.cpp:34-119: 64 -> 64 + s bits through ring_to_ring128"""
    lib, n, bw_a, bw_b = load(), 3000, 64, 64 + s
    v = [(int(r) & ((1 << 63) - 1)) - (1 << 62) for r in splitmix64(SEED + s, n)]   # :44-54
    v[:5] = [0, 1, -1, (1 << 62) - 1, -(1 << 62)]
    sb = edge_mix(n, 1 << 64, SEED + 9)
    sa = np.array([(a - int(b)) & M64 for a, b in zip(v, sb)], dtype=np.uint64)
    x = edge_mix(n, 1 << s, SEED + 10)
    out_a = oracle_r2r128(lib, 1, sa, x, bw_a, bw_b)
    tmp2, msb2 = np.zeros(n, np.uint64), np.zeros(n, np.uint8)
    lib.mmo_r2r_pre(2, A(sb), n, bw_a, A(tmp2), A(msb2), None)
    y = np.zeros(n, np.uint64)
    lib.mmo_ideal_cot(A(x), A(out_a[2]), A(msb2), n, s, 0, 0, A(y))
    out_b = oracle_r2r128(lib, 2, sb, y, bw_a, bw_b)[3]
    for i, want in enumerate(v):
        a = int(out_a[3][2 * i]) | int(out_a[3][2 * i + 1]) << 64
        b = int(out_b[2 * i]) | int(out_b[2 * i + 1]) << 64
        got = (a + b) & ((1 << bw_b) - 1)
        got = got - (1 << bw_b) if got >> (bw_b - 1) else got      # signed_val_128, :100-106
        assert got == want


# ------------------------------------------------------------------ GPU: kernels == oracle at the same pairs
def _gpu(kind, party, x, y, **kw):
    from tests.test_convert_gpu import run_pre_post
    import openmm_amd
    return run_pre_post(openmm_amd, kind, party, x, y, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_gpu_r2f_reference_pairs(party, bw, p):
    n = 20000
    x, y = edge_mix(n, 1 << bw, SEED + bw), edge_mix(n, p, SEED + 1)
    x[20:34], y[20:34] = x[:14][::-1].copy(), y[:14].copy()
    tmp, msb, corr, out = oracle_r2f(load(), party, x, y, bw, p)
    g = _gpu("r2f", party, x, y, bw=bw, p=p)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
    if party == 1:
        assert np.array_equal(g[2], corr)


@pytest.mark.gpu
@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("shift", [0, 12])
@pytest.mark.parametrize("bw,p", REF_PAIRS)
def test_gpu_f2r_reference_pairs(party, bw, p, shift):
    if shift >= bw - 2:
        pytest.skip("shift wider than the ring")
    n = 20000
    x, y = edge_mix(n, p, SEED + bw), edge_mix(n, 1 << bw, SEED + 2)
    x[20:34], y[20:34] = x[:14][::-1].copy(), y[:14].copy()
    tmp, msb, corr, out = oracle_f2r(load(), party, x, y, p, bw, shift)
    g = _gpu("f2r", party, x, y, bw=bw, p=p, shift=shift)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3], out)
    if party == 1:
        assert np.array_equal(g[2], corr)


@pytest.mark.gpu
@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("s", EXT_SHIFTS)
def test_gpu_r2r128_reference_pairs(party, s):
    n = 20000
    x, y = edge_mix(n, 1 << 64, SEED + s), edge_mix(n, 1 << s, SEED + 3)
    tmp, msb, corr, out = oracle_r2r128(load(), party, x, y, 64, 64 + s)
    g = _gpu("r2r128", party, x, y, bw_a=64, bw_b=64 + s)
    assert np.array_equal(g[0], tmp) and np.array_equal(g[1], msb) and np.array_equal(g[3].reshape(-1), out)
    if party == 1:
        assert np.array_equal(g[2], corr)
