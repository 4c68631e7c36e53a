"""CPU: the oracle's restatement of the truncate family's local arithmetic (truncation-cheetah.cpp:2-218,
truncation.cpp:289-317) satisfies the reference's acceptance identity with an ideal wrap bit: both parties'
outputs add up to floor(x / 2^s) minus the dropped carry of the low parts (0 or 1), exactly for
truncate_and_reduce."""
import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64

SEED = 0x4D264D


def _shares(n, bw, secret):
    mask = (1 << bw) - 1
    a = [int(v) & mask for v in splitmix64(SEED + 1, n)]
    b = [(s - x) & mask for s, x in zip(secret, a)]
    return np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)


@pytest.mark.parametrize("signed_arith", [0, 1])
@pytest.mark.parametrize("heuristic", [0, 1])
@pytest.mark.parametrize("bw,shift", [(32, 12), (41, 7), (64, 12), (8, 5)])
def test_truncate_identity(signed_arith, heuristic, bw, shift):
    o, n, mask = load(), 4000, (1 << bw) - 1
    span = 1 << (bw - 3)
    rng = np.random.default_rng(SEED + bw)
    secret = [int(v) - (span // 2 if signed_arith else 0) for v in rng.integers(0, span, size=n, dtype=np.uint64)]
    xa, xb = _shares(n, bw, secret)
    adj, cmp, out = {}, {}, {}
    for party, x in ((1, xa), (2, xb)):
        adj[party], cmp[party] = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        o.mmo_trunc_pre(party, A(x), n, bw, signed_arith, heuristic, A(adj[party]), A(cmp[party]))
    wrap = np.array([int(a) + int(b) > mask for a, b in zip(adj[1], adj[2])], dtype=np.uint64)
    # mill->compare on the prepared operands computes the same bit (truncation-cheetah.cpp:51-58)
    assert np.array_equal(wrap, (cmp[1] > cmp[2]).astype(np.uint64))
    aw1 = splitmix64(SEED + 5, n) & np.uint64((1 << shift) - 1)
    aw2 = (wrap - aw1) & np.uint64((1 << shift) - 1)
    for party, aw in ((1, aw1), (2, aw2)):
        out[party] = np.zeros(n, np.uint64)
        o.mmo_trunc_post(party, A(adj[party]), A(aw), n, shift, bw, signed_arith, heuristic, A(out[party]))
    carries = 0
    for s, a, b, p, q in zip(secret, out[1], out[2], adj[1], adj[2]):
        carry = ((int(p) & ((1 << shift) - 1)) + (int(q) & ((1 << shift) - 1))) >> shift
        carries += carry
        assert (int(a) + int(b)) & mask == ((s >> shift) - carry) & mask
    assert 0 < carries < n   # both outcomes occur


@pytest.mark.parametrize("bw,shift", [(32, 12), (64, 20), (41, 1)])
def test_truncate_and_reduce_identity(bw, shift):
    o, n, mask = load(), 4000, (1 << bw) - 1
    secret = [int(v) & mask for v in splitmix64(SEED + 2, n)]
    xa, xb = _shares(n, bw, secret)
    la, lb = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_trunc_reduce_pre(A(xa), n, shift, A(la))
    o.mmo_trunc_reduce_pre(A(xb), n, shift, A(lb))
    wrap = np.array([(int(a) + int(b)) >> shift for a, b in zip(la, lb)], dtype=np.uint64)
    mo = np.uint64((1 << (bw - shift)) - 1)
    aw1 = splitmix64(SEED + 6, n) & mo
    aw2 = (wrap - aw1) & mo
    oa, ob = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_trunc_reduce_post(A(xa), A(aw1), n, shift, bw, A(oa))
    o.mmo_trunc_reduce_post(A(xb), A(aw2), n, shift, bw, A(ob))
    for s, a, b in zip(secret, oa, ob):
        assert (int(a) + int(b)) & int(mo) == (s >> shift) & int(mo)
