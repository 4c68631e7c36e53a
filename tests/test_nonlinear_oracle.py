"""CPU: the oracle's restatement of the ReLU / multiplexer / B2A local arithmetic (relu-ring.h:216-313,
aux-protocols.cpp:62-126) composed with the ideal COT gives ReLU(x) and the bit exactly."""
import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64

SEED = 0x4D264D


def _cot(o, sender_x, corr, choice, n, l):
    """ideal COT: the sender keeps sender_x, the receiver gets sender_x + choice * corr mod 2^l"""
    recv = np.zeros(n, np.uint64)
    o.mmo_ideal_cot(A(sender_x), A(corr), A(choice), n, l, 0, 0, A(recv))
    mask = np.uint64((1 << l) - 1) if l < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    return sender_x & mask, recv


@pytest.mark.parametrize("l", [32, 41, 64])
def test_relu_identity(l):
    o, n, mask = load(), 3000, (1 << l) - 1
    rng = np.random.default_rng(SEED + l)
    secret = [int(v) - (1 << (l - 2)) for v in rng.integers(0, 1 << (l - 1), size=n, dtype=np.uint64)]
    sa = [int(v) & mask for v in splitmix64(SEED + 1, n)]
    sh = {1: np.array(sa, dtype=np.uint64), 2: np.array([(s - a) & mask for s, a in zip(secret, sa)], dtype=np.uint64)}
    msb, cmp, sel, drelu = {}, {}, {}, {}
    for p in (1, 2):
        msb[p], cmp[p] = np.zeros(n, np.uint8), np.zeros(n, np.uint64)
        o.mmo_relu_pre(p, A(sh[p]), n, l, A(msb[p]), A(cmp[p]))
    gt = (cmp[1] > cmp[2]).astype(np.uint8)            # millionaire: ALICE's > BOB's
    r = (splitmix64(SEED + 2, n) & np.uint64(1)).astype(np.uint8)
    wrap = {1: r, 2: gt ^ r}
    for p in (1, 2):
        drelu[p], sel[p] = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        o.mmo_relu_post(p, A(msb[p]), A(wrap[p]), n, A(drelu[p]), A(sel[p]))
    # multiplexer: straight COT (ALICE sends, BOB chooses with his sel), reversed COT (BOB sends)
    corr = {}
    for p in (1, 2):
        corr[p] = np.zeros(n, np.uint64)
        o.mmo_mux_pre(A(sel[p]), A(sh[p]), n, l, A(corr[p]))
    sA, rB = _cot(o, splitmix64(SEED + 3, n), corr[1], sel[2], n, l)
    sB, rA = _cot(o, splitmix64(SEED + 4, n), corr[2], sel[1], n, l)
    y = {}
    for p, (dr, ds) in ((1, (rA, sA)), (2, (rB, sB))):
        y[p] = np.zeros(n, np.uint64)
        o.mmo_mux_post(A(sel[p]), A(sh[p]), A(dr), A(ds), n, l, A(y[p]))
    for s, a, b, d1, d2 in zip(secret, y[1], y[2], drelu[1], drelu[2]):
        assert (int(d1) ^ int(d2)) == (1 if s < 0 else 0)
        assert (int(a) + int(b)) & mask == max(s, 0) & mask


@pytest.mark.parametrize("bw", [1, 12, 64])
def test_b2a_identity(bw):
    o, n = load(), 3000
    bit = (splitmix64(SEED + 9, n) & np.uint64(1)).astype(np.uint8)
    xa = (splitmix64(SEED + 10, n) & np.uint64(1)).astype(np.uint8)
    xb = bit ^ xa
    corr = np.zeros(n, np.uint64)
    o.mmo_b2a_pre(A(xa), n, bw, A(corr))
    ys, yr = _cot(o, splitmix64(SEED + 11, n), corr, xb, n, bw)
    oa, ob = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_b2a_post(1, A(xa), A(ys), n, bw, A(oa))
    o.mmo_b2a_post(2, A(xb), A(yr), n, bw, A(ob))
    mask = (1 << bw) - 1
    for t, a, b in zip(bit, oa, ob):
        assert ((int(a) ^ int(b)) if bw == 1 else (int(a) + int(b)) & mask) == int(t)
