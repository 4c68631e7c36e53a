"""GPU parity of the truncate family's local share arithmetic (Truncation::truncate / truncate_msb /
truncate_msb0 of the USE_CHEETAH build, truncation-cheetah.cpp:2-218, and truncate_and_reduce,
truncation.cpp:289-317) against the oracle bit for bit, plus the reconstruction identities through the
Truncation mirror with ideal sub-protocols: floor(x / 2^s) up to the one unit Cheetah's truncation gives up
(the carry of the low parts is dropped), exactly for truncate_and_reduce."""
import threading

import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64
from tests.test_convert_gpu import dev, edge_mix, host

pytestmark = pytest.mark.gpu
SEED = 0x4D264D


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("signed_arith", [0, 1])
@pytest.mark.parametrize("heuristic", [0, 1])
@pytest.mark.parametrize("bw,shift", [(32, 12), (37, 12), (41, 1), (64, 20), (8, 5), (4, 1)])
def test_truncate_kernels(mm, party, signed_arith, heuristic, bw, shift):
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib, n = load(), load_library(), 20000
    x = edge_mix(n, 1 << bw, SEED + bw)
    if bw < 64:
        x[-3:] = np.array([(1 << bw) + 5, (1 << 63) + 1, (1 << 64) - 1], dtype=np.uint64)   # bits above bw
    aw = splitmix64(SEED + 7, n) & np.uint64((1 << shift) - 1)
    adj, cmp, out = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_trunc_pre(party, A(x), n, bw, signed_arith, heuristic, A(adj), A(cmp))
    o.mmo_trunc_post(party, A(adj), A(aw), n, shift, bw, signed_arith, heuristic, A(out))
    d_adj, d_cmp, d_out = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(3))
    d_x, d_aw = dev(x), dev(aw)
    check(lib.mmh_trunc_pre(party, dptr(d_x), n, bw, signed_arith, heuristic, dptr(d_adj), dptr(d_cmp), None))
    check(lib.mmh_trunc_post(party, dptr(d_adj), dptr(d_aw), n, shift, bw, signed_arith, heuristic, dptr(d_out),
                             None))
    assert np.array_equal(host(d_adj), adj) and np.array_equal(host(d_cmp), cmp)
    assert np.array_equal(host(d_out), out)
    # an odd length at 8 mod 16 (the one-element-per-lane kernels) and at 0 mod 16 (pairs, the last lane one)
    for lo in (1, 2):
        m = n - 1 - lo
        e_adj, e_cmp, e_out = (torch.empty(n, dtype=torch.int64, device="cuda")[lo:] for _ in range(3))
        check(lib.mmh_trunc_pre(party, dptr(d_x[lo:]), m, bw, signed_arith, heuristic, dptr(e_adj), dptr(e_cmp), None))
        check(lib.mmh_trunc_post(party, dptr(e_adj), dptr(d_aw[lo:]), m, shift, bw, signed_arith, heuristic,
                                 dptr(e_out), None))
        assert np.array_equal(host(e_adj)[:m], adj[lo:lo + m]) and np.array_equal(host(e_cmp)[:m], cmp[lo:lo + m])
        assert np.array_equal(host(e_out)[:m], out[lo:lo + m])


@pytest.mark.parametrize("bw,shift", [(32, 12), (41, 1), (64, 20), (64, 63)])
def test_truncate_and_reduce_kernels(mm, bw, shift):
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib, n = load(), load_library(), 20000
    x = edge_mix(n, 1 << bw, SEED + bw)
    aw = splitmix64(SEED + 8, n)
    lower, out = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_trunc_reduce_pre(A(x), n, shift, A(lower))
    o.mmo_trunc_reduce_post(A(x), A(aw), n, shift, bw, A(out))
    d_lower, d_out = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(2))
    d_x, d_aw = dev(x), dev(aw)   # named: a temporary's block would be handed to the next temporary
    check(lib.mmh_trunc_reduce_pre(dptr(d_x), n, shift, dptr(d_lower), None))
    check(lib.mmh_trunc_reduce_post(dptr(d_x), dptr(d_aw), n, shift, bw, dptr(d_out), None))
    assert np.array_equal(host(d_lower), lower) and np.array_equal(host(d_out), out)


def test_truncate_argument_checks(mm):
    from openmm_amd.lib import load_library
    lib = load_library()
    assert lib.mmh_trunc_post(3, None, None, 0, 1, 32, 0, 0, None, None) == 7        # party: ERR_INVALID_ARG
    assert lib.mmh_trunc_post(1, None, None, 0, 32, 32, 0, 0, None, None) == 7       # truncating the full width
    assert lib.mmh_trunc_post(2, None, None, 0, 30, 32, 0, 1, None, None) == 7       # heuristic offset 2^(bw-3-shift)
    assert lib.mmh_trunc_post(1, None, None, 5, 3, 32, 0, 0, None, None) == 2        # null pointers with n > 0


def _two_parties(fn):
    """run fn(party) for ALICE and BOB in two threads; returns their results"""
    res, err = {}, []

    def body(party):
        try:
            res[party] = fn(party)
        except Exception as e:   # pragma: no cover
            err.append(e)
    th = [threading.Thread(target=body, args=(p,)) for p in (1, 2)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not err, err
    return res[1], res[2]


def _signed(v, bw):
    v &= (1 << bw) - 1
    return v - (1 << bw) if v >> (bw - 1) else v


@pytest.mark.parametrize("variant", ["truncate", "truncate_heuristic", "truncate_msb", "truncate_msb0"])
@pytest.mark.parametrize("signed_arith", [False, True])
@pytest.mark.parametrize("bw,shift", [(32, 12), (41, 12), (64, 12)])
def test_truncate_reconstructs(mm, variant, signed_arith, bw, shift):
    from openmm_amd.truncation import ALICE, IdealAux, Truncation
    n, mask = 5000, (1 << bw) - 1
    # the secret: anything for the plain variants, the heuristic's range (no wrap after + 2^(bw-3)) for that one
    span = 1 << (bw - 3)
    rng = np.random.default_rng(SEED + bw + shift)
    if signed_arith:
        secret = [int(v) - span // 2 for v in rng.integers(0, span, size=n, dtype=np.uint64)]
    else:
        secret = [int(v) for v in rng.integers(0, span, size=n, dtype=np.uint64)]
    share_a = [int(v) & mask for v in splitmix64(SEED + 1, n)]
    share_b = [(s - a) & mask for s, a in zip(secret, share_a)]
    shares = {1: np.array(share_a, dtype=np.uint64), 2: np.array(share_b, dtype=np.uint64)}
    aux = IdealAux()

    def run(party):
        tr = Truncation(party, None, aux=aux)
        x = dev(shares[party])
        if variant == "truncate":
            out = tr.truncate(n, x, shift, bw, signed_arith, None, False)
        elif variant == "truncate_heuristic":
            out = tr.truncate(n, x, shift, bw, signed_arith, None, True)
        elif variant == "truncate_msb":
            out = tr.truncate_msb(n, x, shift, bw, signed_arith, np.zeros(n, np.uint8))
        else:
            out = tr.truncate_msb0(n, x, shift, bw, signed_arith)
        return host(out)
    oa, ob = _two_parties(run)
    for s, a, b in zip(secret, oa, ob):
        got = (int(a) + int(b)) & mask
        want = (s >> shift) & mask          # floor division, also for negative secrets
        assert got in (want, (want - 1) & mask)
    if signed_arith:   # and the value read back as a signed number is the arithmetic shift
        a, b, s = int(oa[0]), int(ob[0]), secret[0]
        assert _signed(a + b, bw) in (s >> shift, (s >> shift) - 1)


@pytest.mark.parametrize("bw,shift", [(32, 12), (41, 12), (64, 12)])
def test_truncate_and_reduce_reconstructs(mm, bw, shift):
    from openmm_amd.truncation import IdealAux, Truncation
    n, mask, mask_out = 5000, (1 << bw) - 1, (1 << (bw - shift)) - 1
    secret = [int(v) & mask for v in splitmix64(SEED + 2, n)]
    share_a = [int(v) & mask for v in splitmix64(SEED + 3, n)]
    share_b = [(s - a) & mask for s, a in zip(secret, share_a)]
    shares = {1: np.array(share_a, dtype=np.uint64), 2: np.array(share_b, dtype=np.uint64)}
    aux = IdealAux()
    oa, ob = _two_parties(lambda party: host(Truncation(party, None, aux=aux).truncate_and_reduce(
        n, dev(shares[party]), shift, bw)))
    for s, a, b in zip(secret, oa, ob):
        assert (int(a) + int(b)) & mask_out == (s >> shift) & mask_out
