"""CPU: oracle FC and BN/BOLE flows meet the reference's acceptance identities."""
import numpy as np
import pytest

from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary
from tests.oracle_binding import OracleCtx, splitmix64, uniform_mod
from tests.oracle_layers import BnOracle, FcOracle

T = 1 << 37
P32 = 4293918721
N = 4096


def test_fc_split_matches_survey():
    o = OracleCtx(N, (60, 49), T)
    fc = FcOracle(o, 1001, 2048, True)
    assert (fc.d0, fc.d1, fc.n_in, fc.n_out) == (39, 103, 20, 26)


@pytest.mark.parametrize("rows,cols,shared", [(10, 37, True), (70, 300, True), (5, 5000, False), (200, 64, True)])
def test_fc_flow_reconstructs(rows, cols, shared):
    o = OracleCtx(N, (60, 49), T)
    fc = FcOracle(o, rows, cols, shared)
    sk, pk = keypair(o)
    vec = uniform_mod(SEED + 1, cols, T)
    s0 = uniform_mod(SEED + 2, cols, T)
    s1 = (vec - s0) & np.uint64(T - 1)
    W = signed_weights(SEED + 3, (rows, cols), 1 << 12, T)
    cts = fc.encrypt_input(sk, s0 if shared else vec, rns_uniform(o, SEED + 4, fc.n_in, 2), noise(SEED + 5, (fc.n_in, N)))
    out_cts, share = fc.server(cts, fc.encode_input(s1) if shared else None, fc.encode_weights(W),
                               np.ascontiguousarray(pk[:, :1]), ternary(SEED + 6, (fc.n_out, N)),
                               noise(SEED + 7, (fc.n_out, N)), rns_uniform(o, SEED + 8, fc.n_out, 1)[:, 0])
    dec = fc.decrypt(np.ascontiguousarray(sk[:1]), out_cts)
    ideal = (W.astype(object) @ vec.astype(object)) % T
    diff = (dec.astype(object) + share.astype(object) - ideal) % T
    assert all(d in (0, 1, T - 1) for d in diff)                       # library_fixed_uniform_cheetah.cpp:404


@pytest.mark.parametrize("n", [100, 4096, 10000])
def test_bole_and_bn_flow_reconstruct(n):
    """tests/bole_test.cpp:164-167: exact reconstruction of the cross terms x0*y1 + x1*y0"""
    o = OracleCtx(N, (60, 49), P32)
    bn = BnOracle(o, n)
    sk, _ = keypair(o)
    sk1 = np.ascontiguousarray(sk[:1])
    x0, y0, x1, y1 = (uniform_mod(SEED + i, n, P32) for i in range(4))
    ex = bn.encrypt(sk, x0, rns_uniform(o, SEED + 10, bn.n_ct, 2), noise(SEED + 11, (bn.n_ct, N)))
    ey = bn.encrypt(sk, y0, rns_uniform(o, SEED + 12, bn.n_ct, 2), noise(SEED + 13, (bn.n_ct, N)))
    random = splitmix64(SEED + 14, n)
    out_cts, share = bn.bole(ex, ey, bn.encode(x1), bn.encode(y1), random)
    dec = bn.decrypt(sk1, out_cts)
    want = (x0.astype(object) * y1.astype(object) + y0.astype(object) * x1.astype(object)) % P32
    assert np.array_equal((dec.astype(object) + share.astype(object)) % P32, want)
    # bn: (x0 + x1) * scale
    scale = uniform_mod(SEED + 20, n, P32)
    out_cts, share = bn.bn(ex, bn.encode(x1), bn.encode(scale), random)
    dec = bn.decrypt(sk1, out_cts)
    want = ((x0.astype(object) + x1.astype(object)) * scale.astype(object)) % P32
    assert np.array_equal((dec.astype(object) + share.astype(object)) % P32, want)


@pytest.mark.parametrize("ishape,shared", [((3, 10, 12), True), ((2, 64, 64), False), ((2, 70, 64), True)])
def test_bn_direct_flow_reconstructs(ishape, shared):
    """BatchNorm() wrapper (library_fixed_uniform_cheetah.cpp:465): out0 + out1 = x * scale mod 2^k,
    exact because the re-sharing is a sub_plain.  Tiles that are not ragged in w only: for those the
    reference's dense used-index / decrypt layout matches its encoder (hom_bn_ss.cc:666-678)."""
    from tests.oracle_layers import BnDirectOracle
    o = OracleCtx(N, (60, 49), T)
    bn = BnDirectOracle(o, ishape, shared)
    sk, pk = keypair(o)
    x0 = uniform_mod(SEED + 1, ishape, T)
    x1 = uniform_mod(SEED + 2, ishape, T) if shared else np.zeros(ishape, np.uint64)
    scales = ((splitmix64(SEED + 3, ishape[0]) % np.uint64(8191)).astype(np.int64) - 4095).view(np.uint64)  # both signs
    cts = bn.encrypt(sk, x0, rns_uniform(o, SEED + 10, bn.n_ct, 2), noise(SEED + 11, (bn.n_ct, N)))
    u, e0, e1 = ternary(SEED + 20, (bn.n_ct, N)), noise(SEED + 21, (bn.n_ct, N)), noise(SEED + 22, (bn.n_ct, N))
    share1 = uniform_mod(SEED + 30, ishape, T)
    out = bn.server(cts, bn.encode(x1) if shared else None, scales, pk, u, e0, e1, share1)
    dec = bn.decrypt(np.ascontiguousarray(sk[:1]), out)
    s = scales.view(np.int64).astype(object)[:, None, None]
    want = ((x0.astype(object) + x1.astype(object)) * s) % T
    assert np.array_equal((dec.astype(object) + share1.astype(object)) % T, want)


def test_fc_split_pruned_search_equals_exhaustive_scan():
    """mmh_fc_split (geometry.hpp fc_split: pruned search, what every FC plan uses) against the oracle's literal
    restatement of getSplit's double loop (hom_fc_ss.cc:36-48) over a rows x cols grid at N = 4096 and 8192: small
    shapes, shapes around N, primes, powers of two, the shipped networks' FC layers, first-minimum ties"""
    import ctypes as C
    from openmm_amd import lib as L
    from tests.oracle_binding import load
    from tests.oracle_layers import setup_geom
    lib, orc = L.load_library(), load()
    setup_geom(orc)
    dims = sorted(set(list(range(1, 34)) + [47, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 500, 512, 1000, 1001,
                                            1024, 2047, 2048, 2049, 4095, 4096, 4097, 5000, 8191, 8192, 8193, 9216,
                                            10007, 25088]))
    got, want = (C.c_int64 * 2)(), (C.c_long * 2)()
    n = 0
    for N in (4096, 8192):
        for rows in dims:
            for cols in dims:
                if N == 8192 and (rows > 2100 and cols > 2100):
                    continue          # the exhaustive scan is O(min(N,rows) * min(N,cols)); keep the sweep in seconds
                assert lib.mmh_fc_split(rows, cols, N, got) == 0
                orc.mmo_fc_split(rows, cols, N, want)
                assert (got[0], got[1]) == (want[0], want[1]), (N, rows, cols)
                n += 1
    assert n > 5000
