"""GPU parity of the HomFCSS and HomBNSS (BOLE) layer calls vs the oracle flows, bit for bit."""
import os

import numpy as np
import pytest

from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary
from tests.oracle_binding import OracleCtx, splitmix64, uniform_mod
from tests.oracle_layers import BnOracle, FcOracle

pytestmark = pytest.mark.gpu
T = 1 << 37
P32 = 4293918721
N = 4096


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
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("rows,cols,shared", [(10, 37, True), (70, 300, True), (5, 5000, False), (200, 64, True),
                                              (1001, 2048, True), (1, 1, True), (4096, 1, True)])
def test_fc_layer_bit_exact(mm, rows, cols, shared):
    from openmm_amd.hom_fc_ss import HomFCSS, Meta
    o, g = OracleCtx(N, (60, 49), T), mm.Context(N, (60, 49), T)
    fc = FcOracle(o, rows, cols, shared)
    meta = Meta((rows, cols), (cols,), shared)
    sk, pk = keypair(o)
    client, server = HomFCSS(), HomFCSS()
    client.setUp(g, sk_ntt=dev(sk))
    server.setUp(g, pk_ntt=dev(pk))
    plan = server.plan(meta)
    assert plan.split == (fc.d0, fc.d1) and (plan.n_in, plan.n_out) == (fc.n_in, fc.n_out)
    vec = uniform_mod(SEED + 1, cols, T)
    vec[:min(3, cols)] = [0, 1, T - 1][:min(3, cols)]
    s0 = uniform_mod(SEED + 2, cols, T)
    s1 = (vec - s0) & np.uint64(T - 1)
    W = signed_weights(SEED + 3, (rows, cols), 1 << 12, T)
    W[W == 0] = 1  # a block that encodes to zero is rejected (hom_fc_ss.cc:414-422)
    a, e = rns_uniform(o, SEED + 4, fc.n_in, 2), noise(SEED + 5, (fc.n_in, N))
    cin = s0 if shared else vec
    ref_cts = fc.encrypt_input(sk, cin, a, e)
    got_cts = client.encryptInputVector(dev(cin), meta, (dev(a), dev(e)))
    assert np.array_equal(host(got_cts), ref_cts)
    ref_w, got_w = fc.encode_weights(W), server.encodeWeightMatrix(dev(W), meta)
    assert np.array_equal(host(got_w), ref_w)
    ref_s = got_s = None
    if shared:
        ref_s, got_s = fc.encode_input(s1), server.encodeInputVector(dev(s1), meta)
        assert np.array_equal(host(got_s), ref_s)
    u, e1 = ternary(SEED + 6, (fc.n_out, N)), noise(SEED + 7, (fc.n_out, N))
    mask = rns_uniform(o, SEED + 8, fc.n_out, 1)[:, 0]
    ref_out, ref_share = fc.server(ref_cts, ref_s, ref_w, np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    got_out, got_share = server.matVecMul(got_w, got_cts, got_s, meta, (dev(u), dev(e1), dev(mask)))
    # (the fused form of the re-sharing tail too: the default picks the split form for so few ciphertexts, bfv.hip)
    os.environ["MMH_TAIL_SPLIT_MAX"] = "0"
    try:
        f_out, f_share = server.matVecMul(got_w, got_cts, got_s, meta, (dev(u), dev(e1), dev(mask)))
    finally:
        del os.environ["MMH_TAIL_SPLIT_MAX"]
    assert np.array_equal(host(f_out), host(got_out)) and np.array_equal(host(f_share), host(got_share))
    assert np.array_equal(host(got_out), ref_out)
    assert np.array_equal(host(got_share), ref_share)
    # USE_APPROX_RESHARE=0 build: bit-exact against the oracle, and the shares reconstruct exactly
    flood, e0 = splitmix64(SEED + 9, fc.n_out * N).reshape(fc.n_out, N), noise(SEED + 10, (fc.n_out, N))
    words = splitmix64(SEED + 11, fc.n_out * N).reshape(fc.n_out, N)
    x_out, x_share = fc.server(ref_cts, ref_s, ref_w, np.ascontiguousarray(pk[:, :1]), u, e1, words, exact=(flood, e0))
    y_out, y_share = server.matVecMul(got_w, got_cts, got_s, meta, (dev(flood), dev(u), dev(e0), dev(e1), dev(words)),
                                      exact=True)
    assert np.array_equal(host(y_out), x_out) and np.array_equal(host(y_share), x_share)
    y_dec = host(client.decryptToVector(y_out, meta))
    assert np.array_equal((y_dec.astype(object) + x_share.astype(object)) % T, (W.astype(object) @ vec.astype(object)) % T)
    got_dec = host(client.decryptToVector(got_out, meta))
    assert np.array_equal(got_dec, fc.decrypt(np.ascontiguousarray(sk[:1]), ref_out))
    ideal = (W.astype(object) @ vec.astype(object)) % T
    diff = (got_dec.astype(object) + host(got_share).astype(object) - ideal) % T
    assert all(d in (0, 1, T - 1) for d in diff)


def test_fc_rejects_zero_block(mm):
    import torch
    from openmm_amd.hom_fc_ss import HomFCSS, Meta
    g = mm.Context(N, (60, 49), T)
    h = HomFCSS()
    h.setUp(g, pk_ntt=torch.zeros((2, 2, N), dtype=torch.int64, device="cuda"))
    meta = Meta((4, 8), (8,), False)
    p = h.plan(meta)
    w = torch.zeros((p.n_out, p.n_in, N), dtype=torch.int64, device="cuda")
    cts = torch.zeros((p.n_in, 2, 2, N), dtype=torch.int64, device="cuda")
    z = torch.zeros((p.n_out, N), dtype=torch.int64, device="cuda")
    with pytest.raises(mm.MMHError) as ei:   # the verdict is read back after the kernels are enqueued
        h.matVecMul(w, cts, None, meta, (z, z, z))
    assert ei.value.code == 7  # ERR_INVALID_ARG, hom_fc_ss.cc:419


@pytest.mark.parametrize("ring_bits,prime", [(32, 4293918721), (41, 2198100901889), (37, 137438822401)])
def test_fc_and_bole_bit_exact_other_moduli(mm, monkeypatch, ring_bits, prime):
    """the other rings and their fields of sci::default_prime_mod (SCI/src/utils/constants.h:45-50; BASELINE runs 37 for
    the FC and the 32-bit prime for BOLE): the tails' constants depend on t, the batching tables on p"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "T", 1 << ring_bits)
    monkeypatch.setattr(sys.modules[__name__], "P32", prime)
    test_fc_layer_bit_exact(mm, 70, 300, True)
    test_bole_bn_bit_exact(mm, 10000)


def test_fc_and_bole_bit_exact_n8192(mm, monkeypatch):
    """the FC and BOLE / bn flows on N = 8192 contexts (the reference's are N = 4096, cheetah-api.cpp:101,152; the entry
    points take either): the LOGN = 13 kernels and the batching tables of the larger ring against the oracle"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "N", 8192)
    test_fc_layer_bit_exact(mm, 70, 300, True)
    test_fc_layer_bit_exact(mm, 5, 9000, False)
    test_bole_bn_bit_exact(mm, 100)
    test_bole_bn_bit_exact(mm, 20000)


@pytest.mark.parametrize("n", [1, 100, 4096, 10000, 1 << 20])   # 2^20: tests/bole_test.cpp:568, BASELINE configs[0,2]
def test_bole_bn_bit_exact(mm, n):
    from openmm_amd.hom_bn_ss import HomBNSS, Meta
    from openmm_amd.truncation import bole_cross_term
    o, g = OracleCtx(N, (60, 49), P32), mm.Context(N, (60, 49), P32)
    bn = BnOracle(o, n)
    meta = Meta((n,), P32, True)
    sk, _ = keypair(o)
    sk1 = np.ascontiguousarray(sk[:1])
    client, server = HomBNSS(), HomBNSS()
    client.setUp_Field(P32, g, sk_ntt=dev(sk))
    server.setUp_Field(P32, g)
    # ring_to_field outputs are only congruent mod p: feed values in [0, 2^32)
    x0, y0, x1, y1 = (splitmix64(SEED + i, n) & np.uint64((1 << 32) - 1) for i in range(4))
    ax, ex_ = rns_uniform(o, SEED + 10, bn.n_ct, 2), noise(SEED + 11, (bn.n_ct, N))
    ay, ey_ = rns_uniform(o, SEED + 12, bn.n_ct, 2), noise(SEED + 13, (bn.n_ct, N))
    ref_ex, ref_ey = bn.encrypt(sk, x0, ax, ex_), bn.encrypt(sk, y0, ay, ey_)
    got_ex = client.encryptVector(dev(x0), meta, (dev(ax), dev(ex_)))
    got_ey = client.encryptVector(dev(y0), meta, (dev(ay), dev(ey_)))
    assert np.array_equal(host(got_ex), ref_ex) and np.array_equal(host(got_ey), ref_ey)
    ref_px, ref_py = bn.encode(x1), bn.encode(y1)
    got_px, got_py = server.encodeVector(dev(x1), meta), server.encodeVector(dev(y1), meta)
    assert np.array_equal(host(got_px), ref_px) and np.array_equal(host(got_py), ref_py)
    random = splitmix64(SEED + 14, n)
    ref_out, ref_share = bn.bole(ref_ex, ref_ey, ref_px, ref_py, random)
    got_out, got_share = server.bole(got_ex, got_ey, got_px, got_py, meta, dev(random))
    assert np.array_equal(host(got_out), ref_out) and np.array_equal(host(got_share), ref_share)
    assert np.array_equal(host(got_ex), ref_ex)  # inputs are not destroyed
    got_dec = client.decryptToVector(got_out, meta)
    assert np.array_equal(host(got_dec), bn.decrypt(sk1, ref_out))
    # CheetahLinear::BOLE tail on both sides (cheetah-api.cpp:648-649) and the protocol identity
    # tests/bole_test.cpp:164-167: out0 + out1 = (x0 + x1) * (y0 + y1) mod p
    out_b = bole_cross_term(got_dec.clone(), dev(x0), dev(y0), P32)
    out_a = bole_cross_term(got_share.clone(), dev(x1), dev(y1), P32)
    want = ((x0.astype(object) + x1.astype(object)) * (y0.astype(object) + y1.astype(object))) % P32
    assert np.array_equal((host(out_a).astype(object) + host(out_b).astype(object)) % P32, want)
    # bn with the scale vector in clear on the server (cheetah-api.cpp:527)
    scale = uniform_mod(SEED + 20, n, P32)
    ref_sc, got_sc = bn.encode(scale), server.encodeScales(dev(scale), meta)
    ref_out, ref_share = bn.bn(ref_ex, ref_px, ref_sc, random)
    got_out, got_share = server.bn(got_ex, got_px, got_sc, meta, dev(random))
    assert np.array_equal(host(got_out), ref_out) and np.array_equal(host(got_share), ref_share)
    meta_ns = Meta((n,), P32, False)
    ref_out, ref_share = bn.bn(ref_ex, None, ref_sc, random)
    got_out, got_share = server.bn(got_ex, None, got_sc, meta_ns, dev(random))
    assert np.array_equal(host(got_out), ref_out) and np.array_equal(host(got_share), ref_share)


@pytest.mark.parametrize("n", [1, 100, 4096, 10000, 70001])
@pytest.mark.parametrize("party", [1, 2])
def test_bn_vector_steps_fused_with_conversions(mm, n, party):
    """ring_to_field's post step straight into encodeVector / encryptVector, and decryptToVector straight into
    field_to_ring's pre step (cheetah-api.cpp:493-561 runs them back to back): the fused entry points against the
    oracle's two steps, bit for bit"""
    from openmm_amd.hom_bn_ss import HomBNSS, Meta
    from tests.oracle_binding import A, load
    ol = load()
    o, g = OracleCtx(N, (60, 49), P32), mm.Context(N, (60, 49), P32)
    bn = BnOracle(o, n)
    meta = Meta((n,), P32, True)
    sk, _ = keypair(o)
    h = HomBNSS()
    h.setUp_Field(P32, g, sk_ntt=dev(sk))
    bw = 37
    x = splitmix64(SEED + 30, n) & np.uint64((1 << bw) - 1)
    x[:min(3, n)] = [0, (1 << bw) - 1, 1 << (bw - 1)][:min(3, n)]
    y = splitmix64(SEED + 31, n) & np.uint64((1 << bw) - 1)
    tmp, msb, corr, fld = (np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64),
                           np.zeros(n, np.uint64))
    ol.mmo_r2f_pre(party, A(x), n, bw, P32, A(tmp), A(msb), A(corr))
    ol.mmo_r2f_post(party, A(tmp), A(msb), A(y), n, bw, P32, A(fld))
    conv = (dev(tmp), dev(msb), dev(y))
    assert np.array_equal(host(h.encodeVectorFromRing(party, conv, bw, meta)), bn.encode(fld))
    a, e = rns_uniform(o, SEED + 32, bn.n_ct, 2), noise(SEED + 33, (bn.n_ct, N))
    cts = h.encryptVectorFromRing(party, conv, bw, meta, (dev(a), dev(e)))
    ref_cts = bn.encrypt(sk, fld, a, e)
    assert np.array_equal(host(cts), ref_cts)
    # the way back: decrypt one-limb results (flood and drop the special prime as bn does) into the pre step
    scale, rnd = bn.encode(uniform_mod(SEED + 34, n, P32)), splitmix64(SEED + 35, n)
    out_ct, _ = bn.bn(ref_cts, bn.encode(np.zeros(n, np.uint64)), scale, rnd)
    dec = bn.decrypt(np.ascontiguousarray(sk[:1]), out_ct)
    for shift in (0, 12):
        ol.mmo_f2r_pre(party, A(dec), n, P32, shift, A(tmp), A(msb), A(corr))
        g_tmp, g_msb, g_corr = h.decryptToVectorForRing(dev(out_ct), meta, party, shift)
        assert np.array_equal(host(g_tmp), tmp) and np.array_equal(g_msb.cpu().numpy(), msb)
        if party == 1:
            assert np.array_equal(host(g_corr), corr)
    with pytest.raises(mm.MMHError):
        h.decryptToVectorForRing(dev(out_ct), meta, 3)


@pytest.mark.parametrize("n", [7 * 7 * 2048, 56 * 56 * 256])
def test_network_bn_layer_bit_exact(mm, n):
    """One live FusedBatchNorm4411 of ResNet50 (networks/main_resnet50.cpp:1060-1093 -> ElemWiseActModelVectorMult,
    library_fixed_uniform_cheetah.cpp:567-600) at the network's own sizes -- 56*56*256 = 802,816 elements = 196
    ciphertexts is the largest of its 18 -- with the shipped ring and field (ell = 37, p = 137438822401, k = 12):
    Ring2Field -> CheetahLinear::bn -> Field2Ring_with_Truncate, both parties, every device step bit for bit against
    the oracle (the conversions' post / pre steps through the entry points fused with the vector steps), the COT by
    the oracle's ideal functionality, and the result within one unit of floor(x * scale / 2^k)
    (tests/bole_test.cpp:286-289)."""
    import torch
    from openmm_amd.hom_bn_ss import HomBNSS, Meta
    from tests.oracle_binding import A, load
    ol = load()
    bw, p, k = 37, 137438822401, 12
    o, g = OracleCtx(N, (60, 49), p), mm.Context(N, (60, 49), p)
    bn = BnOracle(o, n)
    meta = Meta((n,), p, True)
    sk, _ = keypair(o)
    client, server = HomBNSS(), HomBNSS()
    client.setUp_Field(p, g, sk_ntt=dev(sk))
    server.setUp_Field(p, g)
    ring = np.uint64((1 << bw) - 1)
    # activations at scale k with |x| < 2^20, BN scales |s| < 2^13 (fixed point, scale k): |x s| < 2^33 < p / 4
    x = (splitmix64(SEED + 40, n) % np.uint64(1 << 21)).astype(np.int64) - (1 << 20)
    sc = (splitmix64(SEED + 41, n) % np.uint64(1 << 14)).astype(np.int64) - (1 << 13)
    xa = splitmix64(SEED + 42, n) & ring
    xb = (x.astype(np.uint64) - xa) & ring
    scale_f = np.where(sc < 0, np.uint64(p) - (-sc).astype(np.uint64), sc.astype(np.uint64)).astype(np.uint64)  # getFieldElt
    z = lambda dt=np.uint64: np.zeros(n, dt)

    def ideal_cot(corr, choice, l, prime):   # SilentOT's functionality (silent_ot.h:85-242) through the oracle
        y0 = uniform_mod(SEED + 50 + l, n, prime) if prime else splitmix64(SEED + 50 + l, n) & np.uint64((1 << l) - 1)
        y1 = z()
        ol.mmo_ideal_cot(A(y0), A(corr), A(choice), n, l, prime, 1 if prime else 0, A(y1))
        return y0, y1

    # Ring2Field (truncation.cpp:518-581): ALICE is the COT sender
    ta, ma, ca, tb, mb, cb = z(), z(np.uint8), z(), z(), z(np.uint8), z()
    ol.mmo_r2f_pre(1, A(xa), n, bw, p, A(ta), A(ma), A(ca))
    ol.mmo_r2f_pre(2, A(xb), n, bw, p, A(tb), A(mb), A(cb))
    from openmm_amd import lib as L
    lib = L.load_library()
    d_t, d_m, d_c = dev(z()), torch.zeros(n, dtype=torch.uint8, device="cuda"), dev(z())
    L.check(lib.mmh_r2f_pre(1, L.dptr(dev(xa)), n, bw, p, L.dptr(d_t), L.dptr(d_m), L.dptr(d_c), L._stream_ptr()))
    assert np.array_equal(host(d_t), ta) and np.array_equal(d_m.cpu().numpy(), ma) and np.array_equal(host(d_c), ca)
    ya, yb = ideal_cot(ca, mb, 0, p)
    fa, fb = z(), z()
    ol.mmo_r2f_post(1, A(ta), A(ma), A(ya), n, bw, p, A(fa))
    ol.mmo_r2f_post(2, A(tb), A(mb), A(yb), n, bw, p, A(fb))
    # CheetahLinear::bn (cheetah-api.cpp:527-603)
    a, e = rns_uniform(o, SEED + 43, bn.n_ct, 2), noise(SEED + 44, (bn.n_ct, N))
    ref_cts = bn.encrypt(sk, fb, a, e)
    got_cts = client.encryptVectorFromRing(2, (dev(tb), dev(mb), dev(yb)), bw, meta, (dev(a), dev(e)))
    assert np.array_equal(host(got_cts), ref_cts)
    ref_in = bn.encode(fa)
    got_in = server.encodeVectorFromRing(1, (dev(ta), dev(ma), dev(ya)), bw, meta)
    assert np.array_equal(host(got_in), ref_in)
    ref_sc, got_sc = bn.encode(scale_f), server.encodeScales(dev(scale_f), meta)
    assert np.array_equal(host(got_sc), ref_sc)
    rnd = splitmix64(SEED + 45, n)
    ref_out, ref_share = bn.bn(ref_cts, ref_in, ref_sc, rnd)
    got_out, got_share = server.bn(got_cts, got_in, got_sc, meta, dev(rnd))
    assert np.array_equal(host(got_out), ref_out) and np.array_equal(host(got_share), ref_share)
    # Field2Ring_with_Truncate (truncation.cpp:639-695), shift = kScale
    dec = bn.decrypt(np.ascontiguousarray(sk[:1]), ref_out)
    ol.mmo_f2r_pre(1, A(ref_share), n, p, k, A(ta), A(ma), A(ca))
    ol.mmo_f2r_pre(2, A(dec), n, p, k, A(tb), A(mb), A(cb))
    g_tb, g_mb, _ = client.decryptToVectorForRing(got_out, meta, 2, k)
    assert np.array_equal(host(g_tb), tb) and np.array_equal(g_mb.cpu().numpy(), mb)
    L.check(lib.mmh_f2r_pre(1, L.dptr(got_share), n, p, k, L.dptr(d_t), L.dptr(d_m), L.dptr(d_c), L._stream_ptr()))
    assert np.array_equal(host(d_t), ta) and np.array_equal(d_m.cpu().numpy(), ma) and np.array_equal(host(d_c), ca)
    ya, yb = ideal_cot(ca, mb, bw, 0)
    oa, ob = z(), z()
    ol.mmo_f2r_trunc_post(1, A(ta), A(ma), A(ya), n, p, bw, k, A(oa))
    ol.mmo_f2r_trunc_post(2, A(tb), A(mb), A(yb), n, p, bw, k, A(ob))
    d_oa, d_ob = dev(z()), dev(z())
    L.check(lib.mmh_f2r_trunc_post(1, L.dptr(d_t), L.dptr(d_m), L.dptr(dev(ya)), n, p, bw, k, L.dptr(d_oa), L._stream_ptr()))
    L.check(lib.mmh_f2r_trunc_post(2, L.dptr(g_tb), L.dptr(g_mb), L.dptr(dev(yb)), n, p, bw, k, L.dptr(d_ob),
                                   L._stream_ptr()))
    assert np.array_equal(host(d_oa), oa) and np.array_equal(host(d_ob), ob)
    got = (oa + ob) & ring
    got = np.where(got >> np.uint64(bw - 1), got.astype(np.int64) - (1 << bw), got.astype(np.int64))
    # field_to_ring_with_truncate gives floor(v / 2^k) + 1 - c + c2 (c: carry of the two low parts; c2: 1 only when the
    # wrap bit is set and v = -1 mod 2^k, p being 1 mod 2^13): floor + {0, 1}, and floor + 2 with probability ~2^-24
    # (tests/cpp/network_linear.cc has the derivation); tests/bole_test.cpp:286-289 accepts one unit around the
    # quotient rounded towards zero, which contains floor + {0, 1} for either sign
    d = got - ((x * sc) >> k)
    assert d.min() >= 0 and d.max() <= 2 and int((d == 2).sum()) <= 2


@pytest.mark.parametrize("n", [5, 1000, 4096, 3 * 4096 + 77, 7 * 7 * 1024, 56 * 56 * 256])
def test_alice_bn_layer_in_two_calls(mm, n):
    """ALICE's side of a batch-norm layer as mmh_bn_alice_prepare + mmh_bn_alice_finish (five launches): the three
    plaintext batches, her share, her ring share of the result, every byte of both COTs' wires and the result
    ciphertexts against the ORACLE's separate steps (Ring2Field as the COT sender, HomBNSS::encodeVector twice,
    HomBNSS::bn with addMask, Field2Ring[_with_Truncate] as the COT sender) on the same blocks;
    r's words are the ChaCha20 stream that tests/test_sampler_gpu.py pins against RFC 8439."""
    import ctypes as C
    import torch
    from openmm_amd import lib as L
    from tests.oracle_binding import A, load
    ol, lib = load(), L.load_library()
    bw, p, gid0 = 37, 137438822401, 8 * 77
    o, g = OracleCtx(N, (60, 49), p), mm.Context(N, (60, 49), p)
    bn = BnOracle(o, n)
    u2 = lambda v: (C.c_uint64 * 2)(int(v[0]), int(v[1]))
    delta = splitmix64(SEED + 80, 2) | np.uint64(1)
    S, rcm = splitmix64(SEED + 81, 2), splitmix64(SEED + 82, 2 * n)
    x = splitmix64(SEED + 83, n) & np.uint64((1 << bw) - 1)
    scale_f = splitmix64(SEED + 84, n) % np.uint64(p)
    cts = np.stack([rns_uniform(o, SEED + 85, bn.n_ct, 2), rns_uniform(o, SEED + 86, bn.n_ct, 2)], axis=1)
    key = (C.c_uint32 * 8)(*[int(v) for v in splitmix64(SEED + 87, 8) & np.uint64(0xFFFFFFFF)])
    sid = 0x1234
    # the oracle's steps
    tmp, msb, corr, fld = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    ol.mmo_r2f_pre(1, A(x), n, bw, p, A(tmp), A(msb), A(corr))
    nbytes = ol.mmo_cot_wire_bytes(n, 0, p, 1)
    wire, y = np.zeros(nbytes + 8, np.uint8), np.zeros(n, np.uint64)
    ol.mmo_cot_send(A(rcm), A(delta), A(S), gid0, A(corr), n, 0, p, 1, A(y), A(wire))
    ol.mmo_r2f_post(1, A(tmp), A(msb), A(y), n, bw, p, A(fld))
    d_words = torch.empty(n, dtype=torch.int64, device="cuda")
    L.check(lib.mmh_chacha20_words(key, sid, 0, L.dptr(d_words), n, None))
    rnd = host(d_words)
    ref_in, ref_sc = bn.encode(fld), bn.encode(scale_f)
    ref_r = bn.encode(rnd % np.uint64(p))
    ref_out, ref_share = bn.bn(cts, ref_in, ref_sc, rnd)
    k = 12
    S2, rcm2 = splitmix64(SEED + 88, 2), splitmix64(SEED + 89, 2 * n)
    gid1 = gid0 + 8 * ((n + 7) // 8)
    ref_ring, ref_wire2 = {}, {}
    for shift in (k, -1):   # Field2Ring_with_Truncate / Field2Ring on her share, as the sender of the second COT
        ol.mmo_f2r_pre(1, A(ref_share), n, p, max(shift, 0), A(tmp), A(msb), A(corr))
        nb2 = ol.mmo_cot_wire_bytes(n, bw, p, 0)
        w2, y2, o2 = np.zeros(nb2 + 8, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        ol.mmo_cot_send(A(rcm2), A(delta), A(S2), gid1, A(corr), n, bw, p, 0, A(y2), A(w2))
        if shift < 0:
            ol.mmo_f2r_post(1, A(tmp), A(msb), A(y2), n, p, bw, A(o2))
        else:
            ol.mmo_f2r_trunc_post(1, A(tmp), A(msb), A(y2), n, p, bw, shift, A(o2))
        ref_ring[shift], ref_wire2[shift] = o2, w2[:nb2]
    # the two calls
    d_rcm, d_rcm2, d_x, d_sc, d_cts = dev(rcm), dev(rcm2), dev(x), dev(scale_f), dev(cts)   # named: a temporary's memory is reused at once
    for shift, want_share in ((k, True), (-1, False)):
        d_pts3 = torch.full((3, bn.n_ct, N), -1, dtype=torch.int64, device="cuda")
        d_share = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        d_ring = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        d_wire = torch.full(((nbytes + 7) // 8 + 1,), -1, dtype=torch.int64, device="cuda")
        d_wire2 = torch.full(((nb2 + 7) // 8 + 1,), -1, dtype=torch.int64, device="cuda")
        L.check(lib.mmh_bn_alice_prepare(g.handle, L.dptr(d_rcm), L.dptr(d_rcm2), u2(delta), u2(S), u2(S2), gid0, L.dptr(d_x),
                                         n, bw, shift, L.dptr(d_sc), key, sid, L.dptr(d_pts3),
                                         L.dptr(d_share) if want_share else None, L.dptr(d_ring), L.dptr(d_wire),
                                         L.dptr(d_wire2), None))
        pts3 = host(d_pts3)
        assert np.array_equal(pts3[0], ref_in) and np.array_equal(pts3[1], ref_sc) and np.array_equal(pts3[2], ref_r)
        if want_share:
            assert np.array_equal(host(d_share), ref_share)
        assert np.array_equal(host(d_ring), ref_ring[shift]), f"shift {shift}"
        gw = d_wire.cpu().numpy().view(np.uint8)
        assert np.array_equal(gw[:nbytes], wire[:nbytes]) and (gw[nbytes:] == 0xFF).all()
        gw = d_wire2.cpu().numpy().view(np.uint8)
        assert np.array_equal(gw[:nb2], ref_wire2[shift]) and (gw[nb2:] == 0xFF).all()
    d_out = torch.full((bn.n_ct, 2, N), -1, dtype=torch.int64, device="cuda")
    L.check(lib.mmh_bn_alice_finish(g.handle, L.dptr(d_cts), L.dptr(d_pts3), n, L.dptr(d_out), None))
    assert np.array_equal(host(d_out), ref_out.reshape(bn.n_ct, 2, N))
    with pytest.raises(mm.MMHError):
        L.check(lib.mmh_bn_alice_finish(g.handle, None, L.dptr(d_pts3), n, L.dptr(d_out), None))
    for bad in (dict(bw=1), dict(shift=64), dict(ring=None), dict(n=0)):   # argument checks: nothing is launched
        with pytest.raises(mm.MMHError):
            L.check(lib.mmh_bn_alice_prepare(g.handle, L.dptr(d_rcm), L.dptr(d_rcm2), u2(delta), u2(S), u2(S2), gid0,
                                             L.dptr(d_x), bad.get("n", n), bad.get("bw", bw), bad.get("shift", k), L.dptr(d_sc),
                                             key, sid, L.dptr(d_pts3), None, bad.get("ring", L.dptr(d_ring)), L.dptr(d_wire),
                                             L.dptr(d_wire2), None))
    plain = mm.Context(N, (60, 49), 1 << 37)   # no batching under t = 2^37 (setUp_Field: "invalid SIMD modulus")
    with pytest.raises(mm.MMHError):
        L.check(lib.mmh_bn_alice_finish(plain.handle, L.dptr(d_cts), L.dptr(d_pts3), n, L.dptr(d_out), None))


def test_alice_bn_layer_in_two_calls_n8192(mm, monkeypatch):
    """the same on an N = 8192 context (the LOGN = 13 forms of bn_front_kernel and of the transforms; not a configuration
    the reference's bn runs, cheetah-api.cpp:150-162, but one the entry points accept)"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "N", 8192)
    test_alice_bn_layer_in_two_calls(mm, 2 * 8192 + 77)
    test_alice_bn_layer_in_two_calls(mm, 100)


def test_bole_full_size_identity(mm):
    """bole_test flags 1 1 size: 2^20 elements, l=32, |x|,|y| < 2^12 (tests/bole_test.cpp:78,98,568)"""
    import torch
    from openmm_amd.hom_bn_ss import HomBNSS, Meta
    from openmm_amd.truncation import bole_cross_term
    g = mm.Context(N, (60, 49), P32)
    o = OracleCtx(N, (60, 49), P32)
    sk, _ = keypair(o)
    n = 1 << 20
    n_ct = n // N
    meta = Meta((n,), P32, True)
    client, server = HomBNSS(), HomBNSS()
    client.setUp_Field(P32, g, sk_ntt=dev(sk))
    server.setUp_Field(P32, g)
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)
    x, y = rnd(-(1 << 12), 1 << 12, n), rnd(-(1 << 12), 1 << 12, n)
    x0, y0 = rnd(0, P32, n), rnd(0, P32, n)
    x1, y1 = (x - x0) % P32, (y - y0) % P32
    qs = torch.tensor(g.q, dtype=torch.int64, device="cuda").view(1, 2, 1)
    ex = client.encryptVector(x0, meta, (rnd(0, 1 << 62, n_ct, 2, N) % qs, rnd(-8, 9, n_ct, N)))
    ey = client.encryptVector(y0, meta, (rnd(0, 1 << 62, n_ct, 2, N) % qs, rnd(-8, 9, n_ct, N)))
    out_cts, share = server.bole(ex, ey, server.encodeVector(x1, meta), server.encodeVector(y1, meta), meta,
                                 rnd(0, 1 << 62, n))
    dec = client.decryptToVector(out_cts, meta)
    out_b = bole_cross_term(dec, x0, y0, P32)
    out_a = bole_cross_term(share, x1, y1, P32)
    assert torch.equal((out_a + out_b) % P32, (x * y) % P32)


@pytest.mark.parametrize("ishape,shared", [((3, 10, 12), True), ((2, 64, 64), False), ((2, 70, 64), True),
                                           ((1, 10, 70), True), ((5, 1, 1), True)])
def test_bn_direct_bit_exact(mm, ishape, shared):
    """HomBNSS direct path (hom_bn_ss.cc:471-790) against the oracle at every step; (1,10,70) has a
    tile that is ragged in w, where the reference's encoder and its dense decrypt indexing disagree --
    the bits must still match the restatement."""
    from openmm_amd.hom_bn_ss import HomBNSS, TensorMeta
    from tests.oracle_layers import BnDirectOracle
    T = 1 << 37
    o, g = OracleCtx(N, (60, 49), T), mm.Context(N, (60, 49), T)
    bn = BnDirectOracle(o, ishape, shared)
    meta = TensorMeta(ishape, T, shared)
    sk, pk = keypair(o)
    client, server = HomBNSS(), HomBNSS()
    client.setUp_Base(T, g, sk_ntt=dev(sk))
    server.setUp_Base(T, g, pk_ntt=dev(pk))
    x0 = splitmix64(SEED + 1, int(np.prod(ishape))).reshape(ishape)      # unreduced: encode takes mod t
    x1 = uniform_mod(SEED + 2, ishape, T)
    scales = ((splitmix64(SEED + 3, ishape[0]) % np.uint64(8191)).astype(np.int64) - 4095).view(np.uint64)
    a, e = rns_uniform(o, SEED + 10, bn.n_ct, 2), noise(SEED + 11, (bn.n_ct, N))
    ref_cts = bn.encrypt(sk, x0, a, e)
    got_cts = client.encryptTensor(dev(x0), meta, (dev(a), dev(e)))
    assert np.array_equal(host(got_cts), ref_cts)
    ref_pts = got_pts = None
    if shared:
        ref_pts, got_pts = bn.encode(x1), server.encodeTensor(dev(x1), meta)
        assert np.array_equal(host(got_pts), ref_pts)
    u, e0, e1 = ternary(SEED + 20, (bn.n_ct, N)), noise(SEED + 21, (bn.n_ct, N)), noise(SEED + 22, (bn.n_ct, N))
    share1 = uniform_mod(SEED + 30, ishape, T)
    ref_out = bn.server(ref_cts, ref_pts, scales, pk, u, e0, e1, share1)
    got_out, got_share = server.bn_direct(got_cts, got_pts, dev(scales), meta, (dev(u), dev(e0), dev(e1), dev(share1)))
    assert np.array_equal(host(got_out), ref_out)
    assert np.array_equal(host(got_cts), ref_cts)  # inputs are not destroyed
    got_dec = host(client.decryptToTensor(got_out, meta))
    assert np.array_equal(got_dec, bn.decrypt(np.ascontiguousarray(sk[:1]), ref_out))
    if ishape[2] <= 64 or ishape[2] % 64 == 0:      # no tile ragged in w: the protocol identity holds exactly
        s = scales.view(np.int64).astype(object)[:, None, None]
        xs = (x0.astype(object) + (x1.astype(object) if shared else 0)) * s
        assert np.array_equal((got_dec.astype(object) + host(got_share).astype(object)) % T, xs % T)


def test_bn_direct_errors(mm):
    import torch
    from openmm_amd.hom_bn_ss import HomBNSS, TensorMeta
    T = 1 << 37
    g = mm.Context(N, (60, 49), T)
    h = HomBNSS()
    with pytest.raises(mm.MMHError) as ei:   # setUp_Base: plain modulus mismatch -> ERR_INVALID_ARG
        h.setUp_Base(1 << 36, g)
    assert ei.value.code == 7
    h.setUp_Base(T, g)
    meta = TensorMeta((2, 4, 4), T, True)
    with pytest.raises(mm.MMHError) as ei:   # shape mismatch -> ERR_CONFIG (hom_bn_ss.cc:475)
        h.encodeTensor(torch.zeros((2, 4, 5), dtype=torch.int64, device="cuda"), meta)
    assert ei.value.code == 1
    with pytest.raises(mm.MMHError) as ei:   # no public key -> ERR_CONFIG
        h.bn_direct(None, None, None, meta, None)
    assert ei.value.code == 1
