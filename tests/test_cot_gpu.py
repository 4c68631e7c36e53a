"""GPU parity of the COT post-processing (mmh_cot_send / mmh_cot_recv, silent_ot.h:85-242) against the oracle bit
for bit -- data0, data and every byte of the wire -- over ring widths and moduli, ragged lengths and gid offsets;
the two-party correlation at the reference's test size; and ring_to_field / ReLU with this COT in place of the
ideal one.  What pins the oracle is in tests/test_cot_oracle.py."""
import ctypes as C

import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64
from tests.test_convert_gpu import PRIMES, _two_party, dev, host
from tests.test_cot_oracle import cot_pair

pytestmark = pytest.mark.gpu
SEED = 0x4D264D


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def u2(v):
    return (C.c_uint64 * 2)(int(v[0]), int(v[1]))


def gpu_pair(n, l, p, seed, gid0, corr=None):
    """run both kernels on the oracle's inputs; returns (oracle results, gpu results)"""
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib = load(), load_library()
    prime = 1 if p else 0
    data0, data, b, corr, wire, (rcm, rcm_r, delta, S) = cot_pair(o, n, l, p, seed, corr=corr, gid0=gid0)
    nb = C.c_size_t()
    check(lib.mmh_cot_wire_bytes(n, l, p, prime, C.byref(nb)))
    assert nb.value == len(wire)
    d_rcm, d_rcm_r, d_corr, d_b = dev(rcm), dev(rcm_r), dev(corr), dev(b)
    d_data0, d_data = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(2))
    d_wire = torch.full(((len(wire) + 7) // 8 + 1,), -1, dtype=torch.int64, device="cuda")
    check(lib.mmh_cot_send(dptr(d_rcm), u2(delta), u2(S), gid0, dptr(d_corr), n, l, p, prime, dptr(d_data0),
                           dptr(d_wire), None))
    g_wire = d_wire.cpu().numpy().view(np.uint8)
    assert (g_wire[len(wire):] == 0xFF).all(), "the sender wrote past the wire"
    # the receiver reads the ORACLE's wire: its parity does not lean on the sender's
    d_owire = dev(np.concatenate([wire, np.zeros(-len(wire) % 8, np.uint8)]).view(np.uint64)) if len(wire) else d_wire
    check(lib.mmh_cot_recv(dptr(d_rcm_r), dptr(d_b), u2(S), gid0, dptr(d_owire), n, l, p, prime, dptr(d_data), None))
    return (data0, wire, data), (host(d_data0), g_wire[:len(wire)], host(d_data))


def assert_same(ref, got):
    for name, r, g in zip(("data0", "wire", "data"), ref, got):
        assert np.array_equal(r, g), name


@pytest.mark.parametrize("l", [1, 2, 8, 12, 32, 37, 41, 63, 64])
def test_cot_ring_parity(mm, l):
    for n, gid0 in [(1, 0), (7, 5), (8, 8), (9, 1 << 33), (255, 3), (256, 0), (257, (1 << 64) - 100), (20001, 16)]:
        assert_same(*gpu_pair(n, l, 0, SEED + 10 * l + n, gid0))


@pytest.mark.parametrize("p", [2, 65537, PRIMES[32], PRIMES[37] if 37 in PRIMES else (1 << 37) - 25,
                               1152921504606830593, 1 << 37, (1 << 63) - 25, (1 << 63) + 29, (1 << 64) - 59])
def test_cot_prime_parity(mm, p):
    for n, gid0 in [(1, 0), (8, 8), (9, 1 << 40), (255, 3), (257, 1), (20003, 24)]:
        assert_same(*gpu_pair(n, 0, p, SEED + n + p % 997, gid0))


def test_cot_prime_extreme_operands(mm):
    """corr and wire values outside [0, p): the reference reduces them over 128 bits (silent_ot.h:156,:236)"""
    n, p = 4096, PRIMES[32]
    corr = np.full(n, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    corr[::3] = 0
    corr[1::3] = np.uint64(p - 1)
    assert_same(*gpu_pair(n, 0, p, SEED + 9, 0, corr=corr))
    assert_same(*gpu_pair(n, 0, (1 << 64) - 59, SEED + 9, 0, corr=corr))


def test_cot_arguments(mm):
    from openmm_amd.lib import load_library
    lib, nb = load_library(), C.c_size_t()
    assert lib.mmh_cot_wire_bytes(8, 0, 0, 0, C.byref(nb)) == 7        # MMH_ERR_INVALID_ARG: l out of range
    assert lib.mmh_cot_wire_bytes(8, 65, 0, 0, C.byref(nb)) == 7
    assert lib.mmh_cot_wire_bytes(8, 0, 1, 1, C.byref(nb)) == 7        # p < 2
    assert lib.mmh_cot_wire_bytes(8, 32, 0, 0, None) == 2              # MMH_ERR_NULL_POINTER
    z = u2([0, 0])
    assert lib.mmh_cot_send(None, z, z, 0, None, 0, 32, 0, 0, None, None, None) == 0      # empty call
    assert lib.mmh_cot_send(None, z, z, 0, None, 5, 32, 0, 0, None, None, None) == 2
    assert lib.mmh_cot_recv(None, None, z, 0, None, 5, 32, 0, 0, None, None) == 2


def test_silent_ot_two_parties_at_reference_size(mm):
    """N = 10^7 elements (tests/ring_to_field_test.cpp:171): receiver = data0 + b * corr on both forms, the gid
    running on across the calls"""
    import torch
    from openmm_amd.silent_ot import IdealFerret, LoopbackIO, SilentOT
    n, l, p = 10_000_000, 37, PRIMES[32]
    ferret, io = IdealFerret(SEED), LoopbackIO()
    snd, rcv = SilentOT(ferret, io, prg_seed=SEED), SilentOT(ferret, io.peer())
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    corr = torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=gen)
    b = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda", generator=gen)
    x, y = _two_party(lambda: snd.send_cot(corr, n, l), lambda: rcv.recv_cot(b, n, l))
    assert torch.equal(y, (x + b.to(torch.int64) * corr) & ((1 << l) - 1))
    assert io.sent_bytes == 16 + n // 8 * 8 * ((8 * l + 63) // 64)
    x, y = _two_party(lambda: snd.send_cot_prime(corr % p, n, p), lambda: rcv.recv_cot_prime(b, n, p))
    assert torch.equal(y, (x + b.to(torch.int64) * (corr % p)) % p)
    assert snd.gid == rcv.gid == 2 * n


def test_conversions_and_relu_over_device_cot(mm):
    """ring_to_field / field_to_ring and the ReLU flow with SilentOT (device post-processing over an ideal Ferret)
    in place of IdealCOT: same reconstructions as tests/ring_to_field_test.cpp:128-143 and relu-ring.h"""
    import torch
    from openmm_amd.nonlinear import AuxProtocols, ReLURingProtocol
    from openmm_amd.silent_ot import IdealFerret, LoopbackIO, SilentOT
    from openmm_amd.truncation import ALICE, BOB, IdealAux, Truncation
    n, bw, p = 100_003, 32, PRIMES[32]
    f1, io1, f2, io2 = IdealFerret(SEED), LoopbackIO(), IdealFerret(SEED + 1), LoopbackIO()
    straight = {ALICE: SilentOT(f1, io1, prg_seed=1), BOB: SilentOT(f1, io1.peer())}
    reverse = {BOB: SilentOT(f2, io2, prg_seed=2), ALICE: SilentOT(f2, io2.peer())}
    gen = torch.Generator(device="cuda").manual_seed(SEED)
    v = torch.randint(-(1 << (bw - 3)), 1 << (bw - 3), (n,), dtype=torch.int64, device="cuda", generator=gen)
    sa = torch.randint(0, 1 << bw, (n,), dtype=torch.int64, device="cuda", generator=gen)
    sb = (v - sa) & ((1 << bw) - 1)
    alice, bob = Truncation(ALICE, straight[ALICE]), Truncation(BOB, straight[BOB])
    fa, fb = _two_party(lambda: alice.ring_to_field(n, sa, bw, p), lambda: bob.ring_to_field(n, sb, bw, p))
    assert torch.equal((fa % p + fb % p) % p, v % p)
    ra, rb = _two_party(lambda: alice.field_to_ring(n, fa % p, p, bw), lambda: bob.field_to_ring(n, fb % p, p, bw))
    assert torch.equal((ra + rb) & ((1 << bw) - 1), v & ((1 << bw) - 1))
    ideal = IdealAux(SEED)
    relu = {q: ReLURingProtocol(q, bw, AuxProtocols(q, straight[q], reverse[q], ideal)) for q in (ALICE, BOB)}
    (ya, _), (yb, _) = _two_party(lambda: relu[ALICE].relu(n, sa), lambda: relu[BOB].relu(n, sb))
    assert torch.equal((ya + yb) & ((1 << bw) - 1), torch.clamp(v, min=0))


@pytest.mark.parametrize("l,p", [(37, 0), (0, PRIMES[32])])
def test_cot_shards_are_independent(mm, l, p):
    """Element ranges cut at multiples of 8 are independent units (how the path shards over GPUs): a shard run with
    gid0 + offset and its slice of the inputs writes exactly its slice of data0 and of the wire"""
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib = load(), load_library()
    n, cut, gid0, prime = 20011, 8 * 1203, 1 << 35, 1 if p else 0
    data0, _, _, corr, wire, (rcm, _, delta, S) = cot_pair(o, n, l, p, SEED + 17, gid0=gid0)
    got_data0, got_wire = [], []
    for a, b in ((0, cut), (cut, n)):
        m = b - a
        nb = C.c_size_t()
        check(lib.mmh_cot_wire_bytes(m, l, p, prime, C.byref(nb)))
        d_rcm, d_corr = dev(rcm[2 * a:2 * b].copy()), dev(corr[a:b].copy())
        d_data0 = torch.empty(m, dtype=torch.int64, device="cuda")
        d_wire = torch.zeros((nb.value + 7) // 8, dtype=torch.int64, device="cuda")
        check(lib.mmh_cot_send(dptr(d_rcm), u2(delta), u2(S), (gid0 + a) & 0xFFFFFFFFFFFFFFFF, dptr(d_corr), m, l, p,
                               prime, dptr(d_data0), dptr(d_wire), None))
        got_data0.append(host(d_data0))
        got_wire.append(d_wire.cpu().numpy().view(np.uint8)[:nb.value])
    assert np.array_equal(np.concatenate(got_data0), data0)
    assert np.array_equal(np.concatenate(got_wire), wire)


@pytest.mark.parametrize("n", [1, 9, 1003, 4096, 3 * 4096 + 77])
def test_sender_conversions_fused_with_the_cot(mm, n):
    """ALICE's Ring2Field -> encodeVector and Field2Ring[_with_Truncate] as the COT sender, each in one launch
    (mmh_cot_send_r2f_encode, mmh_cot_send_f2r): the outputs and every byte of the wire against the ORACLE's separate
    steps (pre step, sender's post-processing, post step, HomBNSS::encodeVector) on the same blocks."""
    import torch
    from openmm_amd.lib import check, dptr, load_library
    from tests.oracle_binding import OracleCtx
    from tests.oracle_layers import BnOracle
    o, lib = load(), load_library()
    N, bw, p, gid0 = 4096, 37, PRIMES[37], 8 * 1234
    delta = splitmix64(SEED + 70, 2) | np.uint64(1)
    S, rcm = splitmix64(SEED + 71, 2), splitmix64(SEED + 72, 2 * n)
    d_rcm = dev(rcm)

    def oracle_send(corr, l, prime):
        nbytes = o.mmo_cot_wire_bytes(n, l, p, prime)
        wire, data0 = np.zeros(nbytes + 8, np.uint8), np.zeros(n, np.uint64)
        o.mmo_cot_send(A(rcm), A(delta), A(S), gid0, A(corr), n, l, p, prime, A(data0), A(wire))
        return data0, wire[:nbytes]

    def gpu_wire(nbytes):
        return torch.full(((nbytes + 7) // 8 + 1,), -1, dtype=torch.int64, device="cuda")

    # ring -> field, then BatchEncoder's plaintexts
    x = splitmix64(SEED + 73, n) & np.uint64((1 << bw) - 1)
    x[:min(3, n)] = [0, (1 << bw) - 1, 1 << (bw - 1)][:min(3, n)]
    tmp, msb, corr, fld = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_r2f_pre(1, A(x), n, bw, p, A(tmp), A(msb), A(corr))
    y, wire = oracle_send(corr, 0, 1)
    o.mmo_r2f_post(1, A(tmp), A(msb), A(y), n, bw, p, A(fld))
    ref_pts = BnOracle(OracleCtx(N, (60, 49), p), n).encode(fld)
    g = mm.Context(N, (60, 49), p)
    d_pts = torch.full(ref_pts.shape, -1, dtype=torch.int64, device="cuda")
    d_wire = gpu_wire(len(wire))
    check(lib.mmh_cot_send_r2f_encode(g.handle, dptr(d_rcm), u2(delta), u2(S), gid0, dptr(dev(x)), n, bw, dptr(d_pts),
                                      dptr(d_wire), None))
    assert np.array_equal(host(d_pts), ref_pts)
    gw = d_wire.cpu().numpy().view(np.uint8)
    assert np.array_equal(gw[:len(wire)], wire) and (gw[len(wire):] == 0xFF).all()
    # field -> ring, plain and truncating
    share = splitmix64(SEED + 74, n) % np.uint64(p)
    share[:min(3, n)] = [0, p - 1, p // 2 + 1][:min(3, n)]
    for shift in (-1, 0, 12):
        o.mmo_f2r_pre(1, A(share), n, p, max(shift, 0), A(tmp), A(msb), A(corr))
        y, wire = oracle_send(corr, bw, 0)
        ref = np.zeros(n, np.uint64)
        if shift < 0:
            o.mmo_f2r_post(1, A(tmp), A(msb), A(y), n, p, bw, A(ref))
        else:
            o.mmo_f2r_trunc_post(1, A(tmp), A(msb), A(y), n, p, bw, shift, A(ref))
        d_out = torch.empty(n, dtype=torch.int64, device="cuda")
        d_wire = gpu_wire(len(wire))
        check(lib.mmh_cot_send_f2r(dptr(d_rcm), u2(delta), u2(S), gid0, dptr(dev(share)), n, p, bw, shift, dptr(d_out),
                                   dptr(d_wire), None))
        assert np.array_equal(host(d_out), ref), f"shift {shift}"
        gw = d_wire.cpu().numpy().view(np.uint8)
        assert np.array_equal(gw[:len(wire)], wire) and (gw[len(wire):] == 0xFF).all()
    with pytest.raises(mm.MMHError):
        check(lib.mmh_cot_send_f2r(dptr(d_rcm), u2(delta), u2(S), gid0, dptr(dev(share)), n, p, 0, 0, dptr(d_out),
                                   dptr(d_wire), None))
