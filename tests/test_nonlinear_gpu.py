"""GPU parity of the ReLU / multiplexer / B2A local share arithmetic (relu-ring.h:216-313,
aux-protocols.cpp:62-126) against the oracle bit for bit, and the two-party flows over ideal COTs: ReLU
reconstructs to max(x, 0) exactly, B2A to the bit, and the truncations still reconstruct with the COT-based B2A."""
import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64
from tests.test_convert_gpu import dev, edge_mix, host
from tests.test_truncate_gpu import _signed, _two_parties

pytestmark = pytest.mark.gpu
SEED = 0x4D264D


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("l", [32, 37, 41, 64, 2])
def test_relu_kernels(mm, party, l):
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib, n = load(), load_library(), 20000
    x = edge_mix(n, 1 << l, SEED + l)
    wrap = (splitmix64(SEED + 3, n) & np.uint64(1)).astype(np.uint8)
    msb, cmp, drelu, sel = np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    o.mmo_relu_pre(party, A(x), n, l, A(msb), A(cmp))
    o.mmo_relu_post(party, A(msb), A(wrap), n, A(drelu), A(sel))
    d_x, d_wrap = dev(x), dev(wrap)
    d_msb, d_drelu, d_sel = (torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3))
    d_cmp = torch.empty(n, dtype=torch.int64, device="cuda")
    check(lib.mmh_relu_pre(party, dptr(d_x), n, l, dptr(d_msb), dptr(d_cmp), None))
    check(lib.mmh_relu_post(party, dptr(d_msb), dptr(d_wrap), n, dptr(d_drelu), dptr(d_sel), None))
    assert np.array_equal(host(d_msb), msb) and np.array_equal(host(d_cmp), cmp)
    assert np.array_equal(host(d_drelu), drelu) and np.array_equal(host(d_sel), sel)


@pytest.mark.parametrize("party", [1, 2])
@pytest.mark.parametrize("bw", [1, 12, 32, 41, 64])
def test_mux_b2a_kernels(mm, party, bw):
    import torch
    from openmm_amd.lib import check, dptr, load_library
    o, lib, n = load(), load_library(), 20000
    x, r, s_ = edge_mix(n, 1 << bw, SEED + bw), splitmix64(SEED + 4, n), splitmix64(SEED + 5, n)
    sel = (splitmix64(SEED + 6, n) & np.uint64(1)).astype(np.uint8)
    corr, y, bc, bo = (np.zeros(n, np.uint64) for _ in range(4))
    o.mmo_mux_pre(A(sel), A(x), n, bw, A(corr))
    o.mmo_mux_post(A(sel), A(x), A(r), A(s_), n, bw, A(y))
    o.mmo_b2a_pre(A(sel), n, bw, A(bc))
    o.mmo_b2a_post(party, A(sel), A(r), n, bw, A(bo))
    d_sel, d_x, d_r, d_s = dev(sel), dev(x), dev(r), dev(s_)
    g = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(4)]
    check(lib.mmh_mux_pre(dptr(d_sel), dptr(d_x), n, bw, dptr(g[0]), None))
    check(lib.mmh_mux_post(dptr(d_sel), dptr(d_x), dptr(d_r), dptr(d_s), n, bw, dptr(g[1]), None))
    check(lib.mmh_b2a_pre(dptr(d_sel), n, bw, dptr(g[2]), None))
    check(lib.mmh_b2a_post(party, dptr(d_sel), dptr(d_r), n, bw, dptr(g[3]), None))
    for got, want in zip(g, (corr, y, bc, bo)):
        assert np.array_equal(host(got), want)


def _parties(seed):
    """per party: AuxProtocols over two ideal COTs (straight, reversed) and ideal compare / wrap"""
    from openmm_amd.nonlinear import AuxProtocols
    from openmm_amd.truncation import IdealAux, IdealCOT
    straight, rev, ideal = IdealCOT(seed), IdealCOT(seed + 1), IdealAux(seed + 2)
    return lambda party: AuxProtocols(party, straight, rev, ideal)


@pytest.mark.parametrize("l", [32, 41, 64])
def test_relu_reconstructs(mm, l):
    from openmm_amd.nonlinear import ReLURingProtocol
    n, mask = 5000, (1 << l) - 1
    rng = np.random.default_rng(SEED + l)
    secret = [int(v) - (1 << (l - 2)) for v in rng.integers(0, 1 << (l - 1), size=n, dtype=np.uint64)]
    secret[:4] = [0, 1, -1, (1 << (l - 1)) - 1]
    sa = [int(v) & mask for v in splitmix64(SEED + 1, n)]
    shares = {1: np.array(sa, dtype=np.uint64), 2: np.array([(s - a) & mask for s, a in zip(secret, sa)], dtype=np.uint64)}
    make = _parties(SEED)

    def run(party):
        res, drelu = ReLURingProtocol(party, l, make(party)).relu(n, dev(shares[party]))
        return host(res), host(drelu)
    (ra, da), (rb, db) = _two_parties(run)
    for s, a, b, p, q in zip(secret, ra, rb, da, db):
        assert (int(p) ^ int(q)) == (1 if s < 0 else 0)           # drelu_res holds shares of msb(x) (relu-ring.h:244-252)
        assert _signed(int(a) + int(b), l) == max(s, 0)           # ReLU, exactly


@pytest.mark.parametrize("bw", [1, 12, 64])
def test_b2a_reconstructs(mm, bw):
    n = 5000
    bit = (splitmix64(SEED + 9, n) & np.uint64(1)).astype(np.uint8)
    xa = (splitmix64(SEED + 10, n) & np.uint64(1)).astype(np.uint8)
    bits = {1: xa, 2: bit ^ xa}
    make = _parties(SEED + 20)
    oa, ob = _two_parties(lambda party: make(party).B2A(party, bits[party], n, bw))
    mask = (1 << bw) - 1
    for t, a, b in zip(bit, oa, ob):
        got = (int(a) ^ int(b)) if bw == 1 else (int(a) + int(b)) & mask
        assert got == int(t)


def test_truncate_with_cot_b2a(mm):
    """the truncation flow with B2A over the COT (only compare / wrap ideal): still floor(x / 2^s) within one unit"""
    from openmm_amd.truncation import Truncation
    n, bw, shift = 5000, 37, 12
    mask = (1 << bw) - 1
    rng = np.random.default_rng(SEED)
    secret = [int(v) - (1 << (bw - 5)) for v in rng.integers(0, 1 << (bw - 4), size=n, dtype=np.uint64)]
    sa = [int(v) & mask for v in splitmix64(SEED + 1, n)]
    shares = {1: np.array(sa, dtype=np.uint64), 2: np.array([(s - a) & mask for s, a in zip(secret, sa)], dtype=np.uint64)}
    make = _parties(SEED + 40)
    oa, ob = _two_parties(lambda party: host(Truncation(party, None, aux=make(party)).truncate(
        n, dev(shares[party]), shift, bw, True, None, True)))
    for s, a, b in zip(secret, oa, ob):
        assert _signed(int(a) + int(b), bw) in (s >> shift, (s >> shift) - 1)
