"""CPU: the host-side doubles the SilentOT mirror runs over (openmm_amd/silent_ot.py) behave as the protocol needs:
the ideal Ferret hands the receiver the sender's block xor b * Delta, the loop-back channel keeps message order and
counts bytes; and the oracle's sender / receiver agree over blocks drawn that way."""
import numpy as np

from tests.oracle_binding import A, load


def test_ideal_ferret_correlation_and_loopback_framing():
    from openmm_amd.silent_ot import IdealFerret, LoopbackIO
    f = IdealFerret(7)
    assert int(f.Delta[0]) & 1 == 1                     # Ferret's Delta has its least significant bit set
    n = 1000
    b = (np.arange(n) % 3 == 0).astype(np.uint8)
    s = f.send_ot_rcm_cc(n)
    r = f.recv_ot_rcm_cc(b, n)
    assert s.shape == r.shape == (n, 2)
    assert np.array_equal(r[b == 0], s[b == 0]) and np.array_equal(r[b == 1], s[b == 1] ^ f.Delta)
    io = LoopbackIO()
    peer = io.peer()
    io.send_block(np.array([1, 2], dtype=np.uint64))
    io.send_data(b"abc")
    assert list(peer.recv_block()) == [1, 2] and peer.recv_data(3) == b"abc" and io.sent_bytes == 19
    peer.send_data(b"xy")
    assert io.recv_data(2) == b"xy"


def test_oracle_cot_over_ideal_ferret_blocks():
    from openmm_amd.silent_ot import IdealFerret
    o, f, n, l = load(), IdealFerret(11), 515, 41
    b = (np.arange(n) % 2).astype(np.uint8)
    s = np.ascontiguousarray(f.send_ot_rcm_cc(n)).reshape(-1)
    r = np.ascontiguousarray(f.recv_ot_rcm_cc(b, n)).reshape(-1)
    corr = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    S = np.array([3, 4], dtype=np.uint64)
    wire = np.zeros(o.mmo_cot_wire_bytes(n, l, 0, 0), np.uint8)
    data0, data = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    delta = np.ascontiguousarray(f.Delta)
    o.mmo_cot_send(A(s), A(delta), A(S), 0, A(corr), n, l, 0, 0, A(data0), A(wire))
    o.mmo_cot_recv(A(r), A(b), A(S), 0, A(wire), n, l, 0, 0, A(data))
    mask = np.uint64((1 << l) - 1)
    assert np.array_equal(data, (data0 + b.astype(np.uint64) * corr) & mask)
