"""Host-side mirror of ``SilentOT``'s correlated-OT calls (SCI/src/OT/ferret/silent_ot.h:85-242) with the
post-processing on the device: between the blocks Ferret hands over and the values the conversions consume,
everything -- MITCCRH hash, mask or mod p, correction, wire packing -- is one libmm_hip launch per side
(``mmh_cot_send`` / ``mmh_cot_recv``).  What stays on the host is what the reference keeps in other classes:

    ferret  -- the random correlated OT itself: ``Delta`` (sender), ``send_ot_rcm_cc(n)`` and
               ``recv_ot_rcm_cc(b, n)`` returning n 16-byte blocks as an (n, 2) uint64 array
               (silent_ot.h:89,:131,:173,:211; Ferret's LPN expansion is out of scope)
    io      -- ``send_block / recv_block`` (one 16-byte block) and ``send_data / recv_data`` (bytes)

Same method names and argument meaning as the reference; ``send_cot`` .. ``recv_cot_prime`` are the names
``sci::OT<T>`` dispatches to them (SCI/src/OT/ot.h:80-93), so an object of this class is a ``cot`` for
``Truncation`` / ``AuxProtocols``.  ``IdealFerret`` and ``LoopbackIO`` are in-process test doubles.
"""
import os
import ctypes as C
import queue

import numpy as np

from . import lib as L


class IdealFerret:
    """Ideal random correlated OT for two parties in one process: the receiver's block is the sender's block
    xor b * Delta.  Test double only (pattern SCI/src/OT/ideal.h)."""

    def __init__(self, seed=0x4D264D):
        rng = np.random.default_rng(seed)
        self.Delta = rng.integers(0, 1 << 63, size=2, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        self._rng = rng
        self._q = queue.Queue()

    def send_ot_rcm_cc(self, n):
        blocks = self._rng.integers(0, 1 << 63, size=(n, 2), dtype=np.uint64) * np.uint64(2) + \
            self._rng.integers(0, 2, size=(n, 2), dtype=np.uint64)
        self._q.put(blocks.copy())
        return blocks

    def recv_ot_rcm_cc(self, b, n):
        blocks = self._q.get()
        assert len(blocks) == n
        return blocks ^ (np.asarray(b, dtype=np.uint64).reshape(n, 1) * self.Delta.reshape(1, 2))


class LoopbackIO:
    """Two queues standing in for the socket; ``peer()`` is the other end."""

    def __init__(self, rx=None, tx=None):
        self._rx = rx if rx is not None else queue.Queue()
        self._tx = tx if tx is not None else queue.Queue()
        self.sent_bytes = 0

    def peer(self):
        return LoopbackIO(self._tx, self._rx)

    def send_block(self, block):
        self.send_data(np.asarray(block, dtype=np.uint64).tobytes())

    def recv_block(self):
        return np.frombuffer(self.recv_data(16), dtype=np.uint64).copy()

    def send_data(self, data):
        data = bytes(data)
        self.sent_bytes += len(data)
        self._tx.put(data)

    def recv_data(self, nbytes):
        data = self._rx.get()
        assert len(data) == nbytes, (len(data), nbytes)
        return data


def _u64x2(v):
    return (C.c_uint64 * 2)(int(v[0]), int(v[1]))


class SilentOT:
    def __init__(self, ferret, io, device=0, prg_seed=None):
        self.ferret, self.io, self.device = ferret, io, device
        self.lib = L.load_library()
        # the MITCCRH start point S travels in the clear and must be unpredictable (silent_ot.h:91-95 draws it
        # from emp's AES-based PRG): the OS generator per call; a seeded stream only when a test asks for one
        self._prg = None if prg_seed is None else np.random.default_rng(prg_seed)
        self.gid = 0          # MITCCRH's key counter: runs on across calls, setS does not reset it
        self.on_device = True  # results are CUDA tensors; corr / b may be CUDA tensors or numpy arrays

    # -- helpers -----------------------------------------------------------------------
    def _dev(self, x, dtype):
        import torch
        if isinstance(x, torch.Tensor):
            return x.contiguous()
        a = np.ascontiguousarray(x, dtype=dtype)
        if dtype == np.uint64:
            a = a.view(np.int64)
        return torch.from_numpy(a).to(f"cuda:{self.device}")

    def wire_bytes(self, n, l=0, p=0):
        out = C.c_size_t()
        L.check(self.lib.mmh_cot_wire_bytes(n, l, p, 1 if p else 0, C.byref(out)))
        return out.value

    def _advance(self, n):
        g = C.c_uint64()
        L.check(self.lib.mmh_cot_gids(n, C.byref(g)))
        self.gid = (self.gid + g.value) & 0xFFFFFFFFFFFFFFFF

    def _send(self, corr, length, l, p):
        import torch
        rcm = self._dev(self.ferret.send_ot_rcm_cc(length), np.uint64)            # silent_ot.h:89 / :131
        if self._prg is None:
            seed = np.frombuffer(os.urandom(16), dtype=np.uint64).copy()
        else:
            seed = self._prg.integers(0, 1 << 63, size=2, dtype=np.uint64) * np.uint64(2) + \
                self._prg.integers(0, 2, size=2, dtype=np.uint64)
        self.io.send_block(seed)                                                  # :91-95
        dev = f"cuda:{self.device}"
        corr_d = self._dev(corr, np.uint64)
        data0 = torch.empty(length, dtype=torch.int64, device=dev)
        nbytes = self.wire_bytes(length, l, p)
        wire = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
        L.check(self.lib.mmh_cot_send(L.dptr(rcm), _u64x2(self.ferret.Delta), _u64x2(seed), self.gid, L.dptr(corr_d),
                                      length, l, p, 1 if p else 0, L.dptr(data0), L.dptr(wire), L._stream_ptr()))
        self._advance(length)
        self.io.send_data(wire.cpu().numpy().tobytes()[:nbytes])                  # :123 / :159, one message
        return data0

    def _recv(self, b, length, l, p):
        import torch
        b_host = b.cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
        rcm = self._dev(self.ferret.recv_ot_rcm_cc(b_host.astype(np.uint8), length), np.uint64)   # :173 / :211
        seed = self.io.recv_block()                                               # :175-176
        nbytes = self.wire_bytes(length, l, p)
        raw = np.frombuffer(self.io.recv_data(nbytes) + bytes(-nbytes % 8), dtype=np.int64)
        dev = f"cuda:{self.device}"
        wire = torch.from_numpy(raw.copy()).to(dev)
        data = torch.empty(length, dtype=torch.int64, device=dev)
        choice = self._dev(b, np.uint8)
        L.check(self.lib.mmh_cot_recv(L.dptr(rcm), L.dptr(choice), _u64x2(seed), self.gid,
                                      L.dptr(wire), length, l, p, 1 if p else 0, L.dptr(data), L._stream_ptr()))
        self._advance(length)
        return data

    # -- silent_ot.h:85-242 ------------------------------------------------------------
    def send_ot_cam_cc(self, corr, length, l):
        return self._send(corr, length, l, 0)

    def send_ot_cam_cc_prime(self, corr, length, p):
        return self._send(corr, length, 0, p)

    def recv_ot_cam_cc(self, b, length, l):
        return self._recv(b, length, l, 0)

    def recv_ot_cam_cc_prime(self, b, length, p):
        return self._recv(b, length, 0, p)

    # -- sci::OT<T> names (ot.h:80-93) ---------------------------------------------------
    send_cot, recv_cot = send_ot_cam_cc, recv_ot_cam_cc
    send_cot_prime, recv_cot_prime = send_ot_cam_cc_prime, recv_ot_cam_cc_prime
