"""Host-side mirror of SCI's ``Truncation`` modulus-conversion methods
(SCI/src/BuildingBlocks/truncation.h:29-210, truncation.cpp:319-695).

Same method names and argument meaning as the reference; the local share arithmetic runs
in libmm_hip's streaming kernels, the one COT per conversion stays on the host behind a
``cot`` object exposing the four calls the reference makes on ``otpack->iknp_straight``:

    send_cot(corr, n, l) -> y      recv_cot(choice, n, l) -> y          (mod 2^l)
    send_cot_prime(corr, n, p) -> y  recv_cot_prime(choice, n, p) -> y  (mod p)

``IdealCOT`` is the in-process test double in the style of SCI/src/OT/ideal.h:106-156
(it hands x to the sender and x + b*corr to the receiver); production callers plug in
their Ferret binding.
"""
import queue

import numpy as np

from . import lib as L

ALICE, BOB = 1, 2


class IdealCOT:
    """Loop-back ideal correlated OT for two Truncation objects living in two threads."""

    def __init__(self, seed=0x4D264D):
        self._q = queue.Queue()
        self._rng = np.random.default_rng(seed)

    def _send(self, corr, n, l, p):
        if p:
            x = self._rng.integers(0, p, size=n, dtype=np.uint64)
        else:
            x = self._rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + \
                self._rng.integers(0, 2, size=n, dtype=np.uint64)
            if l < 64:
                x &= np.uint64((1 << l) - 1)
        self._q.put((x, np.array(corr, dtype=np.uint64, copy=True)))
        return x

    def _recv(self, choice, n, l, p):
        x, corr = self._q.get()
        b = np.asarray(choice, dtype=np.uint64)
        if p:
            return (x + b * (corr % np.uint64(p))) % np.uint64(p)
        y = x + b * corr
        return y & np.uint64((1 << l) - 1) if l < 64 else y

    def send_cot(self, corr, n, l):
        return self._send(corr, n, l, 0)

    def recv_cot(self, choice, n, l):
        return self._recv(choice, n, l, 0)

    def send_cot_prime(self, corr, n, p):
        return self._send(corr, n, 0, p)

    def recv_cot_prime(self, choice, n, p):
        return self._recv(choice, n, 0, p)


class Truncation:
    """Truncation(party, cot): the five conversion methods of the reference class."""

    def __init__(self, party, cot, device=0):
        assert party in (ALICE, BOB)
        self.party, self.cot, self.device = party, cot, device
        self.lib = L.load_library()

    # -- helpers ---------------------------------------------------------------------
    def _bufs(self, dim):
        import torch
        dev = f"cuda:{self.device}"
        tmp = torch.empty(dim, dtype=torch.int64, device=dev)
        msb = torch.empty(dim, dtype=torch.uint8, device=dev)
        corr = torch.empty(dim, dtype=torch.int64, device=dev) if self.party == ALICE else None
        return tmp, msb, corr

    def _cot(self, msb, corr, dim, l=None, p=None):
        """the one interactive step (host): returns y as a device tensor"""
        if self.party == ALICE:
            c = L.to_host(corr)
            y = self.cot.send_cot_prime(c, dim, p) if p else self.cot.send_cot(c, dim, l)
        else:
            b = msb.cpu().numpy()
            y = self.cot.recv_cot_prime(b, dim, p) if p else self.cot.recv_cot(b, dim, l)
        return L.to_device(y, f"cuda:{self.device}")

    def _out(self, dim, width=1):
        import torch
        shape = (dim,) if width == 1 else (dim, width)
        return torch.empty(shape, dtype=torch.int64, device=f"cuda:{self.device}")

    # -- truncation.cpp:518-581 -------------------------------------------------------
    def ring_to_field(self, dim, inA, ring_bw, field_mod):
        tmp, msb, corr = self._bufs(dim)
        s = L._stream_ptr()
        L.check(self.lib.mmh_r2f_pre(self.party, L.dptr(inA), dim, ring_bw, field_mod, L.dptr(tmp),
                                     L.dptr(msb), L.dptr(corr), s))
        y = self._cot(msb, corr, dim, p=field_mod)
        out = self._out(dim)
        L.check(self.lib.mmh_r2f_post(self.party, L.dptr(tmp), L.dptr(msb), L.dptr(y), dim, ring_bw,
                                      field_mod, L.dptr(out), s))
        return out

    # -- truncation.cpp:583-637 -------------------------------------------------------
    def field_to_ring(self, dim, inA, field_mod, ring_bw):
        tmp, msb, corr = self._bufs(dim)
        s = L._stream_ptr()
        L.check(self.lib.mmh_f2r_pre(self.party, L.dptr(inA), dim, field_mod, 0, L.dptr(tmp), L.dptr(msb),
                                     L.dptr(corr), s))
        y = self._cot(msb, corr, dim, l=ring_bw)
        out = self._out(dim)
        L.check(self.lib.mmh_f2r_post(self.party, L.dptr(tmp), L.dptr(msb), L.dptr(y), dim, field_mod,
                                      ring_bw, L.dptr(out), s))
        return out

    # -- truncation.cpp:639-695 -------------------------------------------------------
    def field_to_ring_with_truncate(self, dim, inA, field_mod, ring_bw, shift_bw):
        tmp, msb, corr = self._bufs(dim)
        s = L._stream_ptr()
        L.check(self.lib.mmh_f2r_pre(self.party, L.dptr(inA), dim, field_mod, shift_bw, L.dptr(tmp),
                                     L.dptr(msb), L.dptr(corr), s))
        y = self._cot(msb, corr, dim, l=ring_bw)
        out = self._out(dim)
        L.check(self.lib.mmh_f2r_trunc_post(self.party, L.dptr(tmp), L.dptr(msb), L.dptr(y), dim, field_mod,
                                            ring_bw, shift_bw, L.dptr(out), s))
        return out

    # -- truncation.cpp:319-381 -------------------------------------------------------
    def ring_to_ring(self, dim, inA, bw_A, bw_B):
        if bw_A >= bw_B:  # :327-330: plain mask, no interaction
            mask = (1 << bw_B) - 1
            return inA & (mask if mask < (1 << 63) else -1)
        tmp, msb, corr = self._bufs(dim)
        s = L._stream_ptr()
        L.check(self.lib.mmh_r2r_pre(self.party, L.dptr(inA), dim, bw_A, L.dptr(tmp), L.dptr(msb),
                                     L.dptr(corr), s))
        y = self._cot(msb, corr, dim, l=bw_B - bw_A)
        out = self._out(dim)
        L.check(self.lib.mmh_r2r_post(self.party, L.dptr(tmp), L.dptr(msb), L.dptr(y), dim, bw_A, bw_B,
                                      L.dptr(out), s))
        return out

    # -- truncation.cpp:383-442: result as [dim, 2] = (low, high) 64-bit words ------------
    def ring_to_ring128(self, dim, inA, bw_A, bw_B):
        assert bw_B - bw_A <= 64  # :392
        tmp, msb, corr = self._bufs(dim)
        s = L._stream_ptr()
        L.check(self.lib.mmh_r2r_pre(self.party, L.dptr(inA), dim, bw_A, L.dptr(tmp), L.dptr(msb),
                                     L.dptr(corr), s))
        y = self._cot(msb, corr, dim, l=bw_B - bw_A)
        out = self._out(dim, 2)
        L.check(self.lib.mmh_r2r128_post(self.party, L.dptr(tmp), L.dptr(msb), L.dptr(y), dim, bw_A, bw_B,
                                         L.dptr(out), s))
        return out


def bole_cross_term(out, x, y, p):
    """out = (out + moduloMult(x, y, p)) % p  (cheetah-api.cpp:648-649), in place"""
    L.check(L.load_library().mmh_bole_cross_term(L.dptr(out), L.dptr(x), L.dptr(y), out.numel(), p,
                                                 L._stream_ptr()))
    return out
