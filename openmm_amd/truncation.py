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


class IdealAux:
    """Loop-back ideal versions of the OT-based sub-protocols Truncation::truncate* call on ``aux`` / ``mill``
    (SCI/src/BuildingBlocks/aux-protocols.h, Millionaire): each returns this party's share of the bit or value
    the real protocol computes.  Two parties in two threads share one object; they must make the same calls in
    the same order (as the real protocol requires).  Test double only, like ``IdealCOT``."""

    def __init__(self, seed=0x4D264D):
        self._to = {ALICE: queue.Queue(), BOB: queue.Queue()}
        self._rng = {ALICE: np.random.default_rng(seed), BOB: np.random.default_rng(seed)}   # run in lock step

    def _swap(self, party, mine):
        other = BOB if party == ALICE else ALICE
        self._to[other].put(np.array(mine, copy=True))
        return self._to[party].get()

    def _bit_shares(self, party, bit):
        r = self._rng[party].integers(0, 2, size=len(bit), dtype=np.uint8)
        return r if party == ALICE else (bit.astype(np.uint8) ^ r)

    @staticmethod
    def _wraps(a, b, bw):   # 1{a + b >= 2^bw} for values below 2^bw
        mask = (1 << bw) - 1
        return np.array([(int(x) & mask) + (int(y) & mask) > mask for x, y in zip(a, b)], dtype=np.uint8)

    def compare(self, party, data, dim, bw):          # mill->compare: shares of 1{ALICE's > BOB's}
        other = self._swap(party, data)
        a, b = (data, other) if party == ALICE else (other, data)
        return self._bit_shares(party, (a > b).astype(np.uint8))

    def wrap(self, party, x, dim, bw):                # MSB_to_Wrap / msb0_to_wrap / msb1_to_wrap / wrap_computation
        other = self._swap(party, x)
        return self._bit_shares(party, self._wraps(x, other, bw))

    def B2A(self, party, bits, dim, bw):              # boolean shares -> arithmetic shares mod 2^bw
        other = self._swap(party, bits)
        bit = (np.asarray(bits, dtype=np.uint8) ^ np.asarray(other, dtype=np.uint8)).astype(np.uint64)
        r = self._rng[party].integers(0, 1 << 62, size=dim, dtype=np.uint64) * np.uint64(4) + \
            self._rng[party].integers(0, 4, size=dim, dtype=np.uint64)
        mask = np.uint64((1 << bw) - 1) if bw < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
        return (r & mask) if party == ALICE else ((bit - r) & mask)


class Truncation:
    """Truncation(party, cot[, aux]): the five conversion methods of the reference class and, given ``aux``
    (the wrap / compare / B2A sub-protocols), its truncate family."""

    def __init__(self, party, cot, device=0, aux=None):
        assert party in (ALICE, BOB)
        self.party, self.cot, self.device, self.aux = party, cot, device, aux
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
        if getattr(self.cot, "on_device", False):   # silent_ot.SilentOT: operands and result stay on the device
            if self.party == ALICE:
                return self.cot.send_cot_prime(corr, dim, p) if p else self.cot.send_cot(corr, dim, l)
            return self.cot.recv_cot_prime(msb, dim, p) if p else self.cot.recv_cot(msb, dim, l)
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


    # -- truncation-cheetah.cpp (the shipped USE_CHEETAH build) ---------------------------------------
    def _truncate(self, dim, inA, shift, bw, signed_arithmetic, heuristic, wrap_of):
        """pre kernel, the wrap sub-protocol chosen by the caller, B2A, post kernel"""
        if shift == 0:
            return inA.clone()
        assert bw - shift > 0 and (not signed_arithmetic or bw - shift - 1 >= 0)   # :12-13
        s = L._stream_ptr()
        adj, cmp = self._out(dim), self._out(dim)
        L.check(self.lib.mmh_trunc_pre(self.party, L.dptr(inA), dim, bw, int(signed_arithmetic), int(heuristic),
                                       L.dptr(adj), L.dptr(cmp), s))
        wrap = wrap_of(L.to_host(adj), L.to_host(cmp))
        aw = L.to_device(self.aux.B2A(self.party, wrap, dim, shift), f"cuda:{self.device}")
        out = self._out(dim)
        L.check(self.lib.mmh_trunc_post(self.party, L.dptr(adj), L.dptr(aw), dim, shift, bw,
                                        int(signed_arithmetic), int(heuristic), L.dptr(out), s))
        return out

    # :2-82
    def truncate(self, dim, inA, shift, bw, signed_arithmetic=True, msb_x=None, apply_msb0_heuristic=True):
        if msb_x is not None:
            return self.truncate_msb(dim, inA, shift, bw, signed_arithmetic, msb_x)
        if apply_msb0_heuristic:   # :20-39: BOB adds 2^(bw-3) first and takes 2^(bw-3-shift) off afterwards
            return self._truncate(dim, inA, shift, bw, signed_arithmetic, True,
                                  lambda adj, cmp: self.aux.wrap(self.party, adj, dim, bw))
        return self._truncate(dim, inA, shift, bw, signed_arithmetic, False,
                              lambda adj, cmp: self.aux.compare(self.party, cmp, dim, bw))

    # :84-151 (the sub-protocol gets msb_x, inverted on ALICE when signed, :118-127; the ideal one ignores it)
    def truncate_msb(self, dim, inA, shift, bw, signed_arithmetic, msb_x):
        return self._truncate(dim, inA, shift, bw, signed_arithmetic, False,
                              lambda adj, cmp: self.aux.wrap(self.party, adj, dim, bw))

    # :155-218
    def truncate_msb0(self, dim, inA, shift, bw, signed_arithmetic=True):
        return self._truncate(dim, inA, shift, bw, signed_arithmetic, False,
                              lambda adj, cmp: self.aux.wrap(self.party, adj, dim, bw))

    # -- truncation.cpp:289-317 --------------------------------------------------------------------------
    def truncate_and_reduce(self, dim, inA, shift, bw):
        if shift == 0:
            return inA.clone()
        assert bw - shift > 0
        s = L._stream_ptr()
        lower = self._out(dim)
        L.check(self.lib.mmh_trunc_reduce_pre(L.dptr(inA), dim, shift, L.dptr(lower), s))
        wrap = self.aux.wrap(self.party, L.to_host(lower), dim, shift)
        aw = L.to_device(self.aux.B2A(self.party, wrap, dim, bw - shift), f"cuda:{self.device}")
        out = self._out(dim)
        L.check(self.lib.mmh_trunc_reduce_post(L.dptr(inA), L.dptr(aw), dim, shift, bw, L.dptr(out), s))
        return out


def bole_cross_term(out, x, y, p):
    """out = (out + moduloMult(x, y, p)) % p  (cheetah-api.cpp:648-649), in place"""
    L.check(L.load_library().mmh_bole_cross_term(L.dptr(out), L.dptr(x), L.dptr(y), out.numel(), p,
                                                 L._stream_ptr()))
    return out
