"""Host-side mirrors of the OT-based non-linear pieces whose local share arithmetic runs in libmm_hip
(SURVEY.md 8(f) next-4): ``AuxProtocols::{multiplexer, B2A}`` (SCI/src/BuildingBlocks/aux-protocols.cpp:62-126) and
``ReLURingProtocol::relu`` of the shipped USE_CHEETAH build (SCI/src/NonLinear/relu-ring.h:140-320).

The interactive steps stay on the host: the correlated OTs go through two ``cot`` objects with the four calls of
``openmm_amd.truncation`` (``cot`` = otpack->iknp_straight, ALICE sends; ``cot_rev`` = otpack->iknp_reversed, BOB
sends), the millionaire comparison and the wrap computations through an ``ideal`` object (``IdealAux``) until a
real binding is plugged in.
"""
import numpy as np

from . import lib as L
from .truncation import ALICE, BOB


class AuxProtocols:
    """AuxProtocols(party, cot, cot_rev, ideal): multiplexer and B2A with their arithmetic on the device; compare
    and the wrap family are forwarded to ``ideal`` (same call signatures as ``IdealAux``), so an object of this
    class can serve as the ``aux`` of ``Truncation``."""

    def __init__(self, party, cot, cot_rev, ideal=None, device=0):
        assert party in (ALICE, BOB)
        self.party, self.cot, self.cot_rev, self.ideal, self.device = party, cot, cot_rev, ideal, device
        self.lib = L.load_library()

    def _dev(self, a):
        import torch
        if isinstance(a, torch.Tensor):   # an on-device COT (silent_ot.SilentOT) already returns device tensors
            return a
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(f"cuda:{self.device}")

    def _empty(self, n):
        import torch
        return torch.empty(n, dtype=torch.int64, device=f"cuda:{self.device}")

    # aux-protocols.cpp:62-93.  sel: uint8 device tensor, x: int64 device tensor -> y device tensor
    def multiplexer(self, sel, x, size, bw_x, bw_y):
        assert bw_x <= 64 and bw_y <= bw_x
        s = L._stream_ptr()
        corr = self._empty(size)
        L.check(self.lib.mmh_mux_pre(L.dptr(sel), L.dptr(x), size, bw_y, L.dptr(corr), s))
        if getattr(self.cot, "on_device", False) and getattr(self.cot_rev, "on_device", False):
            c, b = corr, sel          # silent_ot.SilentOT takes device tensors: nothing crosses PCIe but the wire
        else:
            c, b = L.to_host(corr), sel.cpu().numpy()
        if self.party == ALICE:   # :79-81
            data_s = self.cot.send_cot(c, size, bw_y)
            data_r = self.cot_rev.recv_cot(b, size, bw_y)
        else:                     # :82-85
            data_r = self.cot.recv_cot(b, size, bw_y)
            data_s = self.cot_rev.send_cot(c, size, bw_y)
        y, d_r, d_s = self._empty(size), self._dev(data_r), self._dev(data_s)
        L.check(self.lib.mmh_mux_post(L.dptr(sel), L.dptr(x), L.dptr(d_r), L.dptr(d_s), size, bw_y, L.dptr(y), s))
        return y

    # aux-protocols.cpp:95-126.  Called by Truncation with host bits; returns host shares like IdealAux.B2A
    def B2A(self, party, bits, dim, bw):
        assert party == self.party
        s = L._stream_ptr()
        x = self._dev(np.asarray(bits, dtype=np.uint8))
        out = self._empty(dim)
        if bw == 1:
            L.check(self.lib.mmh_b2a_post(party, L.dptr(x), None, dim, bw, L.dptr(out), s))
            return L.to_host(out)
        if party == ALICE:
            corr = self._empty(dim)
            L.check(self.lib.mmh_b2a_pre(L.dptr(x), dim, bw, L.dptr(corr), s))
            y = self.cot.send_cot(corr if getattr(self.cot, "on_device", False) else L.to_host(corr), dim, bw)
        else:
            y = self.cot.recv_cot(x if getattr(self.cot, "on_device", False) else np.asarray(bits, dtype=np.uint8),
                                  dim, bw)
        d_y = self._dev(y)
        L.check(self.lib.mmh_b2a_post(party, L.dptr(x), L.dptr(d_y), dim, bw, L.dptr(out), s))
        return L.to_host(out)

    def compare(self, party, data, dim, bw):
        return self.ideal.compare(party, data, dim, bw)

    def wrap(self, party, x, dim, bw):
        return self.ideal.wrap(party, x, dim, bw)


class ReLURingProtocol:
    """ReLURingProtocol(party, l, aux): relu(num_relu, share) -> (result, drelu) as the USE_CHEETAH build computes
    it (relu-ring.h:216-252, :308-313): compare on the low l-1 bits, msb(x) = msb(x0) ^ msb(x1) ^ carry, multiplexer
    on its complement.  ``drelu`` is what the reference writes to ``drelu_res``: shares of msb(x) (:244-252)."""

    def __init__(self, party, l, aux):
        assert party in (ALICE, BOB) and 2 <= l <= 64
        self.party, self.l, self.aux = party, l, aux
        self.lib = L.load_library()

    def relu(self, num_relu, share, skip_ot=False):
        import torch
        s = L._stream_ptr()
        dev = share.device
        msb = torch.empty(num_relu, dtype=torch.uint8, device=dev)
        cmp = torch.empty(num_relu, dtype=torch.int64, device=dev)
        L.check(self.lib.mmh_relu_pre(self.party, L.dptr(share), num_relu, self.l, L.dptr(msb), L.dptr(cmp), s))
        wrap = self.aux.compare(self.party, L.to_host(cmp), num_relu, self.l - 1)   # :243
        d_wrap = torch.from_numpy(np.ascontiguousarray(wrap, dtype=np.uint8)).to(dev)
        drelu, sel = torch.empty_like(msb), torch.empty_like(msb)
        L.check(self.lib.mmh_relu_post(self.party, L.dptr(msb), L.dptr(d_wrap), num_relu, L.dptr(drelu),
                                       L.dptr(sel), s))
        if skip_ot:   # :254-259
            return None, drelu
        return self.aux.multiplexer(sel, share, num_relu, self.l, self.l), drelu   # :312
