"""Host-side mirror of ``gemini::HomFCSS`` (include/gemini/cheetah/hom_fc_ss.h:27-121)."""
import ctypes as C
from dataclasses import dataclass

from . import lib as L


class FcMetaC(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("is_shared_input", C.c_int32)]


@dataclass
class Meta:
    """HomFCSS::Meta: weight_shape (rows, cols), input_shape (cols,), is_shared_input"""
    weight_shape: tuple
    input_shape: tuple
    is_shared_input: bool = True


class Plan:
    def __init__(self, ctx, meta):
        if len(meta.weight_shape) != 2 or len(meta.input_shape) != 1 or meta.input_shape[0] != meta.weight_shape[1]:
            raise L.MMHError(7, "getSplit invalid meta")  # hom_fc_ss.cc:26-29 (LOG(FATAL) in the reference)
        self.ctx = ctx
        m = FcMetaC(meta.weight_shape[0], meta.weight_shape[1], int(meta.is_shared_input))
        self.handle = L._vp()
        L.check(ctx.lib.mmh_fc_plan_create(ctx.handle, C.byref(m), C.byref(self.handle)))
        split, a, b = (C.c_int64 * 2)(), L._sz(), L._sz()
        L.check(ctx.lib.mmh_fc_plan_info(self.handle, split, C.byref(a), C.byref(b)))
        self.split, self.n_in, self.n_out = (int(split[0]), int(split[1])), a.value, b.value

    def __del__(self):
        try:
            if self.handle:
                self.ctx.lib.mmh_fc_plan_destroy(self.handle)
                self.handle = L._vp()
        except Exception:
            pass


class HomFCSS:
    def __init__(self):
        self.ctx = self.sk_ntt = self.pk_ntt = None
        self._plans = {}

    def setUp(self, ctx, sk_ntt=None, pk_ntt=None):  # hom_fc_ss.cc:129-155
        self.ctx, self.sk_ntt, self.pk_ntt = ctx, sk_ntt, pk_ntt
        self.pk_last = None if pk_ntt is None else pk_ntt[:, :1].contiguous()
        self.sk_last = None if sk_ntt is None else sk_ntt[:1].contiguous()
        return 0

    def plan(self, meta):
        key = (tuple(meta.weight_shape), bool(meta.is_shared_input))
        if key not in self._plans:
            self._plans[key] = Plan(self.ctx, meta)
        return self._plans[key]

    @staticmethod
    def _need(cond, code, what):
        if not cond:
            raise L.MMHError(code, what)

    def encryptInputVector(self, vec, meta, rand):  # :214-273
        self._need(self.ctx is not None and self.sk_ntt is not None, 1, "encryptInputVector: no secret key")
        self._need(tuple(vec.shape) == tuple(meta.input_shape), 3, "encryptInputVector: input shape")
        p, (a, e) = self.plan(meta), rand
        out = self.ctx._empty(p.n_in, 2, self.ctx.L, self.ctx.N)
        L.check(self.ctx.lib.mmh_fc_encrypt_input(p.handle, L.dptr(self.sk_ntt), L.dptr(vec), L.dptr(a), L.dptr(e),
                                                  L.dptr(out), L._stream_ptr()))
        return out

    def encodeInputVector(self, vec, meta):  # :275-325
        self._need(tuple(vec.shape) == tuple(meta.input_shape), 3, "encodeInputVector: input shape")
        p = self.plan(meta)
        out = self.ctx._empty(p.n_in, self.ctx.N)
        L.check(self.ctx.lib.mmh_fc_encode_input(p.handle, L.dptr(vec), L.dptr(out), L._stream_ptr()))
        return out

    def encodeWeightMatrix(self, w, meta):  # :327-392 -> [n_out, n_in, N]
        self._need(tuple(w.shape) == tuple(meta.weight_shape), 3, "encodeWeightMatrix: weight shape")
        p = self.plan(meta)
        out = self.ctx._empty(p.n_out, p.n_in, self.ctx.N)
        L.check(self.ctx.lib.mmh_fc_encode_weights(p.handle, L.dptr(w), L.dptr(out), L._stream_ptr()))
        # :414-422 rejects a d0 x d1 block that encodes to zero; decide on the raw matrix (small)
        import torch.nn.functional as F
        from .hom_conv2d_ss import _mod_t
        d0, d1 = p.split
        nz = F.pad(_mod_t(w, self.ctx.t) != 0, (0, p.n_in * d1 - w.shape[1], 0, p.n_out * d0 - w.shape[0]))
        out._mm_nonzero = L.DeferredFlag(nz.view(p.n_out, d0, p.n_in, d1).any(dim=3).any(dim=1).all())
        return out

    # :394-471; rand = (u, e1, mask) or, exact=True (USE_APPROX_RESHARE=0 build), (flood, u, e0, e1, mask)
    def matVecMul(self, matrix, vec_share0, vec_share1, meta, rand, exact=False):
        self._need(self.pk_ntt is not None, 1, "matVecMul: no public key")
        p = self.plan(meta)
        self._need(tuple(vec_share0.shape) == (p.n_in, 2, self.ctx.L, self.ctx.N), 7, "matVecMul: #input ciphertexts")
        self._need(len(rand) == (5 if exact else 3), 7, "matVecMul: rand")
        for r in rand:  # every sampled polynomial batch is [n_out, N]
            self._need(r is not None and tuple(r.shape) == (p.n_out, self.ctx.N), 3, "matVecMul: rand shape")
        self._need(tuple(matrix.shape) == (p.n_out, p.n_in, self.ctx.N), 7, "matVecMul: weight plaintexts shape")
        self._need(tuple(matrix.shape[:2]) == (p.n_out, p.n_in), 7, "matVecMul: #weight blocks")
        # :414-422: a sub-matrix that encodes to zero is rejected
        nz = getattr(matrix, "_mm_nonzero", None)
        if nz is None:
            nz = L.DeferredFlag((matrix != 0).any(dim=2).all())
        if meta.is_shared_input:
            self._need(vec_share1 is not None and tuple(vec_share1.shape) == (p.n_in, self.ctx.N), 3, "matVecMul: #shares")
        out_cts = self.ctx._empty(p.n_out, 2, 1, self.ctx.N)
        out_share = self.ctx._empty(meta.weight_shape[0])
        if exact:
            flood, u, e0, e1, mask = rand
            L.check(self.ctx.lib.mmh_fc_server_exact(p.handle, L.dptr(vec_share0),
                                                     L.dptr(vec_share1 if meta.is_shared_input else None),
                                                     L.dptr(matrix), L.dptr(self.pk_last), L.dptr(flood), L.dptr(u),
                                                     L.dptr(e0), L.dptr(e1), L.dptr(mask), L.dptr(out_cts),
                                                     L.dptr(out_share), L._stream_ptr()))
        else:
            u, e1, mask = rand
            L.check(self.ctx.lib.mmh_fc_server(p.handle, L.dptr(vec_share0),
                                               L.dptr(vec_share1 if meta.is_shared_input else None), L.dptr(matrix),
                                               L.dptr(self.pk_last), L.dptr(u), L.dptr(e1), L.dptr(mask),
                                               L.dptr(out_cts), L.dptr(out_share), L._stream_ptr()))
        self._need(bool(nz), 7, "matVecMul: sub-matrix with all zero is not supported")
        return out_cts, out_share

    def decryptToVector(self, enc_vector, meta):  # :572-602
        self._need(self.sk_ntt is not None, 1, "decryptToVector: no secret key")
        p = self.plan(meta)
        self._need(enc_vector.shape[0] == p.n_out, 7, "decryptToVector: #ciphertexts")
        out = self.ctx._empty(meta.weight_shape[0])
        L.check(self.ctx.lib.mmh_fc_decrypt(p.handle, L.dptr(self.sk_last), L.dptr(enc_vector), L.dptr(out),
                                            L._stream_ptr()))
        return out
