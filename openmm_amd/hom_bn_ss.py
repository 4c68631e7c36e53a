"""Host-side mirror of ``gemini::HomBNSS`` (include/gemini/cheetah/hom_bn_ss.h:25-198): the SIMD
path = BOLE (hom_bn_ss.cc:151-406) on a batching context and the direct, coefficient-packed path
(hom_bn_ss.cc:471-790) on a t = 2^k context."""
import ctypes as C
from dataclasses import dataclass

from . import lib as L


@dataclass
class Meta:
    """HomBNSS::Meta fields used by the SIMD path"""
    vec_shape: tuple           # (n,)
    target_base_mod: int       # must equal the context's plain modulus (hom_bn_ss.cc:156)
    is_shared_input: bool = True


class BnTensorMetaC(C.Structure):
    _fields_ = [("c", C.c_int64), ("h", C.c_int64), ("w", C.c_int64), ("is_shared_input", C.c_int32)]


@dataclass
class TensorMeta:
    """HomBNSS::Meta fields used by the direct path"""
    ishape: tuple              # (C, H, W)
    target_base_mod: int
    is_shared_input: bool = True


class HomBNSS:
    def __init__(self):
        self.ctx = self.sk_ntt = self.pk_ntt = None
        self.dctx = self.dsk_ntt = self.dpk_ntt = None

    # setUp_Base (hom_bn_ss.cc:69-105): the context of the direct path, plain modulus = base mod
    def setUp_Base(self, target_base_mod, ctx, sk_ntt=None, pk_ntt=None):
        if ctx.t != target_base_mod:
            raise L.MMHError(7, "HomBNSS: invalid target base mod this SEALContext")
        self.dctx, self.dsk_ntt, self.dpk_ntt = ctx, sk_ntt, pk_ntt
        self.dsk_last = None if sk_ntt is None else sk_ntt[:1].contiguous()
        return 0

    def _tiles(self, meta):
        self._need(self.dctx is not None, 1, "HomBNSS: direct context not set up")
        self._need(len(meta.ishape) == 3, 1, "HomBNSS: ishape")
        m = BnTensorMetaC(meta.ishape[0], meta.ishape[1], meta.ishape[2], int(meta.is_shared_input))
        n = L._sz()
        L.check(self.dctx.lib.mmh_bn_tensor_count(self.dctx.handle, C.byref(m), C.byref(n), None))
        return m, n.value

    def encodeTensor(self, in_tensor, meta):  # :471-520
        m, n = self._tiles(meta)
        self._need(tuple(in_tensor.shape) == tuple(meta.ishape), 1, "encodeTensor: shape")  # ERR_CONFIG :475
        out = self.dctx._empty(n, self.dctx.N)
        L.check(self.dctx.lib.mmh_bn_encode_tensor(self.dctx.handle, C.byref(m), L.dptr(in_tensor), L.dptr(out),
                                                   L._stream_ptr()))
        return out

    def encryptTensor(self, in_tensor, meta, rand):  # :522-576; rand = (a [n,L,N], e [n,N])
        m, n = self._tiles(meta)
        self._need(self.dsk_ntt is not None, 1, "encryptTensor: no secret key")
        self._need(tuple(in_tensor.shape) == tuple(meta.ishape), 1, "encryptTensor: shape")
        a, e = rand
        out = self.dctx._empty(n, 2, self.dctx.L, self.dctx.N)
        L.check(self.dctx.lib.mmh_bn_encrypt_tensor(self.dctx.handle, C.byref(m), L.dptr(self.dsk_ntt),
                                                    L.dptr(in_tensor), L.dptr(a), L.dptr(e), L.dptr(out),
                                                    L._stream_ptr()))
        return out

    # bn_direct (:578-738); rand = (u, e0, e1 [n,N] signed, share1 [C,H,W] uniform mod t).  Returns
    # (out_share0 ciphertexts, out_share1 = share1)
    def bn_direct(self, tensor_share0, tensor_share1, scales, meta, rand):
        m, n = self._tiles(meta)
        self._need(self.dpk_ntt is not None, 1, "bn_direct: no public key")
        self._need(scales.dim() == 1 and scales.shape[0] == meta.ishape[0], 1, "bn_direct: scales")  # :586-588
        self._need(tensor_share0.shape[0] == n, 3, "bn_direct: #ciphertexts")                       # :595
        if meta.is_shared_input:
            self._need(tensor_share1 is not None and tensor_share1.shape[0] == n, 3, "bn_direct: #shares")
        u, e0, e1, share1 = rand
        self._need(tuple(share1.shape) == tuple(meta.ishape), 1, "bn_direct: share shape")
        out = self.dctx._empty(n, 2, 1, self.dctx.N)
        L.check(self.dctx.lib.mmh_bn_direct_server(self.dctx.handle, C.byref(m), L.dptr(tensor_share0),
                                                   L.dptr(tensor_share1 if meta.is_shared_input else None),
                                                   L.dptr(scales), L.dptr(self.dpk_ntt), L.dptr(u), L.dptr(e0),
                                                   L.dptr(e1), L.dptr(share1), L.dptr(out), L._stream_ptr()))
        return out, share1

    def decryptToTensor(self, cts, meta):  # :740-790
        m, n = self._tiles(meta)
        self._need(self.dsk_ntt is not None, 1, "decryptToTensor: no secret key")
        self._need(cts.shape[0] == n, 3, "decryptToTensor: #ciphertexts")
        out = self.dctx._empty(*meta.ishape)
        L.check(self.dctx.lib.mmh_bn_decrypt_tensor(self.dctx.handle, C.byref(m), L.dptr(self.dsk_last), L.dptr(cts),
                                                    L.dptr(out), L._stream_ptr()))
        return out

    # setUp_Field (hom_bn_ss.cc:107-149)
    def setUp_Field(self, target_field_mod, ctx, sk_ntt=None, pk_ntt=None):
        if ctx.t != target_field_mod:
            raise L.MMHError(7, "HomBNSS: invalid target field mod this SEALContext")
        self.ctx, self.sk_ntt, self.pk_ntt = ctx, sk_ntt, pk_ntt
        self.sk_last = None if sk_ntt is None else sk_ntt[:1].contiguous()
        return 0

    def _need(self, cond, code, what):
        if not cond:
            raise L.MMHError(code, what)

    def _n_ct(self, meta):
        n = meta.vec_shape[0]
        sub = min(n, self.ctx.N)
        return n, (n + sub - 1) // sub

    def _check(self, meta):
        self._need(self.ctx is not None, 1, "HomBNSS: not set up")
        self._need(meta.target_base_mod == self.ctx.t, 6, "HomBNSS: target_base_mod")  # ERR_OUT_BOUND

    def encryptVector(self, vec, meta, rand):  # :151-189
        self._check(meta)
        self._need(self.sk_ntt is not None, 1, "encryptVector: no secret key")
        n, n_ct = self._n_ct(meta)
        a, e = rand
        out = self.ctx._empty(n_ct, 2, self.ctx.L, self.ctx.N)
        L.check(self.ctx.lib.mmh_bn_encrypt_vector(self.ctx.handle, L.dptr(self.sk_ntt), L.dptr(vec), n, L.dptr(a),
                                                   L.dptr(e), L.dptr(out), L._stream_ptr()))
        return out

    def encodeVector(self, vec, meta):  # :203-233
        self._check(meta)
        n, n_ct = self._n_ct(meta)
        out = self.ctx._empty(n_ct, self.ctx.N)
        L.check(self.ctx.lib.mmh_bn_encode_vector(self.ctx.handle, L.dptr(vec), n, L.dptr(out), L._stream_ptr()))
        return out

    encodeScales = encodeVector  # :235-239

    # Conversions fused with the vector step next to them (include/mm_hip.h: mmh_bn_*_r2f / _f2r).  `conv` is what
    # ring_to_field holds after its COT -- (tmp [n], msb [n] uint8, y [n]) -- and ring_bw its bit width; the field
    # modulus is the context's t.  Same bits as ring_to_field's post step followed by encodeVector / encryptVector.
    def encodeVectorFromRing(self, party, conv, ring_bw, meta):
        self._check(meta)
        n, n_ct = self._n_ct(meta)
        tmp, msb, y = conv
        self._need(tmp.numel() == n and msb.numel() == n and y.numel() == n, 3, "encodeVector: conversion buffers")
        out = self.ctx._empty(n_ct, self.ctx.N)
        L.check(self.ctx.lib.mmh_bn_encode_vector_r2f(self.ctx.handle, party, L.dptr(tmp), L.dptr(msb), L.dptr(y), n,
                                                      ring_bw, L.dptr(out), L._stream_ptr()))
        return out

    def encryptVectorFromRing(self, party, conv, ring_bw, meta, rand):
        self._check(meta)
        self._need(self.sk_ntt is not None, 1, "encryptVector: no secret key")
        n, n_ct = self._n_ct(meta)
        tmp, msb, y = conv
        self._need(tmp.numel() == n and msb.numel() == n and y.numel() == n, 3, "encryptVector: conversion buffers")
        a, e = rand
        out = self.ctx._empty(n_ct, 2, self.ctx.L, self.ctx.N)
        L.check(self.ctx.lib.mmh_bn_encrypt_vector_r2f(self.ctx.handle, L.dptr(self.sk_ntt), party, L.dptr(tmp),
                                                       L.dptr(msb), L.dptr(y), n, ring_bw, L.dptr(a), L.dptr(e),
                                                       L.dptr(out), L._stream_ptr()))
        return out

    def decryptToVectorForRing(self, in_vec, meta, party, shift=0):
        """decryptToVector feeding field_to_ring's pre step (shift > 0: field_to_ring_with_truncate's): returns
        (tmp [n], msb [n] uint8, corr [n]) as mmh_f2r_pre would from the decrypted vector"""
        import torch
        self._check(meta)
        self._need(self.sk_ntt is not None, 1, "decryptToVector: no secret key")
        n, n_ct = self._n_ct(meta)
        self._need(in_vec.shape[0] == n_ct, 7, "decryptToVector: #ciphertexts")
        tmp, corr = self.ctx._empty(n), self.ctx._empty(n)
        msb = torch.empty(n, dtype=torch.uint8, device=tmp.device)
        L.check(self.ctx.lib.mmh_bn_decrypt_vector_f2r(self.ctx.handle, L.dptr(self.sk_last), L.dptr(in_vec), n, party,
                                                       shift, L.dptr(tmp), L.dptr(msb), L.dptr(corr), L._stream_ptr()))
        return tmp, msb, corr

    def bn(self, vec_share0, vec_share1, scales, meta, random):  # :241-278
        self._check(meta)
        n, n_ct = self._n_ct(meta)
        self._need(vec_share0.shape[0] == n_ct and scales.shape[0] == n_ct, 7, "bn: #ciphertexts")
        if meta.is_shared_input:
            self._need(vec_share1 is not None and vec_share1.shape[0] == n_ct, 7, "bn: #shares")
        out_cts, out_share = self.ctx._empty(n_ct, 2, 1, self.ctx.N), self.ctx._empty(n)
        L.check(self.ctx.lib.mmh_bn_server(self.ctx.handle, L.dptr(vec_share0),
                                           L.dptr(vec_share1 if meta.is_shared_input else None), L.dptr(scales), n,
                                           L.dptr(random), L.dptr(out_cts), L.dptr(out_share), L._stream_ptr()))
        return out_cts, out_share

    def bole(self, encrypted_x, encrypted_y, encoded_x, encoded_y, meta, random):  # :280-316
        import torch
        self._check(meta)
        n, n_ct = self._n_ct(meta)
        for v in (encrypted_x, encrypted_y, encoded_x, encoded_y):
            self._need(v.shape[0] == n_ct, 7, "bole: #ciphertexts")
        enc = torch.cat([encrypted_x, encrypted_y]).contiguous()
        pts = torch.cat([encoded_y, encoded_x]).contiguous()
        out_cts, out_share = self.ctx._empty(n_ct, 2, 1, self.ctx.N), self.ctx._empty(n)
        L.check(self.ctx.lib.mmh_bole_server(self.ctx.handle, L.dptr(enc), L.dptr(pts), n, L.dptr(random),
                                             L.dptr(out_cts), L.dptr(out_share), L._stream_ptr()))
        return out_cts, out_share

    def decryptToVector(self, in_vec, meta):  # :364-406
        self._check(meta)
        self._need(self.sk_ntt is not None, 1, "decryptToVector: no secret key")
        n, n_ct = self._n_ct(meta)
        self._need(in_vec.shape[0] == n_ct, 7, "decryptToVector: #ciphertexts")
        out = self.ctx._empty(n)
        L.check(self.ctx.lib.mmh_bn_decrypt_vector(self.ctx.handle, L.dptr(self.sk_last), L.dptr(in_vec), n,
                                                   L.dptr(out), L._stream_ptr()))
        return out
