"""Reference-shaped layer flows composed from oracle primitives (CPU, slow, small shapes):
the loops of hom_conv2d_ss.cc / hom_fc_ss.cc / hom_bn_ss.cc written out one ciphertext at a
time, exactly as the reference schedules them."""
import ctypes as C

import numpy as np

from tests.oracle_binding import A, load

c_long3 = C.c_long * 3


def _l3(x):
    return c_long3(*[int(v) for v in x])


def setup_geom(lib):
    if getattr(lib, "_geom_ready", False):
        return
    vp, ci, cl, u64 = C.c_void_p, C.c_int, C.c_long, C.c_uint64
    lib.mmo_conv_out_shape.argtypes = [c_long3, c_long3, ci, ci, C.POINTER(cl)]
    lib.mmo_conv_shape_infer.argtypes = [cl, c_long3, c_long3, ci, ci, C.POINTER(cl), C.POINTER(ci), C.POINTER(ci)]
    lib.mmo_conv_encode_image.restype = cl
    lib.mmo_conv_encode_image.argtypes = [cl, u64, vp, c_long3, c_long3, ci, ci, vp]
    lib.mmo_conv_encode_filter.restype = cl
    lib.mmo_conv_encode_filter.argtypes = [cl, u64, vp, c_long3, c_long3, ci, ci, vp]
    lib.mmo_conv_indexer_create.restype = vp
    lib.mmo_conv_indexer_create.argtypes = [cl, c_long3, c_long3, ci, ci]
    lib.mmo_conv_indexer_destroy.argtypes = [vp]
    lib.mmo_conv_indexer_slices.argtypes = [vp, ci]
    lib.mmo_conv_indexer_get.argtypes = [vp, ci, ci, C.POINTER(cl), vp]
    lib.mmo_fc_split.argtypes = [cl, cl, cl, C.POINTER(cl)]
    lib.mmo_fc_encode_input.restype = cl
    lib.mmo_fc_encode_input.argtypes = [cl, u64, vp, cl, cl, vp]
    lib.mmo_fc_encode_weights.restype = cl
    lib.mmo_fc_encode_weights.argtypes = [cl, u64, vp, cl, cl, cl, cl, vp]
    lib._geom_ready = True


class ConvOracle:
    """HomConv2DSS flow for one Meta on the oracle context `o`."""

    def __init__(self, o, ishape, fshape, n_filters, same, stride, shared):
        self.o, self.lib = o, load()
        setup_geom(self.lib)
        self.ishape, self.fshape, self.M = tuple(ishape), tuple(fshape), n_filters
        self.same, self.stride, self.shared = int(same), int(stride), shared
        N = o.N
        strided, pads, ss = (C.c_long * 3)(), (C.c_int * 2)(), (C.c_int * 3)()
        rc = self.lib.mmo_conv_shape_infer(N, _l3(ishape), _l3(fshape), self.same, self.stride, strided, pads, ss)
        assert rc == 0, "shape inference failed"
        self.strided, self.pads, self.slice_strides = tuple(strided), tuple(pads), tuple(ss)
        ohw = (C.c_long * 2)()
        assert self.lib.mmo_conv_out_shape(_l3(ishape), _l3(fshape), self.same, self.stride, ohw) == 0
        self.out_shape = (n_filters, ohw[0], ohw[1])
        self.n_in = self.lib.mmo_conv_encode_image(N, o.t, None, _l3(ishape), _l3(fshape), self.same, self.stride, None)
        self.n_cs = self.lib.mmo_conv_encode_filter(N, o.t, None, _l3(ishape), _l3(fshape), self.same, self.stride, None)
        self.n_sp = self.n_in // self.n_cs
        self.n_out = n_filters * self.n_sp
        compressed = int(np.prod(self.strided)) < int(np.prod(ishape))
        # hom_conv2d_ss.cc:741-744 / :877-880
        self.mask_ix = self._indexer(self.strided if compressed else ishape, 0 if compressed else self.same,
                                     1 if compressed else self.stride)
        # hom_conv2d_ss.cc:817-819
        self.keep_ix = self._indexer(self.strided, 0, 1)

    def _indexer(self, shape, same, stride):
        h = self.lib.mmo_conv_indexer_create(self.o.N, _l3(shape), _l3(self.fshape), int(same), int(stride))
        assert h
        return h

    def targets(self, ix, sh, sw):
        hw = (C.c_long * 2)()
        assert self.lib.mmo_conv_indexer_get(ix, sh, sw, hw, None) == 0
        idx = np.zeros(hw[0] * hw[1], np.uint64)
        assert self.lib.mmo_conv_indexer_get(ix, sh, sw, hw, A(idx)) == 0
        return (hw[0], hw[1]), idx

    def encode_image(self, img):
        out = np.zeros((self.n_in, self.o.N), np.uint64)
        img = np.ascontiguousarray(img, np.uint64)
        n = self.lib.mmo_conv_encode_image(self.o.N, self.o.t, A(img), _l3(self.ishape), _l3(self.fshape), self.same,
                                           self.stride, A(out))
        assert n == self.n_in
        return out

    def encode_filters(self, filters):
        out = np.zeros((self.M, self.n_cs, self.o.N), np.uint64)
        for m in range(self.M):
            f = np.ascontiguousarray(filters[m], np.uint64)
            n = self.lib.mmo_conv_encode_filter(self.o.N, self.o.t, A(f), _l3(self.ishape), _l3(self.fshape),
                                                self.same, self.stride, A(out[m]))
            assert n == self.n_cs
        return out

    def encrypt_image(self, sk, img, a, e):
        pts = self.encode_image(img)
        cts = np.zeros((self.n_in, 2, self.o.L, self.o.N), np.uint64)
        for k in range(self.n_in):
            self.lib.mmo_encrypt_symmetric(self.o.p, 0, A(sk), A(a[k]), A(e[k]), A(pts[k]), self.o.N, A(cts[k]))
        return cts

    def server(self, in_cts, share_pts, filter_pts, pk1, u, e1, mask, exact=None):
        """conv2DSS, USE_APPROX_RESHARE build; exact=(flood, e0) selects the other build"""
        o, lib, N, L = self.o, self.lib, self.o.N, self.o.L
        image = in_cts.copy()
        if self.shared:
            for k in range(self.n_in):                                   # :655-671
                lib.mmo_add_plain(o.p, 0, A(image[k]), A(share_pts[k]), N)
        out = np.zeros((self.n_out, 2, L, N), np.uint64)
        tmp = np.zeros((2, L, N), np.uint64)
        for m in range(self.M):                                          # conv2DOneFilter :563-619
            started = [False] * self.n_sp
            for c in range(self.n_cs):
                if not filter_pts[m, c].any():
                    continue
                for i in range(self.n_sp):
                    ii = c * self.n_sp + i
                    dst = out[m * self.n_sp + i]
                    if started[i]:
                        lib.mmo_multiply_plain(o.p, 0, A(image[ii]), A(filter_pts[m, c]), N, A(tmp))
                        lib.mmo_ct_add(o.p, 0, A(dst), A(tmp), 2)
                    else:
                        lib.mmo_multiply_plain(o.p, 0, A(image[ii]), A(filter_pts[m, c]), N, A(dst))
                        started[i] = True
            assert all(started), "filter with all zero is not supported"
        low = np.zeros((self.n_out, 2, 1, N), np.uint64)
        share = np.zeros(self.out_shape, np.uint64)
        zero = np.zeros((2, 1, N), np.uint64)
        m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        lib.mmo_truncate_masks(o.p, A(m0), A(m1))
        nsh, nsw = lib.mmo_conv_indexer_slices(self.mask_ix, 1), lib.mmo_conv_indexer_slices(self.mask_ix, 2)
        assert nsh * nsw == self.n_sp
        q0 = np.uint64(o.q[0])
        for m in range(self.M):                                          # addRandomMask :761-788
            cid, hoff = m * self.n_sp, 0
            for sh in range(nsh):
                woff = 0
                for sw in range(nsw):
                    (th, tw), tg = self.targets(self.mask_ix, sh, sw)
                    ct = low[cid]
                    if exact is not None:
                        r = exact_reshare(o, lib, out[cid], ct, pk1, exact[0][cid], u[cid], exact[1][cid], e1[cid],
                                          mask[cid])
                        share[m, hoff:hoff + th, woff:woff + tw] = r[tg.astype(np.int64)].reshape(th, tw)
                        woff += tw
                        cid += 1
                        continue
                    lib.mmo_mod_switch_to_last(o.p, A(out[cid]), A(ct), 2)                 # flood :382
                    lib.mmo_encrypt_zero_asymmetric(o.p, 1, A(pk1), A(u[cid]), None, A(e1[cid]), A(zero))
                    lib.mmo_ct_add(o.p, 1, A(ct), A(zero), 2)
                    r = mask[cid]
                    gathered = np.ascontiguousarray(r[tg.astype(np.int64)])               # :209-221
                    coeffs = np.zeros(len(tg), np.uint64)
                    lib.mmo_decrypt_scale_and_round(o.p, 1, A(gathered), len(tg), A(coeffs))
                    ct[0, 0] = np.where(ct[0, 0] >= r, ct[0, 0] - r, ct[0, 0] + q0 - r)   # sub_poly_inplace
                    share[m, hoff:hoff + th, woff:woff + tw] = coeffs.reshape(th, tw)
                    woff += tw
                    cid += 1
                hoff += th
        low[:, 0] &= m0[0]                                               # truncate_for_decryption
        low[:, 1] &= m1[0]
        for m in range(self.M):                                          # removeUnusedCoeffs :828-847
            cid = m * self.n_sp
            for sh in range(nsh):
                for sw in range(nsw):
                    _, tg = self.targets(self.keep_ix, sh, sw)
                    keep = np.zeros(N, bool)
                    keep[tg.astype(np.int64)] = True
                    low[cid, 0, 0, ~keep] = 0
                    cid += 1
        return low, share

    def decrypt(self, sk1, cts):
        """decryptToTensor :853-941"""
        o, lib, N = self.o, self.lib, self.o.N
        out = np.zeros(self.out_shape, np.uint64)
        pt = np.zeros(N, np.uint64)
        nsh, nsw = lib.mmo_conv_indexer_slices(self.mask_ix, 1), lib.mmo_conv_indexer_slices(self.mask_ix, 2)
        for m in range(self.M):
            cid, hoff = m * self.n_sp, 0
            for sh in range(nsh):
                woff = 0
                for sw in range(nsw):
                    (th, tw), tg = self.targets(self.mask_ix, sh, sw)
                    lib.mmo_decrypt(o.p, 1, A(sk1), A(cts[cid]), A(pt))
                    out[m, hoff:hoff + th, woff:woff + tw] = pt[tg.astype(np.int64)].reshape(th, tw)
                    woff += tw
                    cid += 1
                hoff += th
        return out


def ideal_conv(img, filters, same, stride, t):
    """Tensor::Conv2D (tensor.h:274-362) on signed representatives, result mod t"""
    C_, H, W = img.shape
    M, _, fh, fw = filters.shape
    ph, pw = ((fh - 1), (fw - 1)) if same else (0, 0)
    top, left = (ph + 1) // 2, (pw + 1) // 2
    x = np.zeros((C_, H + ph, W + pw), dtype=object)
    x[:, top:top + H, left:left + W] = img.astype(object)
    oh, ow = (H + ph - fh + stride) // stride, (W + pw - fw + stride) // stride
    out = np.zeros((M, oh, ow), dtype=object)
    f = filters.astype(object)
    for r in range(fh):
        for c in range(fw):
            patch = x[:, r:r + (oh - 1) * stride + 1:stride, c:c + (ow - 1) * stride + 1:stride]
            out += np.einsum("mc,chw->mhw", f[:, :, r, c], patch)
    return out % t


class FcOracle:
    """HomFCSS flow (hom_fc_ss.cc) on the oracle context."""

    def __init__(self, o, rows, cols, shared):
        self.o, self.lib, self.rows, self.cols, self.shared = o, load(), rows, cols, shared
        setup_geom(self.lib)
        sp = (C.c_long * 2)()
        self.lib.mmo_fc_split(rows, cols, o.N, sp)
        self.d0, self.d1 = sp[0], sp[1]
        self.n_in, self.n_out = -(-cols // self.d1), -(-rows // self.d0)

    def encode_input(self, vec):
        out = np.zeros((self.n_in, self.o.N), np.uint64)
        v = np.ascontiguousarray(vec, np.uint64)
        assert self.lib.mmo_fc_encode_input(self.o.N, self.o.t, A(v), self.cols, self.d1, A(out)) == self.n_in
        return out

    def encode_weights(self, W):
        out = np.zeros((self.n_out, self.n_in, self.o.N), np.uint64)
        w = np.ascontiguousarray(W, np.uint64)
        n = self.lib.mmo_fc_encode_weights(self.o.N, self.o.t, A(w), self.rows, self.cols, self.d0, self.d1, A(out))
        assert n == self.n_in * self.n_out
        return out

    def encrypt_input(self, sk, vec, a, e):
        pts = self.encode_input(vec)
        cts = np.zeros((self.n_in, 2, self.o.L, self.o.N), np.uint64)
        for k in range(self.n_in):
            self.lib.mmo_encrypt_symmetric(self.o.p, 0, A(sk), A(a[k]), A(e[k]), A(pts[k]), self.o.N, A(cts[k]))
        return cts

    def server(self, in_cts, share_pts, wpts, pk1, u, e1, mask, exact=None):
        o, lib, N, L = self.o, self.lib, self.o.N, self.o.L
        inp = in_cts.copy()
        if self.shared:
            for k in range(self.n_in):
                lib.mmo_add_plain(o.p, 0, A(inp[k]), A(share_pts[k]), N)
        out = np.zeros((self.n_out, 2, L, N), np.uint64)
        tmp = np.zeros((2, L, N), np.uint64)
        for j in range(self.n_out):                                      # :443-457
            lib.mmo_multiply_plain(o.p, 0, A(inp[0]), A(wpts[j, 0]), N, A(out[j]))
            for i in range(1, self.n_in):
                lib.mmo_multiply_plain(o.p, 0, A(inp[i]), A(wpts[j, i]), N, A(tmp))
                lib.mmo_ct_add(o.p, 0, A(out[j]), A(tmp), 2)
        low = np.zeros((self.n_out, 2, 1, N), np.uint64)
        share = np.zeros(self.rows, np.uint64)
        zero = np.zeros((2, 1, N), np.uint64)
        targets = np.arange(self.d0, dtype=np.int64) * self.d1           # :482-485
        q0 = np.uint64(o.q[0])
        for j in range(self.n_out):                                      # :494-511
            if exact is not None:
                r = exact_reshare(o, lib, out[j], low[j], pk1, exact[0][j], u[j], exact[1][j], e1[j], mask[j])
                r0, r1 = j * self.d0, min((j + 1) * self.d0, self.rows)
                share[r0:r1] = r[targets][:r1 - r0]
                continue
            lib.mmo_mod_switch_to_last(o.p, A(out[j]), A(low[j]), 2)
            lib.mmo_encrypt_zero_asymmetric(o.p, 1, A(pk1), A(u[j]), None, A(e1[j]), A(zero))
            lib.mmo_ct_add(o.p, 1, A(low[j]), A(zero), 2)
            r = mask[j]
            gathered = np.ascontiguousarray(r[targets])
            coeffs = np.zeros(len(targets), np.uint64)
            lib.mmo_decrypt_scale_and_round(o.p, 1, A(gathered), len(targets), A(coeffs))
            low[j, 0, 0] = np.where(low[j, 0, 0] >= r, low[j, 0, 0] - r, low[j, 0, 0] + q0 - r)
            r0, r1 = j * self.d0, min((j + 1) * self.d0, self.rows)
            share[r0:r1] = coeffs[:r1 - r0]
        m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        lib.mmo_truncate_masks(o.p, A(m0), A(m1))
        low[:, 0] &= m0[0]
        low[:, 1] &= m1[0]
        for j in range(self.n_out):                                      # :538-556
            r0, r1 = j * self.d0, min((j + 1) * self.d0, self.rows)
            upper = (r1 - r0) * self.d1
            idx = np.arange(N)
            keep = (idx < upper) & (idx % self.d1 == 0)
            low[j, 0, 0, ~keep] = 0
        return low, share

    def decrypt(self, sk1, cts):
        out = np.zeros(self.rows, np.uint64)
        pt = np.zeros(self.o.N, np.uint64)
        for j in range(self.n_out):                                      # :589-599
            self.lib.mmo_decrypt(self.o.p, 1, A(sk1), A(cts[j]), A(pt))
            r0, r1 = j * self.d0, min((j + 1) * self.d0, self.rows)
            out[r0:r1] = pt[(np.arange(r1 - r0)) * self.d1]
        return out


def flood_noise(o, flood_words):
    """set_poly_coeffs_uniform as flood_ciphertext calls it (hom_conv2d_ss.cc:142-177, :358-368):
    [L][N] residues of (word & bitmask), bitmask = (1 << (bitlen % 64)) - 1 as written"""
    Q = 1
    for q in o.q:
        Q *= q
    bitlen = min(64, Q.bit_length() - int(o.t).bit_length() - 1)
    bitmask = np.uint64((1 << (bitlen % 64)) - 1)
    r = flood_words & bitmask
    return np.stack([r % np.uint64(q) for q in o.q])


def exact_reshare(o, lib, acc, low, pk1, flood, u, e0, e1, mask_words):
    """flood_ciphertext + sample_random_mask + sub_poly_inplace of the USE_APPROX_RESHARE=0 build for
    one ciphertext; returns the mask polynomial mod t (the share is read from it at the targets)"""
    N = o.N
    noise = flood_noise(o, flood)
    for l, q in enumerate(o.q):                                          # :370-379, c0 only
        lib.mmo_add_mod(A(acc[0, l]), A(noise[l]), A(acc[0, l]), N, q)
    lib.mmo_mod_switch_to_last(o.p, A(acc), A(low), 2)                   # :382
    zero = np.zeros((2, 1, N), np.uint64)
    lib.mmo_encrypt_zero_asymmetric(o.p, 1, A(pk1), A(u), A(e0), A(e1), A(zero))   # :327 keeps e0
    lib.mmo_ct_add(o.p, 1, A(low), A(zero), 2)
    t = int(o.t)
    r = (mask_words & np.uint64(t - 1)) if t & (t - 1) == 0 else mask_words % np.uint64(t)   # :257-265
    lib.mmo_sub_plain(o.p, 1, A(low), A(np.ascontiguousarray(r)), N)     # sub_poly_inplace :74-81
    return r


class BnOracle:
    """HomBNSS SIMD path (hom_bn_ss.cc:151-406) on an oracle context with t = p."""

    def __init__(self, o, n):
        self.o, self.lib, self.n = o, load(), n
        self.sub = min(n, o.N)
        self.n_ct = -(-n // self.sub)

    def encode(self, vec):
        N = self.o.N
        out = np.zeros((self.n_ct, N), np.uint64)
        for k in range(self.n_ct):
            chunk = np.zeros(N, np.uint64)
            part = vec[k * self.sub:min((k + 1) * self.sub, self.n)]
            chunk[:len(part)] = part
            self.lib.mmo_batch_encode(self.o.p, A(chunk), N, A(out[k]))
        return out

    def encrypt(self, sk, vec, a, e):
        pts = self.encode(vec)
        cts = np.zeros((self.n_ct, 2, self.o.L, self.o.N), np.uint64)
        for k in range(self.n_ct):
            self.lib.mmo_encrypt_symmetric(self.o.p, 0, A(sk), A(a[k]), A(e[k]), A(pts[k]), self.o.N, A(cts[k]))
        return cts

    def _tail(self, acc, random):
        o, lib, N, p = self.o, self.lib, self.o.N, np.uint64(self.o.t)
        low = np.zeros((self.n_ct, 2, 1, N), np.uint64)
        for k in range(self.n_ct):
            lib.mmo_mod_switch_to_last(o.p, A(acc[k]), A(low[k]), 2)    # :269 / :307
        r = random % p                                                  # :333-334
        share = np.where(r != 0, p - r, np.uint64(0))                   # :338-340
        rpts = self.encode(r)
        m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        lib.mmo_truncate_masks(o.p, A(m0), A(m1))
        for k in range(self.n_ct):                                      # :346-357
            lib.mmo_add_plain(o.p, 1, A(low[k]), A(rpts[k]), N)
        low[:, 0] &= m0[0]
        low[:, 1] &= m1[0]
        return low, share

    def bn(self, cts, share_pts, scale_pts, random):
        o, lib, N = self.o, self.lib, self.o.N
        acc = cts.copy()
        for k in range(self.n_ct):                                      # :262-268
            if share_pts is not None:
                lib.mmo_add_plain(o.p, 0, A(acc[k]), A(share_pts[k]), N)
            lib.mmo_multiply_plain(o.p, 0, A(acc[k]), A(scale_pts[k]), N, A(acc[k]))
        return self._tail(acc, random)

    def bole(self, ex, ey, px, py, random):
        o, lib, N = self.o, self.lib, self.o.N
        acc = np.zeros_like(ex)
        tmp = np.zeros_like(ex[0])
        for k in range(self.n_ct):                                      # :303-307
            lib.mmo_multiply_plain(o.p, 0, A(ex[k]), A(py[k]), N, A(acc[k]))
            lib.mmo_multiply_plain(o.p, 0, A(ey[k]), A(px[k]), N, A(tmp))
            lib.mmo_ct_add(o.p, 0, A(acc[k]), A(tmp), 2)
        return self._tail(acc, random)

    def decrypt(self, sk1, cts):
        N = self.o.N
        out = np.zeros(self.n, np.uint64)
        pt, slots = np.zeros(N, np.uint64), np.zeros(N, np.uint64)
        for k in range(self.n_ct):                                      # :385-392
            self.lib.mmo_decrypt(self.o.p, 1, A(sk1), A(cts[k]), A(pt))
            self.lib.mmo_batch_decode(self.o.p, A(pt), A(slots))
            lo, hi = k * self.sub, min((k + 1) * self.sub, self.n)
            out[lo:hi] = slots[:hi - lo]
        return out


class BnDirectOracle:
    """HomBNSS direct path (hom_bn_ss.cc:471-790) on an oracle context with t = 2^k, followed
    line by line, including the dense indexing of the re-sharing / decrypt side."""

    def __init__(self, o, ishape, shared=True):
        import math
        self.o, self.lib, self.ishape, self.shared = o, load(), tuple(ishape), shared
        n = math.isqrt(o.N)                                             # getSplitBN :63
        C_, H, W = ishape
        self.sh, self.sw = min(n, H), min(n, W)
        self.dH, self.dW = -(-H // self.sh), -(-W // self.sw)
        self.n_ct = C_ * self.dH * self.dW

    def _tile(self, cid):
        c, hb, wb = cid // (self.dH * self.dW), (cid // self.dW) % self.dH, cid % self.dW
        return c, hb * self.sh, wb * self.sw

    def encode(self, tensor):                                           # :471-520
        N, t = self.o.N, self.o.t
        C_, H, W = self.ishape
        out = np.zeros((self.n_ct, N), np.uint64)
        for cid in range(self.n_ct):
            c, h0, w0 = self._tile(cid)
            tile = np.zeros((self.sh, self.sw), np.uint64)              # SlicedPaddedTensor pads with zeros
            vh, vw = min(self.sh, H - h0), min(self.sw, W - w0)
            tile[:vh, :vw] = tensor[c, h0:h0 + vh, w0:w0 + vw]
            out[cid, :self.sh * self.sw] = tile.ravel() % np.uint64(t)  # vec2PolyBFV :465
        return out

    def encrypt(self, sk, tensor, a, e):                                # :522-576
        pts = self.encode(tensor)
        cts = np.zeros((self.n_ct, 2, self.o.L, self.o.N), np.uint64)
        for k in range(self.n_ct):
            self.lib.mmo_encrypt_symmetric(self.o.p, 0, A(sk), A(a[k]), A(e[k]), A(pts[k]), self.o.N, A(cts[k]))
        return cts

    def server(self, cts, share_pts, scales, pk, u, e0, e1, share1):    # :578-738
        o, lib, N, t = self.o, self.lib, self.o.N, self.o.t
        C_, H, W = self.ishape
        acc = cts.copy()
        if self.shared:                                                 # :598-604
            for k in range(self.n_ct):
                lib.mmo_add_plain(o.p, 0, A(acc[k]), A(share_pts[k]), N)
        for c in range(C_):                                             # :613-645
            s = int(np.int64(scales[c]))
            v = abs(s) % t
            for cid in range(c * self.dH * self.dW, (c + 1) * self.dH * self.dW):
                for l, q in enumerate(o.q):
                    scalar = v % q
                    if s < 0 and scalar:
                        scalar = q - scalar
                    vec = np.full(N, scalar, np.uint64)
                    for comp in range(2):
                        lib.mmo_dyadic_mul(A(acc[cid, comp, l]), A(vec), A(acc[cid, comp, l]), N, q)
        rnd = self.encode(share1)                                       # :725
        out = np.zeros((self.n_ct, 2, 1, N), np.uint64)
        m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        lib.mmo_truncate_masks(o.p, A(m0), A(m1))
        zero, zero_low = np.zeros((2, o.L, N), np.uint64), np.zeros((2, 1, N), np.uint64)
        for cid in range(self.n_ct):                                    # :648-687
            lib.mmo_mod_switch_to_last(o.p, A(acc[cid]), A(out[cid]), 2)
            lib.mmo_sub_plain(o.p, 1, A(out[cid]), A(rnd[cid]), N)
            # Encryptor::encrypt_zero(last_parms_id): pk encryption one level up, then divide_and_round_q_last
            lib.mmo_encrypt_zero_asymmetric(o.p, 0, A(pk), A(u[cid]), A(e0[cid]), A(e1[cid]), A(zero))
            lib.mmo_mod_switch_to_last(o.p, A(zero), A(zero_low), 2)
            lib.mmo_ct_add(o.p, 1, A(out[cid]), A(zero_low), 2)
            _, h0, w0 = self._tile(cid)
            used = min(self.sh, H - h0) * min(self.sw, W - w0)          # indices 0 .. used-1, :666-678
            out[cid, 0, 0, used:] = 0                                   # remove_unused_coeffs: c0 only
            out[cid, 0] &= m0[0]
            out[cid, 1] &= m1[0]
        return out

    def decrypt(self, sk1, cts):                                        # :740-790
        N = self.o.N
        C_, H, W = self.ishape
        out = np.zeros(self.ishape, np.uint64)
        pt = np.zeros(N, np.uint64)
        for cid in range(self.n_ct):
            self.lib.mmo_decrypt(self.o.p, 1, A(sk1), A(cts[cid]), A(pt))
            c, h0, w0 = self._tile(cid)
            vh, vw = min(self.sh, H - h0), min(self.sw, W - w0)
            out[c, h0:h0 + vh, w0:w0 + vw] = pt[:vh * vw].reshape(vh, vw)
        return out
