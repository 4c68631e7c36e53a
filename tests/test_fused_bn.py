"""FusedBN filter scaling (networks/main_resnet50.cpp:1728-1788): the only floating point on the
path.  CPU: the C oracle against a Python restatement (IEEE doubles, round half away from zero) on random and
hand-picked halfway cases, and the all-zero exit rule.  GPU: the device kernel against the oracle bit for bit in both
output layouts, its flag, the bias add, and the scaled filters fed through prepared-filter HomConv against the
oracle's per-term schedule on the oracle-scaled filters."""
import ctypes as C
import math

import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64

SEED = 0x4D264D


def _bind(lib):
    lib.mmo_fused_bn_filters.restype = C.c_int
    lib.mmo_fused_bn_filters.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int,
                                         C.c_int, C.c_uint64, C.c_void_p]
    lib.mmo_fused_bn_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    return lib


def sign_value(x, bl):
    x &= (1 << bl) - 1
    return x - (1 << bl) if x >= 1 << (bl - 1) else x


def round_half_away(v):
    i = math.trunc(v)
    return i + int(math.copysign(1, v)) if abs(v - i) >= 0.5 else i


def py_scaled(f, s, bl, k):
    scale = float(1 << k)
    v = (sign_value(f, bl) / scale) * (sign_value(s, bl) / scale) * scale     # main_resnet50.cpp:1741-1748
    return round_half_away(v) % (1 << bl)                                       # sci::neg_mod with mod = 2^l


def make_case(bl, fh, fw, ci, co, seed, fbits=13, sbits=14):
    n = fh * fw * ci * co
    mask = np.uint64((1 << bl) - 1)
    f = ((splitmix64(seed, n) % np.uint64(1 << fbits)).astype(np.int64) - (1 << (fbits - 1))).astype(np.uint64) & mask
    s = ((splitmix64(seed + 1, co) % np.uint64(1 << sbits)).astype(np.int64) - (1 << (sbits - 1))).astype(np.uint64) & mask
    return f, s


@pytest.mark.parametrize("bl,k,fbits,sbits", [(37, 12, 13, 14), (41, 12, 13, 14), (41, 12, 36, 26), (32, 8, 10, 30)])
def test_oracle_fused_bn_equals_python_restatement(bl, k, fbits, sbits):
    lib = _bind(load())
    fh, fw, ci, co = 3, 3, 5, 24
    f, s = make_case(bl, fh, fw, ci, co, SEED + bl + fbits, fbits, sbits)
    half = 1 << (k - 1)
    # halfway cases: f * s / 2^k = x.5 exactly -> away from zero; and products that round to zero
    mask = (1 << bl) - 1
    s[:6] = np.array([half, half, half, 3 * half, (-half) & mask, 1], dtype=np.uint64)
    f[:6] = np.array([1, (-1) & mask, 3, 1, 5, 1], dtype=np.uint64)     # entries 0..5 use scales 0..5 (h=w=c=0)
    out = np.zeros_like(f)
    assert lib.mmo_fused_bn_filters(A(f), A(s), fh, fw, ci, co, bl, k, 1 << bl, A(out)) == 0
    want = np.array([py_scaled(int(a), int(s[i % co]), bl, k) for i, a in enumerate(f)], dtype=np.uint64)
    assert np.array_equal(out, want)
    assert [sign_value(int(v), bl) for v in out[:6]] == [1, -1, 2, 2, -3, 0]   # 0.5, -0.5, 1.5, 1.5, -2.5, 2^-k
    # all-zero exit rule (:1754-1762): tiny weights times tiny scales
    f0, s0 = np.full(f.shape, 1, np.uint64), np.full(co, 1, np.uint64)
    assert lib.mmo_fused_bn_filters(A(f0), A(s0), fh, fw, ci, co, bl, k, 1 << bl, A(out)) == 1 and not out.any()
    # bias (:1771-1786)
    px = 7
    o = splitmix64(SEED + 5, px * co)
    b = splitmix64(SEED + 6, co) & np.uint64(mask)
    ref = (o.reshape(px, co) + (b << np.uint64(k))[None, :]).reshape(-1)
    lib.mmo_fused_bn_bias(A(o), A(b), px, co, k)
    assert np.array_equal(o, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("bl,fbits,sbits", [(37, 13, 14), (41, 36, 26)])
def test_gpu_fused_bn_bit_exact(bl, fbits, sbits):
    import torch
    from openmm_amd import lib as L
    olib, lib = _bind(load()), L.load_library()
    k, fh, fw, ci, co = 12, 3, 3, 17, 40
    f, s = make_case(bl, fh, fw, ci, co, SEED + bl, fbits, sbits)
    s[:3] = np.array([1 << 11, 3 << 11, (-(1 << 11)) & ((1 << bl) - 1)], dtype=np.uint64)
    f[:3] = np.array([1, 1, 5], dtype=np.uint64)
    ref = np.zeros_like(f)
    assert olib.mmo_fused_bn_filters(A(f), A(s), fh, fw, ci, co, bl, k, 1 << bl, A(ref)) == 0
    d_f, d_s = L.to_device(f), L.to_device(s)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for oihw in (0, 1):
        out = torch.empty(f.size, dtype=torch.int64, device="cuda")
        L.check(lib.mmh_fused_bn_filters(L.dptr(d_f), L.dptr(d_s), fh, fw, ci, co, bl, k, 1 << bl, oihw, L.dptr(out),
                                         L.dptr(flag), None))
        torch.cuda.synchronize()
        want = ref.reshape(fh, fw, ci, co)
        want = want.transpose(3, 2, 0, 1).reshape(-1) if oihw else want.reshape(-1)
        assert np.array_equal(L.to_host(out), want) and int(flag) == 1
    ones, one_s = L.to_device(np.ones_like(f)), L.to_device(np.ones(co, np.uint64))
    L.check(lib.mmh_fused_bn_filters(L.dptr(ones), L.dptr(one_s), fh, fw, ci, co, bl, k, 1 << bl, 0, L.dptr(out),
                                     L.dptr(flag), None))
    torch.cuda.synchronize()
    assert int(flag) == 0 and not bool(out.any())          # the reference's "all zero filter" exit
    px = 50
    o, b = splitmix64(SEED + 5, px * co), splitmix64(SEED + 6, co) & np.uint64((1 << bl) - 1)
    d_o, d_b = L.to_device(o), L.to_device(b)
    L.check(lib.mmh_fused_bn_bias(L.dptr(d_o), L.dptr(d_b), px, co, k, None))
    olib.mmo_fused_bn_bias(A(o), A(b), px, co, k)
    assert np.array_equal(L.to_host(d_o), o)


@pytest.mark.gpu
def test_gpu_fused_bn_feeds_prepared_conv():
    """FusedBN as the networks use it: scaled filters straight into the prepared-filter server call, against the oracle
    convolution on the oracle-scaled filters (ResNet50's 3x3 128 -> 128 class at 8 filters)"""
    import torch
    import openmm_amd
    from openmm_amd import lib as L
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from tests.layer_fixtures import keypair, noise, rns_uniform, ternary
    from tests.oracle_binding import OracleCtx, uniform_mod
    from tests.oracle_layers import ConvOracle
    olib, lib = _bind(load()), L.load_library()
    bl, k, N = 37, 12, 4096
    T = 1 << bl
    c, h, w, fh, fw, M = 128, 28, 28, 3, 3, 8
    f, s = make_case(bl, fh, fw, c, M, SEED + 77)
    scaled = np.zeros_like(f)
    assert olib.mmo_fused_bn_filters(A(f), A(s), fh, fw, c, M, bl, k, T, A(scaled)) == 0
    filters = np.ascontiguousarray(scaled.reshape(fh, fw, c, M).transpose(3, 2, 0, 1))     # [M][C][fh][fw]
    assert all(filters[m].any() for m in range(M))
    o, g = OracleCtx(N, (60, 49), T), openmm_amd.Context(N, (60, 49), T)
    cv, meta = ConvOracle(o, (c, h, w), (c, fh, fw), M, 1, 1, True), Meta((c, h, w), (c, fh, fw), M, 1, 1, True)
    sk, pk = keypair(o)
    share0, share1 = uniform_mod(SEED + 1, (c, h, w), T), uniform_mod(SEED + 2, (c, h, w), T)
    cts = cv.encrypt_image(sk, share0, rns_uniform(o, SEED + 4, cv.n_in, 2), noise(SEED + 5, (cv.n_in, N)))
    u, e1 = ternary(SEED + 6, (cv.n_out, N)), noise(SEED + 7, (cv.n_out, N))
    mask = rns_uniform(o, SEED + 8, cv.n_out, 1)[:, 0]
    ref_out, ref_share = cv.server(cts, cv.encode_image(share1), cv.encode_filters(filters),
                                   np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64) if x.dtype == np.uint64 else x).cuda()
    d_scaled = torch.empty((M, c, fh, fw), dtype=torch.int64, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_f, d_s = dev(f), dev(s)      # named: a temporary would go back to torch's allocator before the launch
    L.check(lib.mmh_fused_bn_filters(L.dptr(d_f), L.dptr(d_s), fh, fw, c, M, bl, k, T, 1, L.dptr(d_scaled),
                                     L.dptr(flag), L._stream_ptr()))
    assert int(flag) == 1
    srv = HomConv2DSS()
    srv.setUp(g, pk_ntt=dev(pk))
    prep = srv.prepareFilters(d_scaled, meta)
    out, share = srv.conv2DSSPrepared(dev(cts), srv.encodeImage(dev(share1), meta), prep, meta,
                                      (dev(u), dev(e1), dev(mask)))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), ref_out)
    assert np.array_equal(share.cpu().numpy().view(np.uint64), ref_share)
