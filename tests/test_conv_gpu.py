"""GPU parity of the HomConv2DSS layer calls (C ABI via the Python mirror) against the
oracle's reference-shaped flow: encoders, encrypted image, server output ciphertexts, share
tensor and decrypted tensor, bit for bit on identical randomness."""
import os

import numpy as np
import pytest

from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary
from tests.oracle_binding import OracleCtx, splitmix64, uniform_mod
from tests.oracle_layers import ConvOracle, ideal_conv

pytestmark = pytest.mark.gpu
T = 1 << 37
N = 4096
CASES = [
    ((3, 8, 8), (3, 3), 2, 1, 1, True),
    ((2, 70, 70), (3, 3), 1, 1, 1, True),
    ((5, 12, 12), (1, 1), 3, 0, 2, True),
    ((4, 17, 17), (3, 3), 2, 0, 2, True),
    ((6, 30, 30), (5, 5), 2, 1, 1, True),
    ((3, 8, 8), (3, 3), 2, 1, 1, False),
    ((3, 230, 230), (7, 7), 2, 0, 2, False),   # ResNet50 conv1 geometry (Appendix D), 2 filters
    ((8, 9, 11), (2, 3), 3, 1, 1, True),       # fh != fw: FilterIndexer reads "h" from the width (Appendix B)
    ((130, 7, 7), (1, 1), 2, 0, 1, True),      # 83 channels per polynomial, ragged last channel slice
    # 1x1 filters with few channels per polynomial take the channel-wise matrix-product route:
    ((6, 56, 56), (1, 1), 3, 0, 1, True),      # one channel per polynomial (constant filter polynomials)
    ((12, 28, 28), (1, 1), 4, 0, 1, True),     # five channels per polynomial, ragged last slice (5 + 5 + 2)
    ((2, 70, 70), (1, 1), 2, 0, 1, False),     # spatial windows as well
    ((7, 56, 56), (1, 1), 70, 0, 2, True),     # stride trick (28x28 after subsampling), more than 64 filters
    ((40, 14, 14), (1, 1), 3, 0, 1, True),     # 20 channels per polynomial, 40 channels = three staging chunks
    ((1, 4, 4), (1, 1), 1, 0, 1, True),        # smallest layer: one channel, one filter, one ciphertext
    ((1, 5, 5), (3, 3), 1, 1, 1, False),       # smallest 3x3 layer
    ((520, 7, 7), (3, 3), 2, 1, 1, True),      # 4680 taps: near the direct route's limit (74 steps of 64 taps)
]


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def plan_kept(cv, ref_out):
    """longest list of kept c0 coefficients over the spatial windows: bounded by the non-zero pattern of the
    reference ciphertexts (remove_unused_coeffs zeroes everything else; a kept coefficient may be zero by chance)"""
    nz = (ref_out[:, 0, 0, :] != 0).reshape(-1, cv.n_sp, ref_out.shape[-1]).any(axis=0)
    return max(int(nz.sum(axis=1).max()), 1) + 64


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


# weight magnitudes: two byte digits (< 32,640), three (< 2^20), beyond the fast forms (the whole ring);
# layers: 1x1 with several filter tiles of 128, 3x3 with more than one 64-tap step
@pytest.mark.parametrize("wbound", [1 << 12, 1 << 18, 1 << 36])
@pytest.mark.parametrize("ishape,fhw,M,same", [((11, 28, 28), (1, 1), 5, 0), ((70, 14, 14), (1, 1), 140, 0),
                                               ((9, 12, 12), (3, 3), 3, 1)])
def test_direct_route_weight_forms(mm, monkeypatch, wbound, ishape, fhw, M, same):
    """direct route forced, weights of every form, from the encoded plaintexts and from the raw tensors: all must
    equal the oracle's per-term multiply_plain schedule bit for bit"""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    monkeypatch.setenv("MMH_CONV_POINTWISE", "1")
    fshape = (ishape[0],) + tuple(fhw)
    o, g = OracleCtx(N, (60, 49), T), mm.Context(N, (60, 49), T)
    cv, meta = ConvOracle(o, ishape, fshape, M, same, 1, True), Meta(ishape, fshape, M, same, 1, True)
    sk, pk = keypair(o)
    server = HomConv2DSS()
    server.setUp(g, pk_ntt=dev(pk))
    share0, share1 = uniform_mod(SEED + 1, ishape, T), uniform_mod(SEED + 2, ishape, T)
    filters = signed_weights(SEED + 3, (M,) + fshape, wbound, T)
    a, e = rns_uniform(o, SEED + 4, cv.n_in, 2), noise(SEED + 5, (cv.n_in, N))
    cts, f, s1 = cv.encrypt_image(sk, share0, a, e), cv.encode_filters(filters), cv.encode_image(share1)
    u, e1 = ternary(SEED + 6, (cv.n_out, N)), noise(SEED + 7, (cv.n_out, N))
    mask = rns_uniform(o, SEED + 8, cv.n_out, 1)[:, 0]
    ref_out, ref_share = cv.server(cts, s1, f, np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    rand = (dev(u), dev(e1), dev(mask))
    prep = server.prepareFilters(dev(filters), meta)
    for got in (server.conv2DSS(dev(cts), dev(s1), dev(f), meta, rand),
                server.conv2DSSFromWeights(dev(cts), dev(s1), dev(filters), meta, rand),
                server.conv2DSSFromTensors(dev(cts), dev(share1), dev(filters), meta, rand),
                server.conv2DSSPrepared(dev(cts), dev(s1), prep, meta, rand)):
        assert np.array_equal(host(got[0]), ref_out) and np.array_equal(host(got[1]), ref_share)


# the server multiplies either through filter transforms or through the direct tap-wise matrix product; the
# plan's cost model picks one ("model"), MMH_CONV_POINTWISE forces the transform ("0") or the direct route ("1")
@pytest.mark.parametrize("route", ["model", "0", "1", "1p", "1n"])
@pytest.mark.parametrize("ishape,fhw,M,same,stride,shared", [((9, 20, 20), (3, 3), 5, 1, 1, True),
                                                             ((40, 14, 14), (1, 1), 70, 0, 1, True),
                                                             ((3, 33, 33), (3, 3), 4, 0, 2, False)])
def test_conv_layer_bit_exact_n8192(mm, monkeypatch, ishape, fhw, M, same, stride, shared, route):
    """the same flow on an N = 8192 context with two moduli (not a configuration the reference's conv runs, which is
    N = 4096, cheetah-api.cpp:73-105, but one the entry points accept): the LOGN = 13 forms of the front launch, the
    transforms and the fused tail against the oracle"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "N", 8192)
    test_conv_layer_bit_exact(mm, monkeypatch, ishape, fhw, M, same, stride, shared, route)


@pytest.mark.parametrize("route", ["0", "1", "1p", "1n"])
@pytest.mark.parametrize("ring_bits", [32, 41, 44])
def test_conv_layer_bit_exact_other_rings(mm, monkeypatch, ring_bits, route):
    """the ring sizes of the other shipped networks (bitlength 32 for SqueezeNet / DenseNet121, 41 in
    sci::default_prime_mod's table, 44 the largest CheetahLinear accepts): the tail's masks, truncation widths and
    scale-and-round constants all depend on t"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "T", 1 << ring_bits)
    test_conv_layer_bit_exact(mm, monkeypatch, (9, 20, 20), (3, 3), 5, 1, 1, True, route)
    test_conv_layer_bit_exact(mm, monkeypatch, (40, 14, 14), (1, 1), 70, 0, 2, True, route)


@pytest.mark.parametrize("route", ["model", "0", "1", "1p", "1n"])
@pytest.mark.parametrize("ishape,fhw,M,same,stride,shared", CASES)
def test_conv_layer_bit_exact(mm, monkeypatch, ishape, fhw, M, same, stride, shared, route):
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    if route != "model":
        monkeypatch.setenv("MMH_CONV_POINTWISE", route[0])   # read when the plan is made
    # "1p": the direct route with the product kernel's items owning both limbs and the tail fed the mod-switched
    # accumulator (the library pairs only layers with enough items by itself: none of these small cases); "1": never
    # "1n": layers of at most 32 filters through the kernel that builds its digits itself (the default for them);
    # "1" and "1p" keep them on the digit image, so that the wide kernel sees these small cases too
    for var, val in (("MMH_PW_PAIRED", {"1": "0", "1p": "1"}.get(route)), ("MMH_PW_NARROW", {"1": "0", "1p": "0", "1n": "1"}.get(route))):
        if val is None:
            monkeypatch.delenv(var, raising=False)
        else:
            monkeypatch.setenv(var, val)
    o, g = OracleCtx(N, (60, 49), T), mm.Context(N, (60, 49), T)
    fshape = (ishape[0],) + tuple(fhw)
    cv = ConvOracle(o, ishape, fshape, M, same, stride, shared)
    meta = Meta(ishape, fshape, M, same, stride, shared)
    sk, pk = keypair(o)
    client, server = HomConv2DSS(), HomConv2DSS()
    client.setUp(g, sk_ntt=dev(sk))
    server.setUp(g, pk_ntt=dev(pk))
    plan = server.plan(meta)
    assert (plan.n_in, plan.n_cslices, plan.n_spatial, plan.n_out) == (cv.n_in, cv.n_cs, cv.n_sp, cv.n_out)
    assert plan.out_shape == cv.out_shape and plan.slice_strides == cv.slice_strides

    img = uniform_mod(SEED + 20, ishape, T)
    share0 = uniform_mod(SEED + 21, ishape, T)
    share1 = (img - share0) & np.uint64(T - 1)
    filters = signed_weights(SEED + 22, (M,) + fshape, 1 << 12, T)
    a, e = rns_uniform(o, SEED + 30, cv.n_in, 2), noise(SEED + 31, (cv.n_in, N))
    client_in = share0 if shared else img
    # client: encryptImage
    ref_cts = cv.encrypt_image(sk, client_in, a, e)
    got_cts = client.encryptImage(dev(client_in), meta, (dev(a), dev(e)))
    assert np.array_equal(host(got_cts), ref_cts)
    # server: encoders
    ref_f = cv.encode_filters(filters)
    got_f = server.encodeFilters(dev(filters), meta)
    assert np.array_equal(host(got_f), ref_f)
    ref_s = got_s = None
    if shared:
        ref_s = cv.encode_image(share1)
        got_s = server.encodeImage(dev(share1), meta)
        assert np.array_equal(host(got_s), ref_s)
    # server: conv2DSS
    u, e1 = ternary(SEED + 40, (cv.n_out, N)), noise(SEED + 41, (cv.n_out, N))
    mask = rns_uniform(o, SEED + 50, cv.n_out, 1)[:, 0]
    ref_out, ref_share = cv.server(ref_cts, ref_s, ref_f, np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    got_out, got_share = server.conv2DSS(got_cts, got_s, got_f, meta, (dev(u), dev(e1), dev(mask)))
    assert np.array_equal(host(got_out), ref_out)
    assert np.array_equal(host(got_share), ref_share)
    # the one-call form from the raw filters (encode + lift + NTT on chip) gives the same bits
    one_out, one_share = server.conv2DSSFromWeights(got_cts, got_s, dev(filters), meta, (dev(u), dev(e1), dev(mask)))
    assert np.array_equal(host(one_out), ref_out)
    # ... and so does the form that also takes the server's share as the raw tensor (CheetahLinear::conv2d in one call)
    t_out, t_share = server.conv2DSSFromTensors(got_cts, dev(share1) if shared else None, dev(filters), meta,
                                                (dev(u), dev(e1), dev(mask)), inline_verdict=route in ("1", "1p", "1n"))
    assert np.array_equal(host(t_out), ref_out) and np.array_equal(host(t_share), ref_share)
    assert np.array_equal(host(one_share), ref_share)
    # ... and so does the call on filters prepared once (static weights)
    prep = server.prepareFilters(dev(filters), meta)
    for _ in range(2):
        pre_out, pre_share = server.conv2DSSPrepared(got_cts, got_s, prep, meta, (dev(u), dev(e1), dev(mask)))
        assert np.array_equal(host(pre_out), ref_out)
        assert np.array_equal(host(pre_share), ref_share)
    pre_out, pre_share = server.conv2DSSPrepared(got_cts, None, prep, meta, (dev(u), dev(e1), dev(mask)),
                                                 share_tensor=dev(share1) if shared else None)
    assert np.array_equal(host(pre_out), ref_out) and np.array_equal(host(pre_share), ref_share)
    # ... and the online half alone, the encryptions of zero under the re-sharing made ahead of the input from the same u
    zeros = server.prepareZeros(meta, dev(u))
    on_out, on_share = server.conv2DSSOnline(got_cts, dev(share1) if shared else None, prep, zeros, meta,
                                             (dev(e1), dev(mask)))
    assert np.array_equal(host(on_out), ref_out) and np.array_equal(host(on_share), ref_share)
    # the USE_APPROX_RESHARE=0 build of the same call: flood noise, e0 kept, mask mod t; exact identity
    flood, e0 = splitmix64(SEED + 60, cv.n_out * N).reshape(cv.n_out, N), noise(SEED + 61, (cv.n_out, N))
    words = splitmix64(SEED + 62, cv.n_out * N).reshape(cv.n_out, N)
    x_out, x_share = cv.server(ref_cts, ref_s, ref_f, np.ascontiguousarray(pk[:, :1]), u, e1, words, exact=(flood, e0))
    y_out, y_share = server.conv2DSS(got_cts, got_s, got_f, meta, (dev(flood), dev(u), dev(e0), dev(e1), dev(words)),
                                     exact=True)
    assert np.array_equal(host(y_out), x_out) and np.array_equal(host(y_share), x_share)
    y_dec = host(client.decryptToTensor(y_out, meta))
    assert np.array_equal((y_dec.astype(object) + x_share.astype(object)) % T, ideal_conv(img, filters, same, stride, T))
    # client: decryptToTensor, then the reference's acceptance identity
    ref_dec = cv.decrypt(np.ascontiguousarray(sk[:1]), ref_out)
    got_dec = host(client.decryptToTensor(got_out, meta))
    assert np.array_equal(got_dec, ref_dec)
    # the compact wire form of the result: every bit the reference's ciphertext holds comes back, in fewer words
    packed = server.packResult(got_out, meta)
    assert np.array_equal(host(client.unpackResult(packed, meta)), ref_out)
    kept = int(plan_kept(cv, ref_out))
    assert packed.shape[1] < 2 * N
    if N == 4096 and T == 1 << 37:   # (the widths of BASELINE's configuration: 51 bits of c1, 39 of a kept coefficient of c0)
        assert packed.shape[1] <= (N * 51 + 63) // 64 + (kept * 39 + 63) // 64
    ideal = ideal_conv(img, filters, same, stride, T)
    diff = (got_dec.astype(object) + host(got_share).astype(object) - ideal) % T
    assert all(d in (0, 1, T - 1) for d in diff.ravel())
    if not (stride > 1 and fhw == (1, 1)):
        ideal_gpu = host(server.idealFunctionality(dev(img), dev(filters), meta))
        assert np.array_equal(ideal_gpu.astype(object), ideal)


def test_conv_exact_reshare_with_flood_noise(mm):
    """t = 2^44 is the one ring size of CheetahLinear (base_mod < 2^45) where flood_ciphertext's noise
    is not masked to zero by set_poly_coeffs_uniform's (1 << (bitlen % 64)) - 1 (bitlen = 63)."""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from tests.oracle_layers import flood_noise
    t = 1 << 44
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    ishape, fshape, M = (3, 9, 9), (3, 3, 3), 2
    cv, meta = ConvOracle(o, ishape, fshape, M, 1, 1, False), Meta(ishape, fshape, M, 1, 1, False)
    sk, pk = keypair(o)
    client, server = HomConv2DSS(), HomConv2DSS()
    client.setUp(g, sk_ntt=dev(sk))
    server.setUp(g, pk_ntt=dev(pk))
    img, filters = uniform_mod(SEED + 1, ishape, t), signed_weights(SEED + 2, (M,) + fshape, 1 << 12, t)
    a, e = rns_uniform(o, SEED + 3, cv.n_in, 2), noise(SEED + 4, (cv.n_in, N))
    cts, f = cv.encrypt_image(sk, img, a, e), cv.encode_filters(filters)
    u, e0, e1 = ternary(SEED + 5, (cv.n_out, N)), noise(SEED + 6, (cv.n_out, N)), noise(SEED + 7, (cv.n_out, N))
    flood = splitmix64(SEED + 8, cv.n_out * N).reshape(cv.n_out, N)
    words = splitmix64(SEED + 9, cv.n_out * N).reshape(cv.n_out, N)
    assert flood_noise(o, flood[0]).any()
    x_out, x_share = cv.server(cts, None, f, np.ascontiguousarray(pk[:, :1]), u, e1, words, exact=(flood, e0))
    y_out, y_share = server.conv2DSS(dev(cts), None, dev(f), meta,
                                     (dev(flood), dev(u), dev(e0), dev(e1), dev(words)), exact=True)
    assert np.array_equal(host(y_out), x_out) and np.array_equal(host(y_share), x_share)
    dec = host(client.decryptToTensor(y_out, meta))
    assert np.array_equal((dec.astype(object) + x_share.astype(object)) % t, ideal_conv(img, filters, 1, 1, t))


def _random_layers(n, seed):
    """small random layer geometries the reference accepts (fh <= fw, filter not larger than the image)"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        fw = int(rng.choice([1, 1, 2, 3, 3, 5]))
        fh = int(rng.integers(1, fw + 1))
        h, w = int(rng.integers(fh, 40)), int(rng.integers(fw, 40))
        c, m = int(rng.integers(1, 24)), int(rng.integers(1, 9))
        stride, same = int(rng.choice([1, 1, 2])), int(rng.integers(0, 2))
        if (h - fh) // stride < 0 or (w - fw) // stride < 0:
            continue
        out.append(((c, h, w), (fh, fw), m, same, stride, bool(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("ishape,fhw,M,same,stride,shared", _random_layers(24, 20260411))
def test_random_layers_both_routes(mm, monkeypatch, ishape, fhw, M, same, stride, shared):
    """random small geometries through the transform and the direct route against the oracle: output
    ciphertexts and share tensor bit for bit, from the one-call form"""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    o, g = OracleCtx(N, (60, 49), T), mm.Context(N, (60, 49), T)
    fshape = (ishape[0],) + tuple(fhw)
    try:
        cv = ConvOracle(o, ishape, fshape, M, same, stride, shared)
    except Exception:
        pytest.skip("geometry the reference rejects")
    meta = Meta(ishape, fshape, M, same, stride, shared)
    sk, pk = keypair(o)
    img, share0 = uniform_mod(SEED + 20, ishape, T), uniform_mod(SEED + 21, ishape, T)
    share1 = (img - share0) & np.uint64(T - 1)
    filters = signed_weights(SEED + 22, (M,) + fshape, 1 << 12, T)
    a, e = rns_uniform(o, SEED + 30, cv.n_in, 2), noise(SEED + 31, (cv.n_in, N))
    cts = cv.encrypt_image(sk, share0 if shared else img, a, e)
    u, e1 = ternary(SEED + 40, (cv.n_out, N)), noise(SEED + 41, (cv.n_out, N))
    mask = rns_uniform(o, SEED + 50, cv.n_out, 1)[:, 0]
    ref_out, ref_share = cv.server(cts, cv.encode_image(share1) if shared else None, cv.encode_filters(filters),
                                   np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    for route in ("0", "1"):
        monkeypatch.setenv("MMH_CONV_POINTWISE", route)
        server = HomConv2DSS()
        server.setUp(g, pk_ntt=dev(pk))
        got = server.conv2DSSFromTensors(dev(cts), dev(share1) if shared else None, dev(filters), meta,
                                         (dev(u), dev(e1), dev(mask)))
        assert np.array_equal(host(got[0]), ref_out), route
        assert np.array_equal(host(got[1]), ref_share), route


def test_conv_errors(mm):
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    import torch
    g = mm.Context(N, (60, 49), T)
    h = HomConv2DSS()
    h.setUp(g)
    with pytest.raises(mm.MMHError) as ei:   # filter larger than the polynomial
        h.plan(Meta((1, 100, 100), (1, 70, 70), 1, 0, 1))
    assert ei.value.code == 7
    with pytest.raises(mm.MMHError) as ei:   # fh > fw: FilterIndexer yields a negative index, the
        h.plan(Meta((8, 9, 11), (8, 3, 2), 3, 1, 1))   # reference LOG(FATAL)s in Encode (tensor_encoder.cc:268-270)
    assert ei.value.code == 6
    meta = Meta((3, 8, 8), (3, 3, 3), 2, 1, 1)
    with pytest.raises(mm.MMHError) as ei:   # image shape mismatch -> ERR_DIM_MISMATCH
        h.encodeImage(torch.zeros((3, 8, 9), dtype=torch.int64, device="cuda"), meta)
    assert ei.value.code == 3
    with pytest.raises(mm.MMHError) as ei:   # no public key -> ERR_CONFIG
        h.conv2DSS(None, None, None, meta, None)
    assert ei.value.code == 1
    # a filter that is all zero mod t -> ERR_INVALID_ARG (conv2DOneFilter, hom_conv2d_ss.cc:573-580)
    h.setUp(g, pk_ntt=torch.zeros((2, 2, N), dtype=torch.int64, device="cuda"))
    p = h.plan(meta)
    filters = torch.ones((2, 3, 3, 3), dtype=torch.int64, device="cuda")
    filters[1] = T   # zero mod t
    cts = torch.zeros((p.n_in, 2, 2, N), dtype=torch.int64, device="cuda")
    s1 = torch.zeros((p.n_in, N), dtype=torch.int64, device="cuda")
    z = torch.zeros((p.n_out, N), dtype=torch.int64, device="cuda")
    with pytest.raises(mm.MMHError) as ei:
        h.conv2DSSFromWeights(cts, s1, filters, meta, (z, z, z))
    assert ei.value.code == 8
    with pytest.raises(mm.MMHError) as ei:
        h.conv2DSS(cts, s1, h.encodeFilters(filters, meta), meta, (z, z, z))
    assert ei.value.code == 8
    # the one-call form evaluates the rule itself: on the direct route from the weight pass, on the transform
    # route and for unshared inputs by a pass of its own; good filters must not trip it
    share = torch.zeros((3, 8, 8), dtype=torch.int64, device="cuda")
    good = torch.ones((2, 3, 3, 3), dtype=torch.int64, device="cuda")
    for route in ("0", "1"):
        os.environ["MMH_CONV_POINTWISE"] = route
        try:
            for shared in (True, False):
                h2 = HomConv2DSS()
                h2.setUp(g, pk_ntt=torch.zeros((2, 2, N), dtype=torch.int64, device="cuda"))
                m2 = Meta((3, 8, 8), (3, 3, 3), 2, 1, 1, shared)
                for inline in (False, True):
                    with pytest.raises(mm.MMHError) as ei:
                        h2.conv2DSSFromTensors(cts, share if shared else None, filters, m2, (z, z, z), inline)
                    assert ei.value.code == 8
                    h2.conv2DSSFromTensors(cts, share if shared else None, good, m2, (z, z, z), inline)
        finally:
            del os.environ["MMH_CONV_POINTWISE"]


def test_sharded_conv_single_rank_equals_unsharded(mm):
    """world_size 1 on the GPU: ShardedConv must be the identity wrapper around conv2DSS"""
    import os
    import torch
    import torch.distributed as dist
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from openmm_amd.sharding import ShardedConv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = mm.Context(N, (60, 49), T)
        smp = mm.Sampler(g, key=bytes(range(32)))
        sk, pk = smp.keygen()
        meta = Meta((4, 10, 10), (4, 3, 3), 5, 1, 1, True)
        srv = HomConv2DSS()
        srv.setUp(g, pk_ntt=pk)
        plan = srv.plan(meta)
        cts = smp.uniform(2 * plan.n_in).view(plan.n_in, 2, 2, N)
        share1 = torch.randint(0, T, (4, 10, 10), dtype=torch.int64, device="cuda")
        filters = torch.randint(1, 1 << 12, (5, 4, 3, 3), dtype=torch.int64, device="cuda")
        rand = (smp.ternary(plan.n_out), smp.noise(plan.n_out), smp.uniform(plan.n_out, lv=1)[:, 0].contiguous())
        s1 = srv.encodeImage(share1, meta)
        ref = srv.conv2DSS(cts, s1, srv.encodeFilters(filters, meta), meta, rand)
        for mode in ("gather", "d2h"):
            got = ShardedConv(srv, meta, mode=mode).run(cts, share1, filters, lambda n: tuple(r[:n] for r in rand))
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    finally:
        dist.destroy_process_group()


def test_layer_calls_on_two_streams_share_the_scratch_cache(mm):
    """layer calls are re-entrant: scratch blocks are cached per stream, so calls enqueued on two streams
    must not hand each other's live buffers around; results equal the single-stream ones"""
    import torch
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    g = mm.Context(N, (60, 49), T)
    smp = mm.Sampler(g, key=bytes(range(32)))
    _, pk = smp.keygen()
    srv = HomConv2DSS()
    srv.setUp(g, pk_ntt=pk)
    metas = [Meta((16, 14, 14), (16, 3, 3), 8, 1, 1, True), Meta((12, 28, 28), (12, 1, 1), 6, 0, 1, True)]
    jobs = []
    for k, meta in enumerate(metas):
        plan = srv.plan(meta)
        gen = torch.Generator(device="cuda").manual_seed(SEED + k)
        cts = smp.uniform(2 * plan.n_in).view(plan.n_in, 2, 2, N)
        share = torch.randint(0, T, meta.ishape, dtype=torch.int64, device="cuda", generator=gen)
        filters = torch.randint(1, 1 << 12, (meta.n_filters,) + meta.fshape, dtype=torch.int64, device="cuda",
                                generator=gen)
        rand = (smp.ternary(plan.n_out), smp.noise(plan.n_out), smp.uniform(plan.n_out, lv=1)[:, 0].contiguous())
        jobs.append((meta, cts, share, filters, rand))
    run = lambda j: srv.conv2DSSFromWeights(j[1], srv.encodeImage(j[2], j[0]), j[3], j[0], j[4])
    ref = [run(j) for j in jobs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None, None]
    for rep in range(3):
        for k, j in enumerate(jobs):
            streams[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[k]):
                got[k] = run(j)
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(got[k][0], ref[k][0]) and torch.equal(got[k][1], ref[k][1])
    assert g.lib.mmh_scratch_release() == 0


@pytest.mark.parametrize("tail_split_max", [None, "0"])
@pytest.mark.parametrize("route", ["model", "0", "1", "1p", "1n"])
@pytest.mark.parametrize("ishape,fhw,M,same,stride,shared", CASES[:6] + [((40, 14, 14), (1, 1), 70, 0, 1, True)])
def test_conv_call_that_draws_its_own_randomness(mm, monkeypatch, ishape, fhw, M, same, stride, shared, route,
                                                 tail_split_max, n=None):
    """mmh_conv_server_tensors_sampled (conv2DSS draws u, e1 and the mask inside the call, hom_conv2d_ss.cc:302,:329,
    :206): the same ciphertexts and share as the call on the arrays mmh_sample_reshare makes from the same key and
    streams -- which test_conv_layer_bit_exact pins against the oracle -- on every route, with the tail in its split
    form (the front launch finishes the encryption of zero) and in its fused form (it leaves u transformed)."""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from openmm_amd.lib import Sampler
    if route != "model":
        monkeypatch.setenv("MMH_CONV_POINTWISE", route[0])
    for var, val in (("MMH_PW_PAIRED", {"1": "0", "1p": "1"}.get(route)),
                     ("MMH_PW_NARROW", {"1": "0", "1p": "0", "1n": "1"}.get(route)), ("MMH_TAIL_SPLIT_MAX", tail_split_max)):
        if val is None:
            monkeypatch.delenv(var, raising=False)
        else:
            monkeypatch.setenv(var, val)
    n = n or N
    o, g = OracleCtx(n, (60, 49), T), mm.Context(n, (60, 49), T)
    fshape = (ishape[0],) + tuple(fhw)
    meta = Meta(ishape, fshape, M, same, stride, shared)
    _, pk = keypair(o)
    server = HomConv2DSS()
    server.setUp(g, pk_ntt=dev(pk))
    plan = server.plan(meta)
    cts = dev(np.stack([rns_uniform(o, SEED + 70 + k, plan.n_in, 2) for k in range(2)], axis=1))   # [n_in][2][L][N]
    share = dev(uniform_mod(SEED + 75, ishape, T)) if shared else None
    filters = dev(signed_weights(SEED + 72, (M,) + fshape, 1 << 12, T))
    key = bytes(range(7, 39))
    a = Sampler(g, key)
    a.words(5)                                # (streams already spent: the call must take the sampler's NEXT three)
    ref_out, ref_share = server.conv2DSSFromTensors(cts, share, filters, meta, a.reshare(plan.n_out))
    b = Sampler(g, key)
    b.words(5)
    got_out, got_share = server.conv2DSSFromTensors(cts, share, filters, meta, b)
    assert np.array_equal(host(got_out), host(ref_out)) and np.array_equal(host(got_share), host(ref_share))
    assert b.next_stream == a.next_stream
    if route == "1n":   # the encryptions of zero made in the front launch / riding in the narrow kernel's, whatever its grid
        for ride in ("0", "1"):
            monkeypatch.setenv("MMH_ZERO_RIDE", ride)
            c = Sampler(g, key)
            c.words(5)
            got_out, got_share = server.conv2DSSFromTensors(cts, share, filters, meta, c)
            assert np.array_equal(host(got_out), host(ref_out)) and np.array_equal(host(got_share), host(ref_share)), ride
        monkeypatch.delenv("MMH_ZERO_RIDE")
    # deferred verdict form
    got_out, got_share = server.conv2DSSFromTensors(cts, share, filters, meta, Sampler(g, key), defer_verdict=True)
    import torch
    torch.cuda.synchronize()
    server.raise_pending()


def test_conv_call_that_draws_its_own_randomness_n8192(mm, monkeypatch):
    test_conv_call_that_draws_its_own_randomness(mm, monkeypatch, (9, 20, 20), (3, 3), 5, 1, 1, True, "1", None, n=8192)
    test_conv_call_that_draws_its_own_randomness(mm, monkeypatch, (40, 14, 14), (1, 1), 70, 0, 1, True, "model", "0", n=8192)
