"""Bit-exact GPU-vs-oracle parity of HomConv2DSS at the sizes BASELINE.json names.

* configs[2]: the ResNet50 3x3 layer 64x56x56 -> 64 filters, SAME, stride 1, shared input, l = 37 at FULL size
  (64 input cts, 4096 filter polynomials, 64 output cts) through the plan's own route, the forced transform route
  and the forced direct (i8 matrix-core) route, from the raw tensors and from prepared filters: every output
  ciphertext coefficient and every share entry against the oracle's per-term multiply_plain / add_inplace schedule
  (hom_conv2d_ss.cc:563-712), then decrypt + the reference's acceptance identity.
* configs[3-4]: every distinct convolution shape of ResNet50 (networks/main_resnet50.cpp:4981-6333),
  SqueezeNet (main_sqnet.cpp:3107-3764) and the shape classes of DenseNet121 at their real channel counts and
  geometry, with the filter count cut down so that the oracle's one-ciphertext-at-a-time loop stays in seconds;
  plus one layer with 256 filters (two 128-filter tiles of the product kernel) at full K.
The oracle result is computed once per shape and compared with both routes.
"""
import numpy as np
import pytest

from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary
from tests.oracle_binding import OracleCtx, uniform_mod
from tests.oracle_layers import ConvOracle, ideal_conv

pytestmark = pytest.mark.gpu
T = 1 << 37
N = 4096


def dev(a):
    import torch
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def oracle_layer(ishape, fhw, M, same, stride, shared, wbound=1 << 12):
    """inputs and the oracle's outputs for one layer"""
    o = OracleCtx(N, (60, 49), T)
    fshape = (ishape[0],) + tuple(fhw)
    cv = ConvOracle(o, ishape, fshape, M, same, stride, shared)
    sk, pk = keypair(o)
    img = uniform_mod(SEED + 20, ishape, T)
    share0 = uniform_mod(SEED + 21, ishape, T)
    share1 = (img - share0) & np.uint64(T - 1)
    filters = signed_weights(SEED + 22, (M,) + fshape, wbound, T)
    a, e = rns_uniform(o, SEED + 30, cv.n_in, 2), noise(SEED + 31, (cv.n_in, N))
    cts = cv.encrypt_image(sk, share0 if shared else img, a, e)
    u, e1 = ternary(SEED + 40, (cv.n_out, N)), noise(SEED + 41, (cv.n_out, N))
    mask = rns_uniform(o, SEED + 50, cv.n_out, 1)[:, 0]
    ref_out, ref_share = cv.server(cts, cv.encode_image(share1) if shared else None, cv.encode_filters(filters),
                                   np.ascontiguousarray(pk[:, :1]), u, e1, mask)
    return dict(o=o, cv=cv, sk=sk, pk=pk, img=img, share1=share1, filters=filters, cts=cts, u=u, e1=e1, mask=mask,
                ref_out=ref_out, ref_share=ref_share, fshape=fshape)


def check_routes(monkeypatch, L, ishape, M, same, stride, shared, routes=("0", "1", "1p", "1n", "model"), prepared=True, decrypt=False):
    import openmm_amd
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    g = openmm_amd.Context(N, (60, 49), T)
    meta = Meta(ishape, L["fshape"], M, same, stride, shared)
    d_cts, d_share, d_filters = dev(L["cts"]), dev(L["share1"]) if shared else None, dev(L["filters"])
    rand = (dev(L["u"]), dev(L["e1"]), dev(L["mask"]))
    for route in routes:
        if route == "model":
            monkeypatch.delenv("MMH_CONV_POINTWISE", raising=False)
        else:
            monkeypatch.setenv("MMH_CONV_POINTWISE", route[0])     # read when the plan is made
        # the direct route's product items may own both limbs of their columns (conv.hip, PwItems::paired); the
        # library pairs a layer when it has enough items, "1p" / "1" force every layer / none (read on every call)
        # layers of at most 32 filters build their digits inside the product kernel by default ("1n"); "1" / "1p" keep
        # them on the digit image
        for var, val in (("MMH_PW_PAIRED", {"1": "0", "1p": "1"}.get(route)), ("MMH_PW_NARROW", {"1": "0", "1p": "0", "1n": "1"}.get(route))):
            if val is None:
                monkeypatch.delenv(var, raising=False)
            else:
                monkeypatch.setenv(var, val)
        srv = HomConv2DSS()
        srv.setUp(g, pk_ntt=dev(L["pk"]))
        # the re-sharing tail has two forms picked by the number of output ciphertexts (bfv.hip): both, and the default
        for split_max in ("0", "1000000", None):
            if split_max is None:
                monkeypatch.delenv("MMH_TAIL_SPLIT_MAX", raising=False)
            else:
                monkeypatch.setenv("MMH_TAIL_SPLIT_MAX", split_max)     # read on every call
            out, share = srv.conv2DSSFromTensors(d_cts, d_share, d_filters, meta, rand)
            assert np.array_equal(host(out), L["ref_out"]), f"route {route}, tail {split_max}: output ciphertexts"
            assert np.array_equal(host(share), L["ref_share"]), f"route {route}, tail {split_max}: share"
        if prepared:
            prep = srv.prepareFilters(d_filters, meta)
            s1 = srv.encodeImage(d_share, meta) if shared else None
            out, share = srv.conv2DSSPrepared(d_cts, s1, prep, meta, rand)
            assert np.array_equal(host(out), L["ref_out"]), f"route {route}: prepared, output ciphertexts"
            assert np.array_equal(host(share), L["ref_share"]), f"route {route}: prepared, share"
            zeros = srv.prepareZeros(meta, rand[0])     # the online half alone (include/mm_hip.h)
            out, share = srv.conv2DSSOnline(d_cts, d_share, prep, zeros, meta, rand[1:])
            assert np.array_equal(host(out), L["ref_out"]), f"route {route}: online, output ciphertexts"
            assert np.array_equal(host(share), L["ref_share"]), f"route {route}: online, share"
    if decrypt:
        cli = HomConv2DSS()
        cli.setUp(g, sk_ntt=dev(L["sk"]))
        dec = host(cli.decryptToTensor(out, meta))
        assert np.array_equal(dec, L["cv"].decrypt(np.ascontiguousarray(L["sk"][:1]), L["ref_out"]))
        ideal = ideal_conv(L["img"], L["filters"], same, stride, T)
        diff = (dec.astype(object) + host(share).astype(object) - ideal) % T
        assert all(d in (0, 1, T - 1) for d in diff.ravel())     # library_fixed_uniform_cheetah.cpp:199


def test_headline_layer_full_size_bit_exact(monkeypatch):
    """BASELINE.json configs[2]: 64x56x56, 3x3 SAME, 64 filters: 576 taps = 9 steps of 64 x 64 filters on the direct
    route, 4096 filter transforms on the transform route"""
    ishape, M = (64, 56, 56), 64
    L = oracle_layer(ishape, (3, 3), M, 1, 1, True)
    assert (L["cv"].n_in, L["cv"].n_out) == (64, 64)
    check_routes(monkeypatch, L, ishape, M, 1, 1, True, routes=("model", "0", "1", "1p", "1n"), decrypt=True)


def test_two_filter_tiles_full_k(monkeypatch):
    """1024 -> 2048 stride-2 1x1 layer of ResNet50 with 256 of its filters: two 128-filter tiles, 1024 taps"""
    ishape, M = (1024, 14, 14), 256
    L = oracle_layer(ishape, (1, 1), M, 0, 2, True)
    check_routes(monkeypatch, L, ishape, M, 0, 2, True, prepared=False)


# (C, H, W, fh, fw, M, SAME?, stride): ResNet50 (bench.py RESNET50_CONVS)
RESNET50 = [
    (3, 230, 230, 7, 7, 64, 0, 2), (64, 56, 56, 1, 1, 256, 0, 1), (64, 56, 56, 1, 1, 64, 0, 1),
    (256, 56, 56, 1, 1, 64, 0, 1), (256, 56, 56, 1, 1, 512, 0, 2), (256, 56, 56, 1, 1, 128, 0, 1),
    (128, 58, 58, 3, 3, 128, 0, 2), (128, 28, 28, 1, 1, 512, 0, 1), (512, 28, 28, 1, 1, 128, 0, 1),
    (128, 28, 28, 3, 3, 128, 1, 1), (512, 28, 28, 1, 1, 1024, 0, 2), (512, 28, 28, 1, 1, 256, 0, 1),
    (256, 30, 30, 3, 3, 256, 0, 2), (256, 14, 14, 1, 1, 1024, 0, 1), (1024, 14, 14, 1, 1, 256, 0, 1),
    (256, 14, 14, 3, 3, 256, 1, 1), (1024, 14, 14, 1, 1, 2048, 0, 2), (1024, 14, 14, 1, 1, 512, 0, 1),
    (512, 16, 16, 3, 3, 512, 0, 2), (512, 7, 7, 1, 1, 2048, 0, 1), (2048, 7, 7, 1, 1, 512, 0, 1),
    (512, 7, 7, 3, 3, 512, 1, 1),
]
# DenseNet121 shape classes (networks/main_densenet121.cpp; bench.py densenet121_convs): stem, the 1x1 bottlenecks at
# the first / a middle / the last channel count of each block, the 3x3 growth layers, transitions, classifier
DENSENET121 = [
    (3, 224, 224, 7, 7, 64, 1, 2), (64, 56, 56, 1, 1, 128, 0, 1), (224, 56, 56, 1, 1, 128, 0, 1),
    (128, 56, 56, 3, 3, 32, 1, 1), (256, 56, 56, 1, 1, 128, 0, 1), (480, 28, 28, 1, 1, 128, 0, 1),
    (128, 28, 28, 3, 3, 32, 1, 1), (512, 28, 28, 1, 1, 256, 0, 1), (992, 14, 14, 1, 1, 128, 0, 1),
    (128, 14, 14, 3, 3, 32, 1, 1), (1024, 14, 14, 1, 1, 512, 0, 1), (992, 7, 7, 1, 1, 128, 0, 1),
    (128, 7, 7, 3, 3, 32, 1, 1), (1024, 1, 1, 1, 1, 1000, 0, 1),
]
SQUEEZENET = [
    (3, 227, 227, 3, 3, 64, 0, 2), (64, 56, 56, 1, 1, 16, 0, 1), (16, 56, 56, 1, 1, 64, 0, 1),
    (16, 56, 56, 3, 3, 64, 1, 1), (128, 56, 56, 1, 1, 16, 0, 1), (128, 27, 27, 1, 1, 32, 0, 1),
    (32, 27, 27, 1, 1, 128, 0, 1), (32, 27, 27, 3, 3, 128, 1, 1), (256, 27, 27, 1, 1, 32, 0, 1),
    (256, 13, 13, 1, 1, 48, 0, 1), (48, 13, 13, 1, 1, 192, 0, 1), (48, 13, 13, 3, 3, 192, 1, 1),
    (384, 13, 13, 1, 1, 48, 0, 1), (384, 13, 13, 1, 1, 64, 0, 1), (64, 13, 13, 1, 1, 256, 0, 1),
    (64, 13, 13, 3, 3, 256, 1, 1), (512, 13, 13, 1, 1, 64, 0, 1), (512, 13, 13, 1, 1, 1000, 0, 1),
]


def _ids(rows):
    return [f"{c}x{h}x{w}-{fh}x{fw}-{m}-{'same' if s else 'valid'}-s{st}" for c, h, w, fh, fw, m, s, st in rows]


@pytest.mark.parametrize("c,h,w,fh,fw,m,same,stride", RESNET50 + DENSENET121 + SQUEEZENET,
                         ids=_ids(RESNET50 + DENSENET121 + SQUEEZENET))
def test_network_shape_classes_bit_exact(monkeypatch, c, h, w, fh, fw, m, same, stride):
    """real geometry and channel counts, filter count cut to keep the oracle's per-term loop short"""
    ishape, shared = (c, h, w), c != 3        # first convolutions run with kIsSharedInput = false
    probe = ConvOracle(OracleCtx(N, (60, 49), T), ishape, (c, fh, fw), 1, same, stride, shared)
    M = max(2, min(m, 8, 1200 // (probe.n_cs * probe.n_sp)))
    L = oracle_layer(ishape, (fh, fw), M, same, stride, shared)
    check_routes(monkeypatch, L, ishape, M, same, stride, shared, prepared=(c * fh * fw <= 1024))
