"""CPU: the oracle's HomConv flow satisfies the reference's acceptance identity
(share0 + share1 = idealFunctionality within +-1 under USE_APPROX_RESHARE,
library_fixed_uniform_cheetah.cpp:199) on several geometries, including the quirky ones."""
import numpy as np
import pytest

from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary
from tests.oracle_binding import OracleCtx, uniform_mod
from tests.oracle_layers import ConvOracle, ideal_conv

T = 1 << 37
CASES = [
    # ishape, fshape(h,w), M, same, stride
    ((3, 8, 8), (3, 3), 2, 1, 1),
    ((2, 70, 70), (3, 3), 1, 1, 1),      # H*W > N: spatial slicing with overlap
    ((5, 12, 12), (1, 1), 3, 0, 2),      # 1x1 stride 2: compressed input (shape_inference.cc:99-104)
    ((4, 17, 17), (3, 3), 2, 0, 2),      # strided, VALID: keep-set is a superset of the targets
    ((6, 30, 30), (5, 5), 2, 1, 1),      # several channels per polynomial
]


def run_case(ishape, fhw, M, same, stride, shared=True):
    o = OracleCtx(4096, (60, 49), T)
    fshape = (ishape[0],) + tuple(fhw)
    cv = ConvOracle(o, ishape, fshape, M, same, stride, shared)
    sk, pk = keypair(o)
    img = uniform_mod(SEED + 20, ishape, T)
    share0 = uniform_mod(SEED + 21, ishape, T)                      # client share
    share1 = (img - share0) & np.uint64(T - 1) if shared else None  # server share
    filters = signed_weights(SEED + 22, (M,) + fshape, 1 << 12, T)
    cts = cv.encrypt_image(sk, share0 if shared else img, rns_uniform(o, SEED + 30, cv.n_in, 2), noise(SEED + 31, (cv.n_in, 4096)))
    spts = cv.encode_image(share1) if shared else None
    fpts = cv.encode_filters(filters)
    pk1 = np.ascontiguousarray(pk[:, :1])
    u, e1 = ternary(SEED + 40, (cv.n_out, 4096)), noise(SEED + 41, (cv.n_out, 4096))
    mask = rns_uniform(o, SEED + 50, cv.n_out, 1)[:, 0]
    out_cts, out_share = cv.server(cts, spts, fpts, pk1, u, e1, mask)
    dec = cv.decrypt(np.ascontiguousarray(sk[:1]), out_cts)
    ideal = ideal_conv(img, filters, same, stride, T)
    return cv, dec, out_share, ideal


@pytest.mark.parametrize("case", CASES)
def test_conv_flow_reconstructs(case):
    cv, dec, share, ideal = run_case(*case)
    assert dec.shape == ideal.shape == cv.out_shape
    diff = (dec.astype(object) + share.astype(object) - ideal) % T
    assert all(d in (0, 1, T - 1) for d in diff.ravel()), sorted(set(diff.ravel()))[:10]


def test_conv_not_shared_input():
    cv, dec, share, ideal = run_case((3, 8, 8), (3, 3), 2, 1, 1, shared=False)
    diff = (dec.astype(object) + share.astype(object) - ideal) % T
    assert all(d in (0, 1, T - 1) for d in diff.ravel())


def test_resnet_layer_geometry():
    """56x56x64, 3x3, SAME: strides {1,58,58}, 64 input cts, 64x64 filter polys with 9 non-zeros"""
    o = OracleCtx(4096, (60, 49), T)
    cv = ConvOracle(o, (64, 56, 56), (64, 3, 3), 64, 1, 1, True)
    assert cv.slice_strides == (1, 58, 58) and cv.pads == (2, 2)
    assert (cv.n_in, cv.n_cs, cv.n_sp, cv.n_out) == (64, 64, 1, 64)
    (th, tw), tg = cv.targets(cv.mask_ix, 0, 0)
    assert (th, tw) == (56, 56) and len(tg) == 3136
    f = cv.encode_filters(signed_weights(SEED, (1, 64, 3, 3), 1 << 12, T)[:1].repeat(64, 0)[:1].repeat(64, 0))
    assert int((f[0, 0] != 0).sum()) <= 9


@pytest.mark.parametrize("logt", [37, 44])
def test_conv_exact_reshare_flow_reconstructs(logt):
    """USE_APPROX_RESHARE=0 build: sub_plain re-sharing, exact reconstruction (no +-1).  At t = 2^44
    the flood noise is 63 bits wide; at the shipped 2^37 the reference's own mask makes it zero."""
    from tests.oracle_binding import splitmix64
    from tests.oracle_layers import flood_noise
    N, t = 4096, 1 << logt
    o = OracleCtx(N, (60, 49), t)
    ishape, fshape, M = (3, 9, 9), (3, 3, 3), 2
    cv = ConvOracle(o, ishape, fshape, M, 1, 1, False)
    sk, pk = keypair(o)
    img, filters = uniform_mod(SEED + 1, ishape, t), signed_weights(SEED + 2, (M,) + fshape, 1 << 12, t)
    cts = cv.encrypt_image(sk, img, rns_uniform(o, SEED + 3, cv.n_in, 2), noise(SEED + 4, (cv.n_in, N)))
    u, e0, e1 = ternary(SEED + 5, (cv.n_out, N)), noise(SEED + 6, (cv.n_out, N)), noise(SEED + 7, (cv.n_out, N))
    flood = splitmix64(SEED + 8, cv.n_out * N).reshape(cv.n_out, N)
    words = splitmix64(SEED + 9, cv.n_out * N).reshape(cv.n_out, N)
    assert bool(flood_noise(o, flood[0]).any()) == (logt == 44)
    out, share = cv.server(cts, None, cv.encode_filters(filters), np.ascontiguousarray(pk[:, :1]), u, e1, words,
                           exact=(flood, e0))
    dec = cv.decrypt(np.ascontiguousarray(sk[:1]), out)
    assert np.array_equal((dec.astype(object) + share.astype(object)) % t, ideal_conv(img, filters, 1, 1, t))
