"""GPU parity of the BFV batch entry points (C ABI) against the oracle, bit for bit, on
identical injected randomness."""
import numpy as np
import pytest

from tests.oracle_binding import A, OracleCtx, splitmix64, uniform_mod

pytestmark = pytest.mark.gpu
SEED = 0x4D264D
P32 = 4293918721
N = 4096
TS = [1 << 37, P32, 1 << 32, 2198100901889]


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def dev(a, dtype=np.uint64):
    import torch
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def keys(o, seed):
    s = (splitmix64(seed, N) % np.uint64(3)).astype(np.int64) - 1
    sk = np.stack([np.where(s < 0, np.uint64(q - 1), s.astype(np.uint64)) for q in o.q]).astype(np.uint64)
    return o.ntt_forward(sk[None])[0]


def noise(seed, shape):
    r = splitmix64(seed, int(np.prod(shape)))
    return ((r & np.uint64(7)).astype(np.int64) - (r >> np.uint64(61)).astype(np.int64)).reshape(shape)


def ternary(seed, shape):
    return ((splitmix64(seed, int(np.prod(shape))) % np.uint64(3)).astype(np.int64) - 1).reshape(shape)


def rns_uniform(o, seed, n, limbs):
    return np.stack([np.stack([uniform_mod(seed + 13 * k + l, N, o.q[l]) for l in range(limbs)])
                     for k in range(n)])


def fresh_cts(o, sk, n, t, seed):
    """oracle-side symmetric encryptions of seeded plaintexts"""
    a = rns_uniform(o, seed, n, o.L)
    e = noise(seed + 1, (n, N))
    m = uniform_mod(seed + 2, (n, N), t)
    cts = np.zeros((n, 2, o.L, N), np.uint64)
    for k in range(n):
        o.lib.mmo_encrypt_symmetric(o.p, 0, A(sk), A(a[k]), A(e[k]), A(m[k]), N, A(cts[k]))
    return a, e, m, cts


@pytest.mark.parametrize("t", TS)
def test_encrypt_symmetric_and_decrypt(mm, t):
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    assert g.gamma == o.gamma
    sk = keys(o, SEED)
    a, e, m, cts = fresh_cts(o, sk, 3, t, SEED + 5)
    got = g.encrypt_symmetric(dev(sk), dev(a), dev(e), dev(m))
    assert np.array_equal(host(got), cts)
    assert np.array_equal(host(g.decrypt(dev(sk), got)), m)
    # encrypt_zero_symmetric (pts = NULL)
    z = g.encrypt_symmetric(dev(sk), dev(a), dev(e), None)
    ref = np.zeros_like(cts)
    for k in range(3):
        o.lib.mmo_encrypt_symmetric(o.p, 0, A(sk), A(a[k]), A(e[k]), None, 0, A(ref[k]))
    assert np.array_equal(host(z), ref)


@pytest.mark.parametrize("t", TS)
def test_add_sub_plain_both_levels(mm, t):
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    for lv in (0, 1):
        limbs = o.limbs(lv)
        cts = np.stack([rns_uniform(o, SEED + 3 * c, 2, limbs) for c in range(4)])  # [4,2,limbs,N]
        m = uniform_mod(SEED + 77, (4, N), t)
        m[0, :5] = [0, 1, t - 1, t // 2, (t + 1) // 2]
        for sub in (False, True):
            ref = cts.copy()
            for k in range(4):
                (o.lib.mmo_sub_plain if sub else o.lib.mmo_add_plain)(o.p, lv, A(ref[k]), A(m[k]), N)
            got = g.add_plain(dev(cts), dev(m), lv=lv, sub=sub)
            assert np.array_equal(host(got), ref), (lv, sub)


@pytest.mark.parametrize("t", [1 << 37, P32])
def test_mul_plain_acc_matches_reference_schedule(mm, t):
    """conv-shaped MAC: out[m][i] = sum_c ct[c*S+i] * pt[m*C+c], vs per-term multiply_plain+add"""
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    sk = keys(o, SEED)
    C_, S_, M_ = 3, 2, 5
    _, _, _, cts = fresh_cts(o, sk, C_ * S_, t, SEED + 9)
    pts = np.zeros((M_ * C_, N), np.uint64)
    for k in range(M_ * C_):  # sparse signed weights like an encoded 3x3 filter
        idx = splitmix64(SEED + k, 9) % np.uint64(N)
        pts[k, idx] = uniform_mod(SEED + 100 + k, 9, t)
    ref = np.zeros((M_ * S_, 2, o.L, N), np.uint64)
    tmp = np.zeros((2, o.L, N), np.uint64)
    for m in range(M_):
        for i in range(S_):
            for c in range(C_):
                o.lib.mmo_multiply_plain(o.p, 0, A(cts[c * S_ + i]), A(pts[m * C_ + c]), N, A(tmp))
                if c == 0:
                    ref[m * S_ + i] = tmp
                else:
                    o.lib.mmo_ct_add(o.p, 0, A(ref[m * S_ + i]), A(tmp), 2)
    got = g.mul_plain_acc(dev(cts), dev(pts), M_, S_, C_, S_, 1, C_, 1, 0)
    assert np.array_equal(host(got), ref)
    # BOLE-shaped: out[i] = ex[i]*py[i] + ey[i]*px[i]  (hom_bn_ss.cc:304-306)
    n = 3
    got = g.mul_plain_acc(dev(cts), dev(pts[:2 * n]), 1, n, 2, n, 1, 0, n, 1)
    for i in range(n):
        o.lib.mmo_multiply_plain(o.p, 0, A(cts[i]), A(pts[i]), N, A(ref[0]))
        o.lib.mmo_multiply_plain(o.p, 0, A(cts[n + i]), A(pts[n + i]), N, A(tmp))
        o.lib.mmo_ct_add(o.p, 0, A(ref[0]), A(tmp), 2)
        assert np.array_equal(host(got[i]), ref[0])


def test_mac_accumulates_many_terms(mm):
    """K = 70 crosses the 8-term lazy-reduction boundary many times; operands near q stress the 128-bit sums"""
    t = 1 << 37
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    K = 70
    cts = np.zeros((K, 2, 2, N), np.uint64)
    pts = np.zeros((K, 2, N), np.uint64)
    for l, q in enumerate(o.q):
        cts[:, :, l] = q - 1 - (splitmix64(SEED + l, K * 2 * N) % np.uint64(1000)).reshape(K, 2, N)
        pts[:, l] = q - 1 - (splitmix64(SEED + 5 + l, K * N) % np.uint64(1000)).reshape(K, N)
    out = mm.lib.to_device(np.zeros((1, 2, 2, N), np.uint64))
    from openmm_amd.lib import check, dptr, _stream_ptr
    d_cts, d_pts = dev(cts), dev(pts)  # keep the tensors alive across the call
    check(g.lib.mmh_dyadic_mac(g.handle, 0, dptr(d_cts), K, dptr(d_pts), K, 1, 1, K, 1, 0, 0, 1, 0,
                               dptr(out), _stream_ptr()))
    ref = np.zeros((2, 2, N), dtype=object)
    for l, q in enumerate(o.q):
        for c in range(2):
            acc = np.zeros(N, dtype=object)
            for k in range(K):
                acc = acc + cts[k, c, l].astype(object) * pts[k, l].astype(object)
            ref[c, l] = acc % q
    assert np.array_equal(host(out)[0].astype(object), ref)


@pytest.mark.parametrize("t", [1 << 37, P32])
def test_server_tail_modswitch_enczero_mask_truncate(mm, t):
    """flood_ciphertext + sample_random_mask (approx) + sub_poly + truncate + remove_unused"""
    import torch
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    sk = keys(o, SEED)
    n = 3
    _, _, m, cts = fresh_cts(o, sk, n, t, SEED + 21)
    # public key from injected randomness
    a_ntt = rns_uniform(o, SEED + 30, 1, 2)[0]
    e_pk = noise(SEED + 31, N)
    pk = np.zeros((2, 2, N), np.uint64)
    o.lib.mmo_make_public_key(o.p, A(sk), A(a_ntt), A(e_pk), A(pk))
    assert np.array_equal(host(g.make_public_key(dev(sk), dev(a_ntt), dev(e_pk))), pk)
    # mod switch
    low = np.zeros((n, 2, 1, N), np.uint64)
    for k in range(n):
        o.lib.mmo_mod_switch_to_last(o.p, A(cts[k]), A(low[k]), 2)
    glow = g.mod_switch_to_last(dev(cts))
    assert np.array_equal(host(glow), low)
    # asymmetric encryption of zero at the last level, e0 skipped (USE_APPROX_RESHARE)
    pk1 = np.ascontiguousarray(pk[:, :1])
    u, e1 = ternary(SEED + 40, (n, N)), noise(SEED + 41, (n, N))
    zero = np.zeros((n, 2, 1, N), np.uint64)
    for k in range(n):
        o.lib.mmo_encrypt_zero_asymmetric(o.p, 1, A(pk1), A(u[k]), None, A(e1[k]), A(zero[k]))
    gzero = g.encrypt_zero_asym(dev(pk1), dev(u), None, dev(e1), lv=1)
    assert np.array_equal(host(gzero), zero)
    # first level with both noises (exact-reshare build)
    e0 = noise(SEED + 42, (n, N))
    z0 = np.zeros((n, 2, 2, N), np.uint64)
    for k in range(n):
        o.lib.mmo_encrypt_zero_asymmetric(o.p, 0, A(pk), A(u[k]), A(e0[k]), A(e1[k]), A(z0[k]))
    assert np.array_equal(host(g.encrypt_zero_asym(dev(pk), dev(u), dev(e0), dev(e1), lv=0)), z0)
    # add, subtract the RNS mask, share = scale_and_round at the targets
    for k in range(n):
        o.lib.mmo_ct_add(o.p, 1, A(low[k]), A(zero[k]), 2)
    g.ct_add(glow, gzero, lv=1)
    r = rns_uniform(o, SEED + 50, n, 1)
    targets = np.sort(splitmix64(SEED + 51, 300) % np.uint64(N)).astype(np.uint32)
    targets = np.unique(targets)
    share = np.zeros((n, len(targets)), np.uint64)
    for k in range(n):
        low[k, 0, 0] = np.where(low[k, 0, 0] >= r[k, 0], low[k, 0, 0] - r[k, 0],
                                low[k, 0, 0] + np.uint64(o.q[0]) - r[k, 0])
        gathered = np.ascontiguousarray(r[k, 0, targets])
        o.lib.mmo_decrypt_scale_and_round(o.p, 1, A(gathered), len(targets), A(share[k]))
    g.sub_rns_poly(glow, dev(r), lv=1)
    assert np.array_equal(host(glow), low)
    gshare = g.scale_and_round(dev(r), torch.from_numpy(targets.astype(np.int32)).cuda(), lv=1)
    assert np.array_equal(host(gshare), share)
    # truncate + zero unused
    m0, m1 = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
    o.lib.mmo_truncate_masks(o.p, A(m0), A(m1))
    assert g.truncate_masks() == (int(m0[0]), int(m1[0]))
    keep = np.zeros((1, N), np.uint8)
    keep[0, targets] = 1
    low[:, 0] &= m0[0]
    low[:, 1] &= m1[0]
    low[:, 0, :, keep[0] == 0] = 0  # c0 only (hom_conv2d_ss.cc:134-138)
    g.truncate_and_zero(glow, torch.from_numpy(keep).cuda(), 1)
    assert np.array_equal(host(glow), low)
    # client side: decrypt the truncated ciphertexts; share0 + share1 = m at the targets
    # (exactly or off by one: USE_APPROX_RESHARE, library_fixed_uniform_cheetah.cpp:199)
    dec = host(g.decrypt(dev(np.ascontiguousarray(sk[:1])), glow, lv=1))
    ref = np.zeros((n, N), np.uint64)
    for k in range(n):
        o.lib.mmo_decrypt(o.p, 1, A(np.ascontiguousarray(sk[:1])), A(low[k]), A(ref[k]))
    assert np.array_equal(dec, ref)
    rec = (dec[:, targets].astype(object) + share.astype(object)) % t
    diff = (rec - m[:, targets].astype(object)) % t
    assert all(d in (0, 1, t - 1) for d in diff.ravel())


def test_batch_encode_decode(mm):
    o, g = OracleCtx(N, (60, 49), P32), mm.Context(N, (60, 49), P32)
    vals = splitmix64(SEED, 3 * N).reshape(3, N) & np.uint64((1 << 32) - 1)  # < 2^32, some >= p
    ref = np.zeros((3, N), np.uint64)
    for k in range(3):
        o.lib.mmo_batch_encode(o.p, A(vals[k]), N, A(ref[k]))
    got = g.batch_encode(dev(vals))
    assert np.array_equal(host(got), ref)
    back = host(g.batch_decode(got))
    assert np.array_equal(back, vals % np.uint64(P32))
    dec = np.zeros(N, np.uint64)
    o.lib.mmo_batch_decode(o.p, A(ref[0]), A(dec))
    assert np.array_equal(back[0], dec)


def test_scale_and_round_all_coefficients_first_level(mm):
    t = 1 << 37
    o, g = OracleCtx(N, (60, 49), t), mm.Context(N, (60, 49), t)
    r = rns_uniform(o, SEED + 60, 2, 2)
    ref = np.zeros((2, N), np.uint64)
    for k in range(2):
        o.lib.mmo_decrypt_scale_and_round(o.p, 0, A(r[k]), N, A(ref[k]))
    assert np.array_equal(host(g.scale_and_round(dev(r), None, lv=0)), ref)


def test_bad_arguments(mm):
    g = mm.Context(N, (60, 49), 0)  # NTT-only context has no BFV constants
    with pytest.raises(mm.MMHError) as ei:
        g.decrypt(dev(np.zeros((2, N), np.uint64)), dev(np.zeros((1, 2, 2, N), np.uint64)))
    assert ei.value.code == 1  # ERR_CONFIG
    g2 = mm.Context(N, (60, 49), 1 << 37)
    with pytest.raises(mm.MMHError) as ei:  # index pattern runs past the operand arrays
        g2.mul_plain_acc(dev(np.zeros((2, 2, 2, N), np.uint64)), dev(np.zeros((2, N), np.uint64)),
                         2, 2, 2, 2, 1, 2, 1, 0)
    assert ei.value.code == 6  # ERR_OUT_BOUND


def test_fresh_ciphertext_halves_split_and_join(mm):
    """mmh_ct_split_c0 / mmh_ct_join: the wire form of a fresh symmetric ciphertext is c0 + the seed of c1
    (seal::Serializable<Ciphertext>, cheetah-api.cpp:11-29): take c0 out, put (c0, c1) back, same bits"""
    import torch
    import openmm_amd.lib as L
    g = mm.Context(N, (60, 49), 1 << 37)
    lib = L.load_library()
    cts = torch.randint(0, 1 << 62, (5, 2, 2, N), dtype=torch.int64, device="cuda")
    c0, back = torch.empty(5, 2, N, dtype=torch.int64, device="cuda"), torch.empty_like(cts)
    L.check(lib.mmh_ct_split_c0(g.handle, 0, L.dptr(cts), 5, L.dptr(c0), L._stream_ptr()))
    assert torch.equal(c0, cts[:, 0])
    c1 = cts[:, 1].contiguous()
    L.check(lib.mmh_ct_join(g.handle, 0, L.dptr(c0), L.dptr(c1), 5, L.dptr(back), L._stream_ptr()))
    assert torch.equal(back, cts)
    last = torch.randint(0, 1 << 59, (3, 2, 1, N), dtype=torch.int64, device="cuda")
    c0l = torch.empty(3, 1, N, dtype=torch.int64, device="cuda")
    L.check(lib.mmh_ct_split_c0(g.handle, 1, L.dptr(last), 3, L.dptr(c0l), L._stream_ptr()))
    assert torch.equal(c0l, last[:, 0])
