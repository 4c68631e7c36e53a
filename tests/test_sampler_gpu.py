"""Device sampler: ChaCha20 known-answer test (RFC 8439 2.3.2), distribution checks and an
end-to-end keygen/encrypt/decrypt round trip on sampled randomness."""
import ctypes as C
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 4096


def chacha_block(key_words, w12, w13, w14, w15):
    """RFC 8439 block function, written from the specification"""
    rotl = lambda x, n: ((x << n) & 0xffffffff) | (x >> (32 - n))

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xffffffff; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xffffffff; s[b] = rotl(s[b] ^ s[c], 7)
    init = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [w12, w13, w14, w15]
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return [(a + b) & 0xffffffff for a, b in zip(s, init)]


RFC_8439_2_3_2 = ("e4e7f110 15593bd1 1fdd0f50 c47120a3 c7f4d1c7 0368c033 9aaa2204 4e6cd4c3 "
                  "466482d2 09aa9f07 05d7c214 a2028bd9 d19c12b5 b94e16de e883d0cb 4e3c50a2")


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available()
    import openmm_amd
    return openmm_amd


def test_chacha20_known_answer(mm):
    import torch
    from openmm_amd.lib import check, dptr, load_library
    lib = load_library()
    key = bytes(range(32))
    kw = (C.c_uint32 * 8).from_buffer_copy(key)
    out = torch.zeros(24, dtype=torch.int64, device="cuda")
    # RFC vector: counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 -> words 13,14,15
    check(lib.mmh_chacha20_words(kw, 0x000000004a000000, 0x0900000000000001, dptr(out), 24, None))
    words = out.cpu().numpy().view(np.uint32)
    assert " ".join("%08x" % w for w in words[:16]) == RFC_8439_2_3_2
    kws = struct.unpack("<8I", key)
    assert list(words[:16]) == chacha_block(kws, 1, 0x09000000, 0x4a000000, 0)
    assert list(words[16:32]) == chacha_block(kws, 2, 0x09000000, 0x4a000000, 0)   # next block = counter + 1


def test_distributions_and_roundtrip(mm):
    import torch
    g = mm.Context(N, (60, 49), 1 << 37)
    smp = mm.Sampler(g, key=bytes(range(32)))
    t = smp.ternary(64).cpu().numpy()
    assert set(np.unique(t)) == {-1, 0, 1}
    assert all(abs((t == v).mean() - 1 / 3) < 0.01 for v in (-1, 0, 1))
    e = smp.noise(64).cpu().numpy()
    assert abs(e.mean()) < 0.05 and abs(e.var() - 10.5) < 0.3 and np.abs(e).max() <= 21
    u = smp.uniform(8).cpu().numpy().view(np.uint64)
    for l, q in enumerate(g.q):
        assert (u[:, l] < q).all() and abs(u[:, l].astype(np.float64).mean() / q - 0.5) < 0.01
    assert not np.array_equal(smp.words(16).cpu().numpy(), smp.words(16).cpu().numpy())  # fresh stream per call
    # keygen + encrypt + decrypt on sampled randomness
    sk, pk = smp.keygen()
    m = torch.randint(0, 1 << 37, (5, N), dtype=torch.int64, device="cuda")
    cts = g.encrypt_symmetric(sk, smp.uniform(5), smp.noise(5), m)
    assert torch.equal(g.decrypt(sk, cts), m)
    z = g.encrypt_zero_asym(pk, smp.ternary(5), smp.noise(5), smp.noise(5))
    z = g.add_plain(z, m)
    assert torch.equal(g.decrypt(sk, z), m)


def test_reshare_launch_equals_the_three_samplers(mm):
    """mmh_sample_reshare = ternary(sid) | noise(sid + 1) | uniform at the last level (sid + 2), same values; lengths
    that end inside a tile of every kind"""
    import torch
    g = mm.Context(N, (60, 49), 1 << 37)
    for n in (1, 3, 64):
        a, b = mm.Sampler(g, key=bytes(range(32))), mm.Sampler(g, key=bytes(range(32)))
        u, e1, mask = a.reshare(n)
        assert torch.equal(u, b.ternary(n)) and torch.equal(e1, b.noise(n))
        assert torch.equal(mask, b.uniform(n, lv=1).view(n, N))
        assert a.next_stream == b.next_stream


def test_reshare_range_is_a_slice_of_the_whole_draw(mm):
    """mmh_sample_reshare_range(first, n) = rows [first, first + n) of mmh_sample_reshare: ranges that start and end
    inside tiles of every kind (a noise tile is 3072 samples, a polynomial 4096)"""
    import torch
    g = mm.Context(N, (60, 49), 1 << 37)
    smp = mm.Sampler(g, key=bytes(range(32)))
    sid = smp.next_stream
    whole = smp.reshare(23)
    for first, n in ((0, 23), (1, 1), (5, 7), (22, 1), (11, 12)):
        part = smp.reshare_range(sid, first, n)
        for w, p in zip(whole, part):
            assert torch.equal(w[first:first + n], p)


def test_sample_mapping_against_the_block_function(mm):
    """the three distributions spend the keystream as include/mm_hip.h / sampler.hip say: thread t of tile T owns
    block 256 T + t and its j-th sample is element 256 W T + 256 j + t (W = 16 ternary from 32-bit words, 12 noise
    from 42-bit fields, 8 uniform from 64-bit words); checked against the RFC block function in Python"""
    g = mm.Context(N, (60, 49), 1 << 37)
    key = bytes(range(32))
    kws = struct.unpack("<8I", key)
    smp = mm.Sampler(g, key=key)
    t = smp.ternary(2).cpu().numpy().reshape(-1)      # stream 1
    e = smp.noise(2).cpu().numpy().reshape(-1)        # stream 2
    u = smp.uniform(2, lv=1).cpu().numpy().view(np.uint64).reshape(-1)   # stream 3
    q0 = g.q[0]
    for tile, tid in ((0, 0), (0, 255), (1, 17)):
        for sid, per, arr in ((1, 16, t), (2, 12, e), (3, 8, u)):
            blk = chacha_block(kws, tile * 256 + tid, 0, sid, 0)
            bits = sum(w << (32 * i) for i, w in enumerate(blk))
            for j in range(per):
                idx = tile * 256 * per + 256 * j + tid
                if idx >= arr.size:
                    continue
                if per == 16:
                    want = ((blk[j] * 3) >> 32) - 1
                elif per == 12:
                    v = (bits >> (42 * j)) & ((1 << 42) - 1)
                    want = bin(v & 0x1FFFFF).count("1") - bin(v >> 21).count("1")
                else:
                    w = (bits >> (64 * j)) & ((1 << 64) - 1)
                    if w >= q0 * ((1 << 64) // q0):
                        continue   # (2^-46: the fallback block)
                    want = w % q0
                assert int(arr[idx]) == want, (sid, tile, tid, j)
