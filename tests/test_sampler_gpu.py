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
