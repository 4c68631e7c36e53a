"""CPU: pins for the COT post-processing oracle (oracle/mm_cot_oracle.c, silent_ot.h:85-242).

AES-128 is pinned by the FIPS-197 known answers (Appendix B, Appendix C.1 and the last round keys of A.1).  The
mask / mod-p / correction / packing arithmetic is checked through what the reference's own code guarantees:
the receiver ends with data0 + b * corr (mod 2^l or mod p) whatever the hash is, the wire has the reference's
size, and the packed words are the elements' bits at the positions ot-utils.h:170-184 gives.  The MITCCRH key
derivation itself is restated from emp-tool (absent here): parity unpinned, see the oracle's header."""
import numpy as np
import pytest

from tests.oracle_binding import A, load, splitmix64

SEED = 0x4D264D
FIPS = [  # key, plaintext, ciphertext, last round key
    ("000102030405060708090a0b0c0d0e0f", "00112233445566778899aabbccddeeff",
     "69c4e0d86a7b0430d8cdb78070b4c55a", "13111d7fe3944a17f307a78b4d2b30c5"),
    ("2b7e151628aed2a6abf7158809cf4f3c", "3243f6a8885a308d313198a2e0370734",
     "3925841d02dc09fbdc118597196a0b32", "d014f9a8c9ee2589e13f0cc8b6630ca6"),
]


def hexbytes(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8).copy()


@pytest.mark.parametrize("key,pt,ct,last", FIPS)
def test_aes128_fips197(key, pt, ct, last):
    o = load()
    rk, out = np.zeros(176, np.uint8), np.zeros(16, np.uint8)
    o.mmo_aes128_expand(A(hexbytes(key)), A(rk))
    o.mmo_aes128_encrypt(A(rk), A(hexbytes(pt)), A(out))
    assert out.tobytes().hex() == ct and rk[160:].tobytes().hex() == last


def test_mitccrh_is_block_xor_aes_under_counter_keys():
    """out = blk ^ AES_{S ^ (gid << 64)}(blk); the H blocks of an element share its key"""
    o, n, H = load(), 19, 2
    S = splitmix64(SEED, 2)
    blocks = splitmix64(SEED + 1, 2 * n * H)
    out = np.zeros_like(blocks)
    o.mmo_mitccrh_hash(A(S), 1000, A(blocks), n, H, A(out))
    rk, enc = np.zeros(176, np.uint8), np.zeros(16, np.uint8)
    for j in (0, 7, 8, 18):
        key = np.array([S[0], S[1] ^ np.uint64(1000 + j)], dtype=np.uint64)
        o.mmo_aes128_expand(A(key), A(rk))
        for h in range(H):
            blk = blocks[2 * (j * H + h): 2 * (j * H + h) + 2].copy()
            o.mmo_aes128_encrypt(A(rk), A(blk), A(enc))
            assert np.array_equal(out[2 * (j * H + h): 2 * (j * H + h) + 2], blk ^ enc.view(np.uint64))


def cot_pair(o, n, l, p, seed, corr=None, gid0=77):
    """sender and receiver of the oracle over an ideal Ferret; returns data0, data, b, corr, wire"""
    prime = 1 if p else 0
    delta = splitmix64(seed, 2) | np.uint64(1)
    S = splitmix64(seed + 1, 2)
    rcm = splitmix64(seed + 2, 2 * n)
    b = (splitmix64(seed + 3, n) & np.uint64(1)).astype(np.uint8)
    if corr is None:
        corr = splitmix64(seed + 4, n)
    rcm_r = (rcm.reshape(n, 2) ^ (b.astype(np.uint64).reshape(n, 1) * delta.reshape(1, 2))).reshape(-1)
    nbytes = o.mmo_cot_wire_bytes(n, l, p, prime)
    wire = np.zeros(nbytes + 8, np.uint8)
    data0, data = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    o.mmo_cot_send(A(rcm), A(delta), A(S), gid0, A(corr), n, l, p, prime, A(data0), A(wire))
    assert not wire[nbytes:].any()
    o.mmo_cot_recv(A(rcm_r), A(b), A(S), gid0, A(wire), n, l, p, prime, A(data))
    return data0, data, b, corr, wire[:nbytes], (rcm, rcm_r, delta, S)


@pytest.mark.parametrize("l", [1, 2, 8, 12, 32, 37, 41, 63, 64])
@pytest.mark.parametrize("n", [1, 7, 8, 9, 1003])
def test_cot_ring_correlation(l, n):
    o = load()
    data0, data, b, corr, wire, _ = cot_pair(o, n, l, 0, SEED + 100 * l + n)
    mask = (1 << l) - 1
    for j in range(n):
        assert int(data0[j]) <= mask
        assert int(data[j]) == (int(data0[j]) + int(b[j]) * int(corr[j])) & mask      # silent_ot.h:111-116,198-204
    full, rem = divmod(n, 8)
    assert len(wire) == 8 * (full * ((8 * l + 63) // 64) + (rem * l + 63) // 64)      # :117-123


@pytest.mark.parametrize("p", [2, 65537, (1 << 32) - 5, 1152921504606830593, (1 << 37), (1 << 64) - 59])
@pytest.mark.parametrize("n", [1, 8, 9, 1003])
def test_cot_prime_correlation(p, n):
    o = load()
    data0, data, b, corr, wire, _ = cot_pair(o, n, 0, p, SEED + n + p % 1000)
    lb = ((p - 1).bit_length() + 7) // 8 if p & (p - 1) else (p.bit_length() - 1 + 7) // 8
    assert len(wire) == n * lb                                                         # :139,:159
    for j in range(n):
        assert int(data0[j]) < p
        assert int(data[j]) == (int(data0[j]) + int(b[j]) * int(corr[j])) % p         # :153-156,:234-238


def test_packed_words_follow_ot_utils():
    """l = 37, one full group: element i occupies bits [37 i, 37 i + 37) of the group's five words"""
    o, n, l = load(), 8, 37
    data0, data, b, corr, wire, (rcm, _, delta, S) = cot_pair(o, n, l, 0, SEED + 5)
    words = wire.view(np.uint64)
    assert len(words) == 5
    big = sum(int(w) << (64 * i) for i, w in enumerate(words))
    # the sender's corrections are data(b = 1): corr + data0 (+ hash of the other pad, which data removes)
    h = np.zeros(2 * n, np.uint64)
    o.mmo_mitccrh_hash(A(S), 77, A(rcm ^ np.tile(delta, n)), n, 1, A(h))
    for i in range(n):
        elem = (big >> (l * i)) & ((1 << l) - 1)
        assert elem == (int(corr[i]) + int(data0[i]) + int(h[2 * i])) & ((1 << l) - 1)
    assert big >> (l * n) == 0


def test_gid_advance_covers_ragged_batches():
    o = load()
    assert [o.mmo_cot_gids(n) for n in (0, 1, 8, 9, 16, 1003)] == [0, 8, 8, 16, 16, 1008]


def test_abi_wire_sizes_match_the_oracle_without_gpu():
    """mmh_cot_wire_bytes / mmh_cot_gids are host arithmetic: same values as the oracle over widths, moduli, lengths"""
    import ctypes as C
    from openmm_amd import lib as L
    o, lib = load(), L.load_library()
    nb, g = C.c_size_t(), C.c_uint64()
    for n in (0, 1, 7, 8, 9, 255, 1003, 10 ** 7):
        assert lib.mmh_cot_gids(n, C.byref(g)) == 0 and g.value == o.mmo_cot_gids(n)
        for l in (1, 2, 7, 8, 12, 32, 37, 41, 63, 64):
            assert lib.mmh_cot_wire_bytes(n, l, 0, 0, C.byref(nb)) == 0
            assert nb.value == o.mmo_cot_wire_bytes(n, l, 0, 0)
        for p in (2, 3, 255, 256, 257, 65537, 4293918721, 1 << 37, (1 << 61) - 1, (1 << 64) - 59):
            assert lib.mmh_cot_wire_bytes(n, 0, p, 1, C.byref(nb)) == 0
            assert nb.value == o.mmo_cot_wire_bytes(n, 0, p, 1)
    assert lib.mmh_cot_wire_bytes(8, 0, 0, 0, C.byref(nb)) == 7 and lib.mmh_cot_wire_bytes(8, 0, 1, 1, C.byref(nb)) == 7
