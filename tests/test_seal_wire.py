"""SEAL 3.7's serialisation as the reference puts it on the wire (include/seal/wire.h; cheetah-api.cpp
:11-29,:113-135,:723-737; patch/SEAL.patch:43-71).  PARITY UNPINNED against SEAL itself (its sources are not in the
reference tree and no serialised object is either); what can be pinned here is pinned:
  * BLAKE2b against RFC 7693 appendix A and against Python's hashlib (keyed and unkeyed, every block boundary);
  * BLAKE2xb: the root hash against hashlib (which accepts its parameters), the output blocks against a pure-Python
    BLAKE2b that is itself checked against hashlib (hashlib refuses their depth = 0);
  * the byte layout against a second, independent statement of it in this file (struct.pack), sizes included;
  * save -> load round trips, the seeded form's expansion;
  * the Zstandard decoder (include/seal/zstd_decode.h) against the libzstd inside pyarrow, and a ciphertext object
    compressed the way SEAL's default save() compresses it.
Host code only: the tool is compiled with g++ here, no device library involved."""
import hashlib
import os
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q0, Q1, T = 1152921504606830593, 562949953216513, 1 << 37     # SEAL's {60, 49}-bit primes for N = 4096 (the ring's)

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not present")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("wire") / "wire_tool")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT}/include", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "wire_tool.cc")], check=True, timeout=280)

    def run(*args, stdin=None, ok=True):
        r = subprocess.run([exe] + [str(a) for a in args], input=stdin, capture_output=True, text=True, timeout=120)
        assert (r.returncode == 0) == ok, (args[:2], r.stdout[-300:], r.stderr[-300:])
        return r.stdout.split("\n")[:-1]
    return run


# ---------------------------------------------------------------- a pure-Python BLAKE2b over a raw parameter block
IV = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1,
      0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b, 0x5be0cd19137e2179]
SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
         [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
         [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
         [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
         [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]
M64 = (1 << 64) - 1


def py_blake2b(param_block, data, key=b""):
    h = [iv ^ w for iv, w in zip(IV, struct.unpack("<8Q", param_block))]
    outlen = param_block[0]
    if key:
        data = key.ljust(128, b"\0") + data
    blocks = [data[i:i + 128] for i in range(0, len(data), 128)] or [b""]
    t = 0
    rot = lambda x, n: ((x >> n) | (x << (64 - n))) & M64
    for bi, blk in enumerate(blocks):
        last = bi == len(blocks) - 1
        t += len(blk)
        m = struct.unpack("<16Q", blk.ljust(128, b"\0"))
        v = h + IV[:]
        v[12] ^= t & M64
        v[13] ^= t >> 64
        if last:
            v[14] ^= M64
        for r in range(12):
            s = SIGMA[r % 10]
            for i, (a, b, c, d) in enumerate(((0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15),
                                              (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14))):
                v[a] = (v[a] + v[b] + m[s[2 * i]]) & M64
                v[d] = rot(v[d] ^ v[a], 32)
                v[c] = (v[c] + v[d]) & M64
                v[b] = rot(v[b] ^ v[c], 24)
                v[a] = (v[a] + v[b] + m[s[2 * i + 1]]) & M64
                v[d] = rot(v[d] ^ v[a], 16)
                v[c] = (v[c] + v[d]) & M64
                v[b] = rot(v[b] ^ v[c], 63)
        h = [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]
    return struct.pack("<8Q", *h)[:outlen]


def py_blake2xb(outlen, data, key):
    root = hashlib.blake2b(data, digest_size=64, key=key, node_offset=outlen << 32).digest()   # xof_length: upper half
    out = b""
    i = 0
    while len(out) < outlen:
        take = min(64, outlen - len(out))
        p = bytearray(64)
        p[0] = take
        p[4:8] = struct.pack("<I", 64)
        p[8:12] = struct.pack("<I", i)
        p[12:16] = struct.pack("<I", outlen)
        p[17] = 64
        out += py_blake2b(bytes(p), root)
        i += 1
    return out


def test_python_blake2b_matches_hashlib():
    for n in (0, 1, 3, 64, 127, 128, 129, 255, 256, 257, 1000):
        data = bytes((7 * i + n) & 255 for i in range(n))
        for outlen in (16, 32, 64):
            p = bytearray(64)
            p[0], p[2], p[3] = outlen, 1, 1
            assert py_blake2b(bytes(p), data) == hashlib.blake2b(data, digest_size=outlen).digest()
        p = bytearray(64)
        p[0], p[1], p[2], p[3] = 64, 64, 1, 1
        key = bytes(range(64))
        assert py_blake2b(bytes(p), data, key) == hashlib.blake2b(data, key=key).digest()


def test_blake2b_rfc7693_and_hashlib(tool):
    # RFC 7693 appendix A: BLAKE2b-512("abc")
    assert tool("blake2b", 64, b"abc".hex())[0] == (
        "ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
        "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    key = bytes(range(64))
    for n in (1, 55, 127, 128, 129, 256, 300):
        data = bytes((11 * i + n) & 255 for i in range(n))
        for outlen in (32, 64):
            assert tool("blake2b", outlen, data.hex())[0] == hashlib.blake2b(data, digest_size=outlen).hexdigest()
        assert tool("blake2b", 64, data.hex(), key.hex())[0] == hashlib.blake2b(data, key=key).hexdigest()


def test_blake2xb(tool):
    key = bytes((3 * i + 1) & 255 for i in range(64))
    for outlen, data in ((4096, struct.pack("<Q", 0)), (4096, struct.pack("<Q", 5)), (100, b"xof"), (64, b""), (65, b"a")):
        if not data:
            continue   # the tool cannot take an empty hex argument
        assert tool("blake2xb", outlen, data.hex(), key.hex())[0] == py_blake2xb(outlen, data, key).hex()


def py_parms_id(q, t, N=4096, scheme=1, n_special=0):
    words = [scheme, N, n_special] + list(q) + [t]
    return struct.unpack("<4Q", hashlib.blake2b(struct.pack(f"<{len(words)}Q", *words), digest_size=32).digest())


def test_parms_id_is_blake2b_of_the_patched_word_list(tool):
    for q in ((Q0, Q1), (Q0,)):
        got = tuple(int(x, 16) for x in tool("parms", 1, 4096, 0, T, *q))
        assert got == py_parms_id(q, T)
    # the patch's extra word matters: with and without it the ids differ (an unpatched SEAL would not accept ours)
    words = [1, 4096, Q0, Q1, T]
    unpatched = struct.unpack("<4Q", hashlib.blake2b(struct.pack("<5Q", *words), digest_size=32).digest())
    assert unpatched != py_parms_id((Q0, Q1), T)


def py_uniform(seed, q, N):
    """SEAL's Blake2xb generator and sample_poly_uniform, second statement"""
    state = {"buf": b"", "ctr": 0}

    def gen(n):
        out = b""
        while len(out) < n:
            if not state["buf"]:
                state["buf"] = py_blake2xb(4096, struct.pack("<Q", state["ctr"]), seed)
                state["ctr"] += 1
            take = min(n - len(out), len(state["buf"]))
            out += state["buf"][:take]
            state["buf"] = state["buf"][take:]
        return out
    words = list(struct.unpack(f"<{len(q) * N}Q", gen(len(q) * N * 8)))
    for l, m in enumerate(q):
        limit = M64 - (M64 % m) - 1
        for i in range(N):
            r = words[l * N + i]
            while r >= limit:
                r = struct.unpack("<Q", gen(8))[0]
            words[l * N + i] = r % m
    return words


def test_seed_expansion(tool):
    seed = bytes((5 * i + 9) & 255 for i in range(64))
    N = 1024   # two buffers per limb
    got = [int(x, 16) for x in tool("uniform", seed.hex(), N, Q0, Q1)]
    assert got == py_uniform(seed, (Q0, Q1), N)
    # a modulus just above 2^63 rejects half the words: the redraw path
    big = (1 << 63) + 29
    got = [int(x, 16) for x in tool("uniform", seed.hex(), 256, big)]
    assert got == py_uniform(seed, (big,), 256)
    assert all(x < big for x in got)


def fill_word(i, q):
    return ((0x9E3779B97F4A7C15 * (i + 1) + 12345) & M64) % q


def py_header(total, compr=0):
    return struct.pack("<HBBBBHQ", 0xA15E, 0x10, 3, 7, compr, 0, total)


def py_ciphertext(q, t, N, size, ntt, seed=None):
    L = len(q)
    data = [fill_word((k * L + l) * N + i, q[l]) for k in range(size) for l in range(L) for i in range(N)]
    if seed is not None:
        data = data[:L * N]
    arr = struct.pack("<Q", len(data)) + struct.pack(f"<{len(data)}Q", *data)
    arr = py_header(16 + len(arr)) + arr
    members = (struct.pack("<4Q", *py_parms_id(q, t, N)) + struct.pack("<B", ntt) +
               struct.pack("<4Q", size, N, L, 1) + struct.pack("<d", 1.0))
    body = members + arr
    if seed is not None:
        body += py_header(16 + 1 + 64) + b"\x01" + seed
    return py_header(16 + len(body)) + body, data


@pytest.mark.parametrize("q,N,ntt", [((Q0, Q1), 4096, 0), ((Q0,), 4096, 0), ((Q0, Q1), 64, 1)])
def test_ciphertext_layout_and_round_trip(tool, q, N, ntt):
    seed = bytes(range(64))
    for seeded in (0, 1):
        want, data = py_ciphertext(q, T, N, 2, ntt, seed if seeded else None)
        got = tool("save", seeded, ntt, 2, N, seed.hex(), T, *q)[0]
        assert bytes.fromhex(got) == want
        # SEAL's sizes: 16 + 73 members + 24 array framing + words (+ 81 for the generator info)
        L = len(q)
        assert len(want) == 16 + 73 + 24 + 8 * (L * N if seeded else 2 * L * N) + (81 if seeded else 0)
        out = tool("load", stdin=got)
        assert out[0].split()[1:] == ["%016x" % w for w in py_parms_id(q, T, N)]
        f = out[1].split()
        assert f[1::2][:5] == [str(ntt), "2", str(N), str(L), "1"] and f[f.index("seeded") + 1] == str(seeded)
        assert int(f[f.index("consumed") + 1]) == len(want)
        s = 0
        for w in data:
            s = (s * 0x100000001B3 + w) & M64
        assert out[-1] == "checksum %016x" % s
        if seeded:
            assert out[2] == seed.hex()


def zstd_inputs():
    """inputs that walk the decoder's branches: raw / RLE / compressed blocks, Huffman literals with direct and
    FSE-compressed weights, one and four streams, repeat offsets, long matches, several blocks per frame"""
    import numpy as np
    rng = np.random.default_rng(0x4D264D)
    yield "empty", b""
    yield "one byte", b"x"
    yield "random 1000 (raw block)", rng.bytes(1000)
    yield "zeros 100000 (RLE)", bytes(100000)
    yield "text", (b"the quick brown fox jumps over the lazy dog. " * 300)[:12345]
    yield "skewed bytes (Huffman, few symbols)", bytes(rng.choice([0, 1, 2, 3, 250], size=5000, p=[.6, .2, .1, .05, .05]).astype(np.uint8))
    yield "skewed bytes 200 KiB (several blocks, four streams)", bytes(rng.geometric(0.08, size=200 * 1024).astype(np.uint8))
    yield "many symbols (FSE-compressed weights)", bytes(np.clip(rng.normal(128, 30, size=60000), 0, 255).astype(np.uint8))
    yield "repeats with gaps", b"".join(bytes([i % 7]) * (i % 50) + b"abcdefgh" for i in range(3000))
    words = rng.integers(0, 1 << 60, size=2 * 4096, dtype=np.uint64)
    yield "ciphertext words", words.tobytes()
    trunc = (words >> np.uint64(23)) << np.uint64(23)
    trunc[4096 + 200:] = 0
    yield "truncated ciphertext (cleared low bits, removed coefficients)", trunc.tobytes()
    yield "1.5 MiB mixed", b"".join([rng.bytes(300000), bytes(500000), (b"0123456789" * 70000)[:600000],
                                    bytes(rng.geometric(0.3, size=100000).astype(np.uint8))])


def test_zstd_decoder_against_libzstd(tool):
    """the decoder of include/seal/zstd_decode.h against the libzstd inside pyarrow: one-shot frames (content size in
    the header) at several levels and streaming frames (no content size, as SEAL's writer produces them)"""
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("zstd"):
        pytest.skip("pyarrow built without zstd")
    for name, data in zstd_inputs():
        for level in (1, 3, 9, 19):
            z = pa.Codec("zstd", compression_level=level).compress(data, asbytes=True)
            got = tool("inflate", stdin=z.hex() or "00")[0] if z else ""
            assert bytes.fromhex(got) == data, (name, level)
        sink = pa.BufferOutputStream()
        with pa.CompressedOutputStream(sink, "zstd") as f:
            for i in range(0, len(data), 70000):
                f.write(data[i:i + 70000])
        z = sink.getvalue().to_pybytes()
        if z:
            assert bytes.fromhex(tool("inflate", stdin=z.hex())[0]) == data, (name, "stream")
        # two frames back to back decode to the concatenation
        if data and len(data) < 20000:
            z2 = pa.Codec("zstd").compress(data, asbytes=True) * 2
            assert bytes.fromhex(tool("inflate", stdin=z2.hex())[0]) == data * 2, (name, "two frames")
    out = tool("inflate", stdin=pa.Codec("zstd").compress(b"abc" * 100, asbytes=True)[:-3].hex(), ok=False)
    assert out[0].startswith("error: zstd")


@pytest.mark.parametrize("seeded", [0, 1])
def test_ciphertext_saved_with_seals_default_compression_loads(tool, seeded):
    """what an unmodified reference party sends: SEALHeader with compr_mode zstd, then the members as one Zstandard
    stream (nested objects uncompressed inside it)"""
    pa = pytest.importorskip("pyarrow")
    q, N = (Q0, Q1), 4096
    seed = bytes(range(64))
    plain, data = py_ciphertext(q, T, N, 2, 0, seed if seeded else None)
    sink = pa.BufferOutputStream()
    with pa.CompressedOutputStream(sink, "zstd") as f:
        f.write(plain[16:])
    z = sink.getvalue().to_pybytes()
    obj = py_header(16 + len(z), compr=2) + z
    out = tool("load", stdin=obj.hex())
    ref = tool("load", stdin=plain.hex())
    assert out[0] == ref[0] and out[-1] == ref[-1]
    f = out[1].split()
    assert int(f[f.index("consumed") + 1]) == len(obj) and f[f.index("seeded") + 1] == str(seeded)
    # a truncated-for-decryption result compresses, and still loads to the same words
    import numpy as np
    words = (np.frombuffer(plain[16 + 73 + 24:16 + 73 + 24 + 8 * 2 * N], dtype=np.uint64) >> np.uint64(20)) << np.uint64(20)
    if not seeded:
        body = plain[16:16 + 73 + 24] + words.tobytes() + plain[16 + 73 + 24 + 8 * 2 * N:]
        z = pa.Codec("zstd").compress(body, asbytes=True)
        assert len(z) < len(body) * 0.9
        out = tool("load", stdin=(py_header(16 + len(z), compr=2) + z).hex())
        assert out[1].split()[1::2][:4] == ["0", "2", str(N), "2"]


def test_compressed_and_damaged_streams_are_refused(tool):
    want, _ = py_ciphertext((Q0,), T, 64, 2, 0)
    bad = bytearray(want)
    bad[5] = 1      # zlib: no decoder
    out = tool("load", stdin=bytes(bad).hex(), ok=False)
    assert "compressed stream" in out[0] and "zlib" in out[0]
    bad[5] = 2      # marked zstd, but the bytes behind the header are no Zstandard frame
    assert "zstd" in tool("load", stdin=bytes(bad).hex(), ok=False)[0]
    for mutate in (lambda b: b[:-8], lambda b: b[:100], lambda b: b"\x00" + b[1:]):
        out = tool("load", stdin=bytes(mutate(bytes(want))).hex(), ok=False)
        assert out[0].startswith("error:")
    longer = bytearray(want)
    longer[8:16] = struct.pack("<Q", len(want) + 8)     # size field beyond the bytes present
    assert tool("load", stdin=bytes(longer).hex(), ok=False)[0].startswith("error:")
