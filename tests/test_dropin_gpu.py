"""The batched implementations of the reference's protocol classes (openmm_amd/dropin: member functions of
gemini::HomConv2DSS / HomFCSS / HomBNSS as the REFERENCE'S OWN headers declare them, on the seal:: facade, one libmm_hip
layer call per method) -- what a maintainer links in place of include/gemini/cheetah/hom_*_ss.cc.

* tests/cpp/_ref/ref_dropin_dump runs conv / fc / bn through them with injected randomness and writes everything to a
  file; the same layers are replayed here through the CPU oracle with the randomness the device samplers give for that
  key, and the output ciphertexts and shares must be equal bit for bit.
* tests/cpp/_ref/ref_dropin_api_test is the reference's protocol driver (SCI/src/cheetah/cheetah-api.cpp, unchanged,
  two parties over sci::NetIO) on top of them, with the reference's own acceptance checks.
The binaries are built where /root/reference exists (tests/cpp/Makefile) and travel with the snapshot."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.oracle_binding import OracleCtx
from tests.oracle_layers import BnOracle, ConvOracle, FcOracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "cpp", "_ref")
N = 4096


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            head = f.read(4)
            if not head:
                return out
            name = f.read(struct.unpack("<I", head)[0]).decode()
            nd = struct.unpack("<I", f.read(4))[0]
            dims = struct.unpack(f"<{nd}Q", f.read(8 * nd))
            out[name] = np.frombuffer(f.read(8 * int(np.prod(dims))), dtype=np.uint64).reshape(dims).copy()


@pytest.fixture(scope="module", autouse=True)
def gpu():
    import torch   # (torch brings the HIP runtime up first, as in every other GPU test module)
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = os.path.join(REF, "ref_dropin_dump")
    if not os.path.exists(exe):
        pytest.skip("tests/cpp/_ref/ref_dropin_dump is built only where the reference tree exists")
    path = str(tmp_path_factory.mktemp("dropin") / "dump.bin")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=200)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL WRITTEN" in r.stdout
    return read_dump(path)


def sampler_for(dump, index, t):
    """the generator the index-th server object got from the fixed factory: key K with k[7] += index"""
    import openmm_amd as mm
    k = [int(x) for x in dump["key"]]
    k[7] = (k[7] + index) & 0xFFFFFFFF
    g = mm.Context(N, (60, 49), t)
    return g, mm.Sampler(g, key=struct.pack("<8I", *k))


def host(t):
    return t.cpu().numpy().view(np.uint64)


def test_conv_through_the_reference_class_equals_the_oracle(dump):
    t = 1 << 37
    o = OracleCtx(N, (60, 49), t)
    cv = ConvOracle(o, (20, 14, 14), (20, 3, 3), 12, 1, 1, True)
    g, smp = sampler_for(dump, 0, t)
    u, e1, mask = (host(x) for x in smp.reshare(cv.n_out))           # streams 1, 2, 3 of that key
    pk1 = np.ascontiguousarray(dump["pk"][:, :1])
    ref_out, ref_share = cv.server(dump["conv_in_cts"], cv.encode_image(dump["conv_share1"]),
                                   cv.encode_filters(dump["conv_filters"]), pk1, u.view(np.int64), e1.view(np.int64), mask)
    assert np.array_equal(dump["conv_out_cts"], ref_out), "HomConv2DSS::conv2DSS ciphertexts differ from the oracle"
    assert np.array_equal(dump["conv_out_share"], ref_share.reshape(-1))
    sk1 = np.ascontiguousarray(dump["sk"][:1])
    assert np.array_equal(dump["conv_dec"], cv.decrypt(sk1, ref_out).reshape(-1))
    # plaintexts of the lazy batches, read on the host: what TensorEncoder writes
    assert np.array_equal(dump["conv_filter_pt_5_1"], cv.encode_filters(dump["conv_filters"])[5, 1])
    assert np.array_equal(dump["conv_share_pt_1"], cv.encode_image(dump["conv_share1"])[1])
    # and the protocol identity: shares reconstruct to the convolution within one unit
    x = ((dump["conv_share0"] + dump["conv_share1"]) & np.uint64(t - 1)).astype(np.int64)
    w = dump["conv_filters"].astype(np.int64)
    xp = np.zeros((20, 16, 16), np.int64)
    xp[:, 1:15, 1:15] = x
    want = np.zeros((12, 14, 14), np.int64)
    for r in range(3):
        for q in range(3):   # 64-bit wrap-around is harmless: everything is taken mod 2^37 below
            want += np.einsum("mc,chw->mhw", w[:, :, r, q], xp[:, r:r + 14, q:q + 14])
    d = (dump["conv_dec"] + dump["conv_out_share"] - want.reshape(-1).astype(np.uint64)) & np.uint64(t - 1)
    d = np.where(d >= t // 2, d.astype(np.int64) - t, d.astype(np.int64))
    assert np.abs(d).max() <= 1


def test_fc_through_the_reference_class_equals_the_oracle(dump):
    t = 1 << 37
    o = OracleCtx(N, (60, 49), t)
    fc = FcOracle(o, 70, 300, True)
    g, smp = sampler_for(dump, 1, t)
    u, e1, mask = (host(x) for x in smp.reshare(fc.n_out))
    pk1 = np.ascontiguousarray(dump["pk"][:, :1])
    ref_out, ref_share = fc.server(dump["fc_in_cts"], fc.encode_input(dump["fc_vec1"]), fc.encode_weights(dump["fc_w"]),
                                   pk1, u.view(np.int64), e1.view(np.int64), mask)
    assert np.array_equal(dump["fc_out_cts"], ref_out) and np.array_equal(dump["fc_out_share"], ref_share)
    assert np.array_equal(dump["fc_dec"], fc.decrypt(np.ascontiguousarray(dump["sk"][:1]), ref_out))


def test_bn_through_the_reference_class_equals_the_oracle(dump):
    p = 137438822401
    o = OracleCtx(N, (60, 49), p)
    n = 10000
    bn = BnOracle(o, n)
    g, smp = sampler_for(dump, 2, p)
    rnd = host(smp.words(n))                                         # addMask's r: stream 1 of that key
    ref_out, ref_share = bn.bn(dump["bn_in_cts"], bn.encode(dump["bn_x1"]), bn.encode(dump["bn_scale"]), rnd)
    assert np.array_equal(dump["bn_out_cts"], ref_out) and np.array_equal(dump["bn_out_share"], ref_share)
    assert np.array_equal(dump["bn_dec"], bn.decrypt(np.ascontiguousarray(dump["bn_sk"][:1]), ref_out))
    x = (dump["bn_x0"].astype(object) + dump["bn_x1"].astype(object)) * dump["bn_scale"].astype(object) % p
    assert np.array_equal((dump["bn_dec"].astype(object) + dump["bn_out_share"].astype(object)) % p, x)


def test_reference_protocol_driver_over_the_batched_classes():
    """SCI/src/cheetah/cheetah-api.cpp, unchanged, two parties over sci::NetIO on 127.0.0.1: conv2d (incl. BASELINE
    configs[2] at full size), fc, bn, BOLE, bn_direct with the reference's own checks"""
    exe = os.path.join(REF, "ref_dropin_api_test")
    if not os.path.exists(exe):
        pytest.skip("built only where the reference tree exists")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=280)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.parametrize("alice,bob", [("fast", "slow"), ("slow", "fast"), ("fast", "fast")])
def test_cross_stack_two_processes(alice, bob):
    """Reference-code BOB (the reference's own hom_*_ss.cc over the facade) against fast-path ALICE (the batched classes),
    and the other way round: two processes, SEAL's serialisation (seal/wire.h) over TCP on 127.0.0.1, every layer kind
    reconstructing (cheetah-api.cpp:11-29,710-737).  SURVEY 8 f-1."""
    exes = {k: os.path.join(REF, f"ref_party_{k}") for k in ("fast", "slow")}
    if not all(os.path.exists(e) for e in exes.values()):
        pytest.skip("built only where the reference tree exists")
    port = 23000 + (os.getpid() * 7 + hash((alice, bob))) % 20000
    big = "1" if (alice, bob) == ("fast", "fast") else "0"
    pa = subprocess.Popen([exes[alice], "1", str(port), big], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pb = subprocess.Popen([exes[bob], "2", str(port), big], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        oa, ea = pa.communicate(timeout=250)
        ob, eb = pb.communicate(timeout=30)
    finally:
        for p in (pa, pb):
            if p.poll() is None:
                p.kill()
    print(oa[-3000:], ea[-1500:], ob[-1500:], eb[-1500:])
    assert pa.returncode == 0 and pb.returncode == 0
    assert "ALL PASSED" in oa and oa.count("PASS") >= 8 and "FAIL" not in oa and "FAIL" not in ob
