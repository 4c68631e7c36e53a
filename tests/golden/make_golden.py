#!/usr/bin/env python3
"""Regenerates tests/golden/vectors.json from the CPU oracle.

The reference ships no golden vectors for this path and cannot be built here (SEAL/HEXL are
not in /root/reference), so these vectors pin the ORACLE's behaviour, not the
reference's: they guard against silent drift of oracle or kernels between rounds.  Inputs are
seeded (splitmix64, 0x4D264D); each entry stores the first values of an output and the SHA-256
of the whole array (little-endian uint64 bytes)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.layer_fixtures import SEED, keypair, noise, rns_uniform, signed_weights, ternary  # noqa: E402
from tests.oracle_binding import A, OracleCtx, load, splitmix64, uniform_mod  # noqa: E402
from tests.oracle_layers import BnOracle, ConvOracle, FcOracle  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a)
    return {"shape": list(a.shape), "head": [int(x) for x in a.reshape(-1)[:8]],
            "sha256": hashlib.sha256(a.tobytes()).hexdigest()}


def build():
    out = {"seed": SEED}
    lib = load()
    # parameters
    o = OracleCtx(4096, (60, 49), 1 << 37)
    o8 = OracleCtx(8192, (60, 60, 60, 49), 0)
    out["params"] = {"q4096": o.q, "psi4096": o.psi, "gamma4096": o.gamma, "q8192": o8.q, "psi8192": o8.psi}
    # NTT
    a = rns_uniform(o, SEED, 2, 2)
    out["ntt4096_forward"] = digest(o.ntt_forward(a))
    a8 = np.stack([np.stack([uniform_mod(SEED + l, 8192, q) for l, q in enumerate(o8.q)])])
    out["ntt8192_forward"] = digest(o8.ntt_forward(a8))
    # BFV pieces, t = 2^37 and t = p
    for name, t in (("pow2", 1 << 37), ("prime", 4293918721)):
        c = OracleCtx(4096, (60, 49), t)
        sk, pk = keypair(c)
        m = uniform_mod(SEED + 2, 4096, t)
        ct = np.zeros((2, 2, 4096), np.uint64)
        lib.mmo_encrypt_symmetric(c.p, 0, A(sk), A(rns_uniform(c, SEED + 3, 1, 2)[0]), A(noise(SEED + 4, 4096)), A(m),
                                  4096, A(ct))
        low = np.zeros((2, 1, 4096), np.uint64)
        lib.mmo_mod_switch_to_last(c.p, A(ct), A(low), 2)
        r = rns_uniform(c, SEED + 5, 1, 2)[0]
        sr = np.zeros(4096, np.uint64)
        lib.mmo_decrypt_scale_and_round(c.p, 0, A(r), 4096, A(sr))
        out[f"bfv_{name}"] = {"pk": digest(pk), "ct": digest(ct), "mod_switch": digest(low),
                              "scale_and_round": digest(sr)}
    # conv layer (small) and the ResNet geometry
    c = OracleCtx(4096, (60, 49), 1 << 37)
    sk, pk = keypair(c)
    cv = ConvOracle(c, (5, 12, 12), (5, 3, 3), 2, 1, 1, True)
    img, s0 = uniform_mod(SEED + 20, (5, 12, 12), c.t), uniform_mod(SEED + 21, (5, 12, 12), c.t)
    s1 = (img - s0) & np.uint64(c.t - 1)
    f = signed_weights(SEED + 22, (2, 5, 3, 3), 1 << 12, c.t)
    cts = cv.encrypt_image(sk, s0, rns_uniform(c, SEED + 30, cv.n_in, 2), noise(SEED + 31, (cv.n_in, 4096)))
    oc, sh = cv.server(cts, cv.encode_image(s1), cv.encode_filters(f), np.ascontiguousarray(pk[:, :1]),
                       ternary(SEED + 40, (cv.n_out, 4096)), noise(SEED + 41, (cv.n_out, 4096)),
                       rns_uniform(c, SEED + 50, cv.n_out, 1)[:, 0])
    out["conv_5x12x12_3x3_same"] = {"filters": digest(cv.encode_filters(f)), "out_cts": digest(oc), "share": digest(sh),
                                    "decrypted": digest(cv.decrypt(np.ascontiguousarray(sk[:1]), oc))}
    # a 1x1 layer: the HIP side serves it through its coefficient-form product route
    pv = ConvOracle(c, (6, 14, 14), (6, 1, 1), 4, 0, 1, True)
    img, s0 = uniform_mod(SEED + 70, (6, 14, 14), c.t), uniform_mod(SEED + 71, (6, 14, 14), c.t)
    s1 = (img - s0) & np.uint64(c.t - 1)
    f = signed_weights(SEED + 72, (4, 6, 1, 1), 1 << 12, c.t)
    cts = pv.encrypt_image(sk, s0, rns_uniform(c, SEED + 73, pv.n_in, 2), noise(SEED + 74, (pv.n_in, 4096)))
    oc, sh = pv.server(cts, pv.encode_image(s1), pv.encode_filters(f), np.ascontiguousarray(pk[:, :1]),
                       ternary(SEED + 75, (pv.n_out, 4096)), noise(SEED + 76, (pv.n_out, 4096)),
                       rns_uniform(c, SEED + 77, pv.n_out, 1)[:, 0])
    out["conv_6x14x14_1x1"] = {"out_cts": digest(oc), "share": digest(sh),
                               "decrypted": digest(pv.decrypt(np.ascontiguousarray(sk[:1]), oc))}
    big = ConvOracle(c, (64, 56, 56), (64, 3, 3), 64, 1, 1, True)
    (th, tw), tg = big.targets(big.mask_ix, 0, 0)
    out["conv_resnet_geometry"] = {"slice_strides": list(big.slice_strides), "pads": list(big.pads), "n_in": big.n_in,
                                   "n_out": big.n_out, "targets": digest(tg)}
    fc = FcOracle(c, 1001, 2048, True)
    out["fc_1001x2048_split"] = [fc.d0, fc.d1, fc.n_in, fc.n_out]
    # conversions, both parties
    n, bw, p = 4096, 32, 4293918721
    x, y = splitmix64(SEED + 60, n) & np.uint64((1 << bw) - 1), uniform_mod(SEED + 61, n, p)
    for party in (1, 2):
        tmp, msb, corr, o_ = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        lib.mmo_r2f_pre(party, A(x), n, bw, p, A(tmp), A(msb), A(corr))
        lib.mmo_r2f_post(party, A(tmp), A(msb), A(y), n, bw, p, A(o_))
        out[f"r2f_party{party}"] = {"tmp": digest(tmp), "msb": digest(msb), "out": digest(o_)}
        lib.mmo_f2r_pre(party, A(y), n, p, 12, A(tmp), A(msb), A(corr))
        lib.mmo_f2r_trunc_post(party, A(tmp), A(msb), A(x), n, p, bw, 12, A(o_))
        out[f"f2r_trunc_party{party}"] = {"tmp": digest(tmp), "msb": digest(msb), "out": digest(o_)}
    # truncate family (truncation-cheetah.cpp), both parties, signed, with and without the msb0 heuristic
    aw = splitmix64(SEED + 62, n) & np.uint64((1 << 12) - 1)
    for party in (1, 2):
        for heur in (0, 1):
            adj, cmpv, o_ = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            lib.mmo_trunc_pre(party, A(x), n, bw, 1, heur, A(adj), A(cmpv))
            lib.mmo_trunc_post(party, A(adj), A(aw), n, 12, bw, 1, heur, A(o_))
            out[f"truncate_party{party}_heuristic{heur}"] = {"adj": digest(adj), "cmp": digest(cmpv), "out": digest(o_)}
    # COT post-processing (silent_ot.h:85-242): both sides, mod 2^37 and mod p, ragged length, gid offset 40
    from tests.test_cot_oracle import cot_pair
    for name, l_, p_ in (("ring37", 37, 0), ("prime", 0, 4293918721)):
        data0, data, _, _, wire, _ = cot_pair(lib, 4099, l_, p_, SEED + 80, gid0=40)
        out[f"cot_{name}"] = {"data0": digest(data0), "wire": digest(wire), "data": digest(data)}
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    with open(path, "w") as fh:
        json.dump(build(), fh, indent=1)
    print("wrote", path)
