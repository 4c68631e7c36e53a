"""The committed golden vectors (tests/golden/vectors.json, made by make_golden.py from the
oracle) still match what the oracle computes today."""
import json
import os

from tests.golden.make_golden import build

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_reproduces_golden_vectors():
    want = json.load(open(os.path.join(HERE, "golden", "vectors.json")))
    got = json.loads(json.dumps(build()))
    assert got == want


import hashlib

import numpy as np
import pytest


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.gpu
def test_gpu_reproduces_golden_vectors():
    """HIP path vs the committed fixtures directly: outputs are compared with the fixture digests, not with the
    oracle (the COT case at the end reuses the oracle-side helper only to regenerate its seeded inputs)."""
    import torch
    import openmm_amd
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from openmm_amd.lib import to_device, to_host
    from tests.layer_fixtures import SEED, noise, rns_uniform, signed_weights, ternary
    from tests.oracle_binding import OracleCtx, uniform_mod
    want = json.load(open(os.path.join(HERE, "golden", "vectors.json")))
    g = openmm_amd.Context(4096, (60, 49), 1 << 37)
    assert g.q == want["params"]["q4096"] and g.psi == want["params"]["psi4096"]
    assert g.gamma == want["params"]["gamma4096"]
    g8 = openmm_amd.Context(8192, (60, 60, 60, 49), 0)
    assert g8.q == want["params"]["q8192"] and g8.psi == want["params"]["psi8192"]

    class Q:  # only the moduli are needed to regenerate the seeded inputs
        N, q = 4096, g.q
    d = to_device(rns_uniform(Q, SEED, 2, 2))
    assert _sha(to_host(g.ntt_forward(d))) == want["ntt4096_forward"]["sha256"]
    a8 = np.stack([np.stack([uniform_mod(SEED + l, 8192, q) for l, q in enumerate(g8.q)])])
    assert _sha(to_host(g8.ntt_forward(to_device(a8)))) == want["ntt8192_forward"]["sha256"]
    # the small conv layer end to end, keys regenerated through the C ABI
    t, N = 1 << 37, 4096
    s = (np.array([int(x) for x in __import__("tests.oracle_binding", fromlist=["splitmix64"]).splitmix64(SEED, N)
                   % np.uint64(3)], dtype=np.int64) - 1)
    sk = np.stack([np.where(s < 0, np.uint64(q - 1), s.astype(np.uint64)) for q in g.q]).astype(np.uint64)
    d_sk = g.ntt_forward(to_device(sk))
    a_ntt = np.stack([uniform_mod(SEED + 1 + l, N, q) for l, q in enumerate(g.q)])
    d_pk = g.make_public_key(d_sk, to_device(a_ntt), torch.from_numpy(noise(SEED + 9, N)).cuda())
    meta = Meta((5, 12, 12), (5, 3, 3), 2, 1, 1, True)
    client, server = HomConv2DSS(), HomConv2DSS()
    client.setUp(g, sk_ntt=d_sk)
    server.setUp(g, pk_ntt=d_pk)
    plan = server.plan(meta)
    img, s0 = uniform_mod(SEED + 20, (5, 12, 12), t), uniform_mod(SEED + 21, (5, 12, 12), t)
    s1 = (img - s0) & np.uint64(t - 1)
    f = signed_weights(SEED + 22, (2, 5, 3, 3), 1 << 12, t)
    dev_i = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    cts = client.encryptImage(to_device(s0), meta, (to_device(rns_uniform(Q, SEED + 30, plan.n_in, 2)),
                                                    dev_i(noise(SEED + 31, (plan.n_in, N)))))
    fp = server.encodeFilters(to_device(f), meta)
    assert _sha(to_host(fp)) == want["conv_5x12x12_3x3_same"]["filters"]["sha256"]
    oc, sh = server.conv2DSS(cts, server.encodeImage(to_device(s1), meta), fp, meta,
                             (dev_i(ternary(SEED + 40, (plan.n_out, N))), dev_i(noise(SEED + 41, (plan.n_out, N))),
                              to_device(rns_uniform(Q, SEED + 50, plan.n_out, 1)[:, 0])))
    assert _sha(to_host(oc)) == want["conv_5x12x12_3x3_same"]["out_cts"]["sha256"]
    assert _sha(to_host(sh)) == want["conv_5x12x12_3x3_same"]["share"]["sha256"]
    assert _sha(to_host(client.decryptToTensor(oc, meta))) == want["conv_5x12x12_3x3_same"]["decrypted"]["sha256"]
    # the 1x1 layer from raw weights (served by the coefficient-form product route on the device)
    meta = Meta((6, 14, 14), (6, 1, 1), 4, 0, 1, True)
    plan = server.plan(meta)
    img, s0 = uniform_mod(SEED + 70, (6, 14, 14), t), uniform_mod(SEED + 71, (6, 14, 14), t)
    s1 = (img - s0) & np.uint64(t - 1)
    f = signed_weights(SEED + 72, (4, 6, 1, 1), 1 << 12, t)
    cts = client.encryptImage(to_device(s0), meta, (to_device(rns_uniform(Q, SEED + 73, plan.n_in, 2)),
                                                    dev_i(noise(SEED + 74, (plan.n_in, N)))))
    oc, sh = server.conv2DSSFromWeights(cts, server.encodeImage(to_device(s1), meta), to_device(f), meta,
                                        (dev_i(ternary(SEED + 75, (plan.n_out, N))),
                                         dev_i(noise(SEED + 76, (plan.n_out, N))),
                                         to_device(rns_uniform(Q, SEED + 77, plan.n_out, 1)[:, 0])))
    assert _sha(to_host(oc)) == want["conv_6x14x14_1x1"]["out_cts"]["sha256"]
    assert _sha(to_host(sh)) == want["conv_6x14x14_1x1"]["share"]["sha256"]
    assert _sha(to_host(client.decryptToTensor(oc, meta))) == want["conv_6x14x14_1x1"]["decrypted"]["sha256"]
    # truncate family against the committed vectors (inputs as make_golden.py seeds them)
    from openmm_amd.lib import check, dptr, load_library
    from tests.oracle_binding import splitmix64
    lib, n, bw = load_library(), 4096, 32
    x = splitmix64(SEED + 60, n) & np.uint64((1 << bw) - 1)
    aw = splitmix64(SEED + 62, n) & np.uint64((1 << 12) - 1)
    d_x, d_aw = to_device(x), to_device(aw)
    for party in (1, 2):
        for heur in (0, 1):
            adj, cmpv, o_ = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(3))
            check(lib.mmh_trunc_pre(party, dptr(d_x), n, bw, 1, heur, dptr(adj), dptr(cmpv), None))
            check(lib.mmh_trunc_post(party, dptr(adj), dptr(d_aw), n, 12, bw, 1, heur, dptr(o_), None))
            w = want[f"truncate_party{party}_heuristic{heur}"]
            assert (_sha(to_host(adj)), _sha(to_host(cmpv)), _sha(to_host(o_))) == \
                (w["adj"]["sha256"], w["cmp"]["sha256"], w["out"]["sha256"])
    # COT post-processing: the sender's data0 and wire, the receiver's data from that wire
    import ctypes as C
    from tests.test_cot_gpu import gpu_pair
    for name, l_, p_ in (("ring37", 37, 0), ("prime", 0, 4293918721)):
        _, (g_data0, g_wire, g_data) = gpu_pair(4099, l_, p_, SEED + 80, 40)
        w = want[f"cot_{name}"]
        assert (_sha(g_data0), _sha(g_wire), _sha(g_data)) == \
            (w["data0"]["sha256"], w["wire"]["sha256"], w["data"]["sha256"])
