#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on N GPUs of one node.

Step = one batched forward negacyclic NTT (N=4096, SEAL moduli {60,49} bits) over a
resident batch of `--polys` two-limb polynomials (default 16384 polys = 32768 limb-NTTs =
1 GiB, past the Infinity Cache).  Weak scaling: every rank transforms its own batch,
no collective on the data path.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_NTT = {4096: 65536, 8192: 131072}   # 2*N*8 B per limb-NTT
BUTTERFLIES = {4096: 24576, 8192: 53248}       # (N/2) log2 N
# Irreducible full-rate VALU operations per butterfly: 9 v_mad_u64_u32 for the 64-bit Shoup product of the
# 60-bit limb, 8 FP64 operations (5-op exact modular product + 3 adds) for the 49-bit limb;
# the batch is half and half.  Both instruction kinds issue at the same measured rate.
MADS_PER_BUTTERFLY = (9 + 8) / 2
MAD_PEAK = 34.5e12         # lanes/s measured on MI355X (profiles/r01_a_valu_probe.txt, r01_f64_probe.txt)


PMC_FILE = "profiles/r02_ntt_pmc.json"


def pmc_traffic(ntts_per_launch):
    """HBM bytes per launch of ntt_fwd_kernel<12>: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of
    MI355X_MICROARCH.md) from separate rocprofv3 --pmc passes over the same kernel (tools/evidence.sh ->
    profiles/r02_ntt_pmc.json, collected on 8192 limb-NTTs per launch), scaled to this launch: the RECORDED figure,
    used when measured_traffic() below cannot run; `traffic_source` in the line says which of the two it is."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)["traffic_bytes"] * ntts_per_launch / 8192
    except (OSError, KeyError, ValueError):
        return None


def measured_traffic(ntts_per_launch, limit_s=60):
    """The same figure measured in THIS run: two child `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE: the
    guide's separate passes) over tools/ntt_bench, which launches ntt_fwd_kernel<12> on 8192 limb-NTTs through the
    same library; 2 x FETCH_SIZE + WRITE_SIZE (KiB), scaled to this launch.  None when the profiler or the binary is
    missing, a pass fails or overruns its limit, or this process itself runs under a profiler.  Called before this
    process initialises the GPU (main): the passes are over by the time it does."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe, prof = os.path.join(ROOT, "tools", "ntt_bench"), shutil.which("rocprofv3")
    if not (prof and os.access(exe, os.X_OK)) or any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ):
        return None
    env = dict(os.environ, TMPDIR="/tmp",
               LD_LIBRARY_PATH=os.path.join(ROOT, "openmm_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    avg = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            child = subprocess.Popen([prof, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--",
                                      exe, "12", "4096", "5"], cwd=tmp, env=env, stdout=subprocess.DEVNULL,
                                     stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                if child.wait(timeout=limit_s) != 0:
                    return None
            except subprocess.TimeoutExpired:
                os.killpg(child.pid, signal.SIGKILL)   # the group this call started, nothing else
                child.wait()
                return None
            vals = [float(row["Counter_Value"])
                    for f in glob.glob(f"{out}/**/*_counter_collection.csv", recursive=True)
                    for row in csv.DictReader(open(f))
                    if "ntt_fwd_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not vals:
                return None
            avg[counter] = sum(vals) / len(vals)
    return (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 * ntts_per_launch / 8192


def cpu_baseline(seconds=12.0):
    """oracle (a port, not HEXL) forward NTT N=4096 on host cores, bounded sample; the AVX-512 restatement of the
    same butterflies (oracle/mm_oracle_avx512.c, 8 lanes, 64-bit products from 32-bit halves like HEXL's 64-bit
    path) where the host has it, the scalar C otherwise: `isa` says which"""
    import numpy as np
    from tests.oracle_binding import OracleCtx, A, uniform_mod
    o = OracleCtx(4096, (60, 49), 0)
    vector = bool(o.lib.mmo_set_vector_isa(1))
    threads = min(16, len(os.sched_getaffinity(0)))
    n_polys = 64 * threads
    a = np.stack([np.stack([uniform_mod(1 + l, 4096, q) for l, q in enumerate(o.q)])] * n_polys)
    o.lib.mmo_ntt_forward_batch(o.p, A(a), n_polys, 2, threads)  # warm-up
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        o.lib.mmo_ntt_forward_batch(o.p, A(a), n_polys, 2, threads)
        done += 2 * n_polys
    dt = time.perf_counter() - t0
    # the same at the reference's shipped worker count for the networks (nt = 4, scripts/common.sh:56)
    done4, t4 = 0, time.perf_counter()
    while time.perf_counter() - t4 < seconds / 4:
        o.lib.mmo_ntt_forward_batch(o.p, A(a), n_polys, 2, min(4, threads))
        done4 += 2 * n_polys
    dt4 = time.perf_counter() - t4
    return {"value": done / dt, "unit": "NTT/s", "cores": threads, "kind": "port",
            "value_nt4": done4 / dt4, "cores_nt4": min(4, threads),
            "isa": "avx512f+dq, 8 x u64 lanes" if vector else "scalar x86-64",
            "sample": f"{done} forward limb-NTTs N=4096 (oracle Harvey/Shoup C, OpenMP x{threads}) in {dt:.1f}s"}


def cpu_homconv_ms(reps=3):
    """conv2DOneFilter loop of the ResNet50 3x3 56x56x64->64 layer on the host cores, in the reference's
    own schedule (4096 multiply_plain + add_inplace, workers over filters like its ThreadPool;
    hom_conv2d_ss.cc:563-619, :679-690) through the oracle port: the dominant part of what the GPU leg
    `homconv_resnet50_3x3_56x56x64_ms` replaces (encoding and the re-sharing tail are not in this figure)."""
    import numpy as np
    from tests.oracle_binding import OracleCtx, A, uniform_mod, splitmix64
    t, N, M, Cs = 1 << 37, 4096, 64, 64
    o = OracleCtx(N, (60, 49), t)
    o.lib.mmo_set_vector_isa(1)   # the AVX-512 transforms where the host has them (same bits)
    threads = min(16, len(os.sched_getaffinity(0)))
    image = np.stack([np.stack([np.stack([uniform_mod(9 * k + 2 * j + l, N, q) for l, q in enumerate(o.q)])
                                for j in range(2)]) for k in range(Cs)])
    filters = np.zeros((M, Cs, N), dtype=np.uint64)
    pos = (splitmix64(7, M * Cs * 9) % np.uint64(N)).reshape(M, Cs, 9).astype(np.int64)
    val = (splitmix64(8, M * Cs * 9) % np.uint64(1 << 12)).reshape(M, Cs, 9) + np.uint64(1)
    np.put_along_axis(filters, pos, val, axis=2)
    out = np.zeros((M, 2, 2, N), dtype=np.uint64)
    o.lib.mmo_conv_mac_reference(o.p, A(image), A(filters), M, Cs, 1, A(out), threads)
    t0 = time.perf_counter()
    for _ in range(reps):
        o.lib.mmo_conv_mac_reference(o.p, A(image), A(filters), M, Cs, 1, A(out), threads)
    return (time.perf_counter() - t0) / reps * 1e3, threads


def prewarm(fn, torch, seconds=0.05):
    """fn repeated untimed for `seconds` (see timed(): an idle chip takes tens of ms of load to settle)"""
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def conversion_leg(torch, openmm_amd):
    """ring_to_field local share arithmetic on 10^7 elements (tests/ring_to_field_test.cpp:171), l=32,
    p=4293918721, ALICE's side: the pre kernel (in -> tmp, msb, corr) and the post kernel (tmp, msb, y -> out)
    around the host COT, device-resident; 25 algorithmic bytes per element each."""
    import openmm_amd.lib as L
    lib = L.load_library()
    n, l, p = 10_000_000, 32, 4293918721
    gen = torch.Generator(device="cuda").manual_seed(0x4D264D)
    x = torch.randint(0, 1 << l, (n,), dtype=torch.int64, device="cuda", generator=gen)
    y = torch.randint(0, p, (n,), dtype=torch.int64, device="cuda", generator=gen)
    tmp, corr, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    msb = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = L._stream_ptr()

    def both():
        L.check(lib.mmh_r2f_pre(1, L.dptr(x), n, l, p, L.dptr(tmp), L.dptr(msb), L.dptr(corr), s))
        L.check(lib.mmh_r2f_post(1, L.dptr(tmp), L.dptr(msb), L.dptr(y), n, l, p, L.dptr(out), s))

    both()
    prewarm(both, torch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        both()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    res = {"ring_to_field_1e7_pre_post_us": us, "ring_to_field_hbm_frac": 50.0 * n / (us * 1e-6) / HBM_PEAK}
    # Truncation::truncate (truncation-cheetah.cpp:2-82, no heuristic, signed, l=32, shift 12) on the same 10^7
    # elements: pre (in -> adj, cmp: 24 B) and post (adj, aw -> out: 24 B) around the host compare + B2A
    adj, cmp, aw = torch.empty_like(x), torch.empty_like(x), y & 0xFFF

    def trunc():
        L.check(lib.mmh_trunc_pre(1, L.dptr(x), n, l, 1, 0, L.dptr(adj), L.dptr(cmp), s))
        L.check(lib.mmh_trunc_post(1, L.dptr(adj), L.dptr(aw), n, 12, l, 1, 0, L.dptr(out), s))

    trunc()
    prewarm(trunc, torch)
    e0.record()
    for _ in range(20):
        trunc()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    res.update({"truncate_1e7_pre_post_us": us, "truncate_hbm_frac": 48.0 * n / (us * 1e-6) / HBM_PEAK})
    # COT post-processing of the same 10^7 elements (silent_ot.h:129-163,:209-242, mod p): the sender's launch
    # hashes two blocks per element, the receiver's one; blocks, choices and the wire resident in HBM
    import ctypes as C
    rcm = torch.randint(-(1 << 62), 1 << 62, (n, 2), dtype=torch.int64, device="cuda", generator=gen)
    choice = (x & 1).to(torch.uint8)
    two = (C.c_uint64 * 2)
    delta, seed = two(0x0123456789ABCDEF, 0xFEDCBA9876543211), two(0x1122334455667788, 0x99AABBCCDDEEFF00)
    for prime, key in ((1, "prime"), (0, "ring")):
        nb = C.c_size_t()
        L.check(lib.mmh_cot_wire_bytes(n, l, p, prime, C.byref(nb)))
        wire = torch.empty((nb.value + 7) // 8, dtype=torch.int64, device="cuda")

        def cot():
            L.check(lib.mmh_cot_send(L.dptr(rcm), delta, seed, 0, L.dptr(corr), n, l, p, prime, L.dptr(tmp),
                                     L.dptr(wire), s))
            L.check(lib.mmh_cot_recv(L.dptr(rcm), L.dptr(choice), seed, 0, L.dptr(wire), n, l, p, prime, L.dptr(out),
                                     s))
        cot()
        prewarm(cot, torch)
        e0.record()
        for _ in range(10):
            cot()
        e1.record()
        torch.cuda.synchronize()
        res[f"cot_{key}_1e7_send_recv_us"] = e0.elapsed_time(e1) / 10 * 1e3
    # ALICE's whole ring_to_field with the COT post-processing on the device: pre -> cot_send (mod p) -> post,
    # operands resident (what is left on the host is Ferret's output and the 4 B per element of wire)
    nb = C.c_size_t()
    L.check(lib.mmh_cot_wire_bytes(n, l, p, 1, C.byref(nb)))
    wire = torch.empty((nb.value + 7) // 8, dtype=torch.int64, device="cuda")
    y0 = torch.empty_like(x)

    def chain():
        L.check(lib.mmh_r2f_pre(1, L.dptr(x), n, l, p, L.dptr(tmp), L.dptr(msb), L.dptr(corr), s))
        L.check(lib.mmh_cot_send(L.dptr(rcm), delta, seed, 0, L.dptr(corr), n, l, p, 1, L.dptr(y0), L.dptr(wire), s))
        L.check(lib.mmh_r2f_post(1, L.dptr(tmp), L.dptr(msb), L.dptr(y0), n, l, p, L.dptr(out), s))
    chain()
    prewarm(chain, torch)
    e0.record()
    for _ in range(10):
        chain()
    e1.record()
    torch.cuda.synchronize()
    res["ring_to_field_1e7_sender_with_device_cot_us"] = e0.elapsed_time(e1) / 10 * 1e3
    # three AES-128 encryptions + two key schedules per element pair of launches
    res["cot_aes_blocks_per_s"] = 3.0 * n / (res["cot_prime_1e7_send_recv_us"] * 1e-6)
    return res


def layer_legs(torch, openmm_amd, local):
    """Secondary legs of BASELINE.json's metric (configs[2]): server-side HomConv of the ResNet50
    3x3 56x56x64->64 layer and BOLE on 2^20 elements, ms per layer, inputs resident in HBM."""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta, SAME
    from openmm_amd.hom_bn_ss import HomBNSS, Meta as BnMeta
    out = {}
    gen = torch.Generator(device="cuda").manual_seed(0x4D264D)
    rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)

    def ms(fn, reps=20):
        fn()
        prewarm(fn, torch, 0.03)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t, N = 1 << 37, 4096
    ctx = openmm_amd.Context(N, (60, 49), t, device=local)
    qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda").view(1, 2, 1)
    meta = Meta((64, 56, 56), (64, 3, 3), 64, SAME, 1, True)
    srv = HomConv2DSS()
    srv.setUp(ctx, pk_ntt=rnd(0, 1 << 62, 2, 2, N) % qs)
    plan = srv.plan(meta)
    cts = rnd(0, 1 << 62, plan.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
    share = rnd(0, t, 64, 56, 56)
    filters = (rnd(-(1 << 12) + 1, 1 << 12, 64, 64, 3, 3) | 1) & (t - 1)
    rand = (rnd(-1, 2, plan.n_out, N), rnd(-8, 9, plan.n_out, N), rnd(0, 1 << 62, plan.n_out, N) % ctx.q[0])

    def conv_layer():  # what CheetahLinear::conv2d runs on the server per layer (cheetah-api.cpp:417-445)
        return srv.conv2DSSFromTensors(cts, share, filters, meta, rand)

    out["homconv_resnet50_3x3_56x56x64_ms"] = ms(conv_layer)
    out["homconv_reference_schedule_ntts"] = 40960
    p = 4293918721
    bctx = openmm_amd.Context(N, (60, 49), p, device=local)
    bn = HomBNSS()
    bn.setUp_Field(p, bctx)
    n = 1 << 20
    bmeta = BnMeta((n,), p, True)
    ex, ey = (rnd(0, 1 << 62, n // N, 2, 2, N) % qs.view(1, 1, 2, 1) for _ in range(2))
    x1, y1, rr = rnd(0, 1 << 32, n), rnd(0, 1 << 32, n), rnd(0, 1 << 62, n)

    def bole_layer():  # ALICE branch of CheetahLinear::BOLE (cheetah-api.cpp:627-645)
        return bn.bole(ex, ey, bn.encodeVector(x1, bmeta), bn.encodeVector(y1, bmeta), bmeta, rr)

    out["bole_2p20_server_ms"] = ms(bole_layer)
    return out


# ResNet50 linear layers (from networks/main_resnet50.cpp:4981-6333):
# (C, H, W, fh, fw, M, SAME?, stride, count).  VALID layers arrive pre-padded from the network code.
RESNET50_CONVS = [
    (3, 230, 230, 7, 7, 64, 0, 2, 1), (64, 56, 56, 1, 1, 256, 0, 1, 4), (64, 56, 56, 1, 1, 64, 0, 1, 1),
    (64, 56, 56, 3, 3, 64, 1, 1, 3), (256, 56, 56, 1, 1, 64, 0, 1, 2), (256, 56, 56, 1, 1, 512, 0, 2, 1),
    (256, 56, 56, 1, 1, 128, 0, 1, 1), (128, 58, 58, 3, 3, 128, 0, 2, 1), (128, 28, 28, 1, 1, 512, 0, 1, 4),
    (512, 28, 28, 1, 1, 128, 0, 1, 3), (128, 28, 28, 3, 3, 128, 1, 1, 3), (512, 28, 28, 1, 1, 1024, 0, 2, 1),
    (512, 28, 28, 1, 1, 256, 0, 1, 1), (256, 30, 30, 3, 3, 256, 0, 2, 1), (256, 14, 14, 1, 1, 1024, 0, 1, 6),
    (1024, 14, 14, 1, 1, 256, 0, 1, 5), (256, 14, 14, 3, 3, 256, 1, 1, 5), (1024, 14, 14, 1, 1, 2048, 0, 2, 1),
    (1024, 14, 14, 1, 1, 512, 0, 1, 1), (512, 16, 16, 3, 3, 512, 0, 2, 1), (512, 7, 7, 1, 1, 2048, 0, 1, 3),
    (2048, 7, 7, 1, 1, 512, 0, 1, 2), (512, 7, 7, 3, 3, 512, 1, 1, 2),
]


# SqueezeNet (BASELINE.json configs[3]; networks/main_sqnet.cpp:3107-3764): 26 convolutions, no FC.
SQUEEZENET_CONVS = [
    (3, 227, 227, 3, 3, 64, 0, 2, 1), (64, 56, 56, 1, 1, 16, 0, 1, 1), (16, 56, 56, 1, 1, 64, 0, 1, 2),
    (16, 56, 56, 3, 3, 64, 1, 1, 2), (128, 56, 56, 1, 1, 16, 0, 1, 1), (128, 27, 27, 1, 1, 32, 0, 1, 1),
    (32, 27, 27, 1, 1, 128, 0, 1, 2), (32, 27, 27, 3, 3, 128, 1, 1, 2), (256, 27, 27, 1, 1, 32, 0, 1, 1),
    (256, 13, 13, 1, 1, 48, 0, 1, 1), (48, 13, 13, 1, 1, 192, 0, 1, 2), (48, 13, 13, 3, 3, 192, 1, 1, 2),
    (384, 13, 13, 1, 1, 48, 0, 1, 1), (384, 13, 13, 1, 1, 64, 0, 1, 1), (64, 13, 13, 1, 1, 256, 0, 1, 2),
    (64, 13, 13, 3, 3, 256, 1, 1, 2), (512, 13, 13, 1, 1, 64, 0, 1, 1), (512, 13, 13, 1, 1, 1000, 0, 1, 1),
]


def densenet121_convs():
    """DenseNet121 (BASELINE.json configs[4]; networks/main_densenet121.cpp, 121 Conv2DWrapper / FusedBN calls):
    stem, four dense blocks of (1x1 C -> 128, 3x3 128 -> 32) pairs with C growing by 32, three 1x1
    transitions and the 1x1 classifier on the pooled 1x1 image."""
    convs = [(3, 224, 224, 7, 7, 64, 1, 2, 1)]
    for hw, c0, layers, trans in ((56, 64, 6, 128), (28, 128, 12, 256), (14, 256, 24, 512), (7, 512, 16, None)):
        convs += [(c0 + 32 * k, hw, hw, 1, 1, 128, 0, 1, 1) for k in range(layers)]
        convs.append((128, hw, hw, 3, 3, 32, 1, 1, layers))
        if trans:
            convs.append((c0 + 32 * layers, hw, hw, 1, 1, trans, 0, 1, 1))
    convs.append((1024, 1, 1, 1, 1, 1000, 0, 1, 1))
    return convs


LAYER_PREWARM_S = 0.03


def resnet50_leg(torch, dist, openmm_amd, local, world, rank, reps=20, convs=None, prefix="resnet50", with_fc=True,
                 all_modes=True, prepared=True):
    """Second half of BASELINE.json's metric: server-side HE compute of all 53 ResNet50 conv layers
    + FC 2048->1001 (what CheetahLinear::conv2d / ::fc run between receiving the client's
    ciphertexts and sending the result, cheetah-api.cpp:417-445, :520-548): filter encoding and
    lift, image-share encoding, multiply-accumulate, mod-switch, re-share, truncation.  Output
    filters are sharded over the ranks (hom_conv2d_ss.cc:679-690) and gathered on rank 0; each
    layer's time is the max over ranks, barrier to barrier.  FC runs replicated (520 products)."""
    from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta
    from openmm_amd.hom_fc_ss import HomFCSS, Meta as FcMeta
    from openmm_amd.sharding import ShardedConv
    t, N = 1 << 37, 4096
    ctx = openmm_amd.Context(N, (60, 49), t, device=local)
    gen = torch.Generator(device="cuda").manual_seed(0x4D264D)  # same inputs on every rank
    rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)
    qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda")
    pk = rnd(0, 1 << 62, 2, 2, N) % qs.view(1, 2, 1)

    def wall_ms(fn, prewarm_s=LAYER_PREWARM_S, reps=reps):
        """per-call wall time in steady state: the call repeated untimed for prewarm_s first (the same count on every
        rank: the sharded forms hold a collective), since an idle MI355X needs tens of ms of load to settle (timed()
        above; one layer call: 0.219 ms over 30 calls from idle, 0.199 over 300, 0.195 over 1000)"""
        fn()
        fn()   # warm-up: the scratch cache and torch's allocator have their blocks by now
        torch.cuda.synchronize()
        if prewarm_s > 0:
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            one = max(time.perf_counter() - t0, 1e-5)
            n = int(max_over_ranks(float(min(2000, max(1, round(prewarm_s / one)))), torch, dist, world))
            for i in range(n):
                fn()
                if i % 8 == 7:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return max_over_ranks((time.perf_counter() - t0) / reps * 1e3, torch, dist, world)

    srv = HomConv2DSS()
    srv.setUp(ctx, pk_ntt=pk)
    per_layer, total, cached_total, d2h_total, repl_total, online_total, offline_total = [], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    cold_total = 0.0
    for (c, h, w, fh, fw, m, same, stride, count) in (convs or RESNET50_CONVS):
        shared = c != 3  # first conv runs with kIsSharedInput=false (main_resnet50.cpp:4979)
        meta = Meta((c, h, w), (c, fh, fw), m, same, stride, shared)
        plan = srv.plan(meta)
        cts = rnd(0, 1 << 62, plan.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
        share = rnd(0, t, c, h, w)
        filters = (rnd(-(1 << 12) + 1, 1 << 12, m, c, fh, fw) | 1) & (t - 1)
        n_out = plan.n_out
        rand = (rnd(-1, 2, n_out, N), rnd(-8, 9, n_out, N), rnd(0, 1 << 62, n_out, N) % ctx.q[0])

        def layer():   # encodeImage + encodeFilters + conv2DSS in one device call
            return srv.conv2DSSFromTensors(cts, share if shared else None, filters, meta, rand)

        if world == 1 and not prepared:
            ms = wall_ms(layer)
        elif world == 1:
            if with_fc:   # (the ResNet50 leg) the same call timed from a chip that has only seen two warm-up calls
                cold_total += count * wall_ms(layer, prewarm_s=0, reps=5)
            ms = wall_ms(layer)
            # static weights prepared once per model (not what the reference does: reported apart)
            prep = srv.prepareFilters(filters, meta)
            cached_total += count * wall_ms(lambda: srv.conv2DSSPrepared(
                cts, None, prep, meta, rand, share_tensor=share if shared else None))
            # ... and with the input-independent encryptions of zero of the re-sharing made ahead as well (offline
            # phase; fresh ones are needed per inference, their making is timed apart)
            zeros = srv.prepareZeros(meta, rand[0])
            online_total += count * wall_ms(lambda: srv.conv2DSSOnline(cts, share if shared else None, prep, zeros,
                                                                       meta, rand[1:]))
            offline_total += count * wall_ms(lambda: srv.prepareZeros(meta, rand[0]))
            del prep, zeros
        else:
            # output filters sharded over the ranks, one-call form per rank, assembled on rank 0 by (a) one RCCL
            # gather into a pre-sized device buffer, (b) per-rank async D2H into one shared page-locked host buffer
            times = {}
            for mode in (("gather", "d2h") if all_modes else ("gather",)):
                sh = ShardedConv(srv, meta, mode=mode)
                lo = sh.lo * plan.n_spatial
                times[mode] = wall_ms(lambda: sh.run(cts, share if shared else None, filters,
                                                     lambda k: tuple(r[lo:lo + k] for r in rand)))
                del sh
            ms = times["gather"]
            if all_modes:
                d2h_total += times["d2h"] * count
                # (c) replicated: every rank runs whole layers of its own inference (no sharding, no exchange)
                repl_total += wall_ms(layer) * count
        per_layer.append(round(ms, 3))
        total += ms * count
        del cts, share, filters, rand
    multi = {}
    if world > 1 and all_modes:
        multi = {f"{prefix}_he_linear_d2h_ms": d2h_total,
                 f"{prefix}_replicated_ms_per_inference": repl_total,
                 f"{prefix}_replicated_inferences_per_s": world / (repl_total / 1e3),
                 f"{prefix}_modes": "he_linear_ms: filters sharded, RCCL gather to rank 0 (device); d2h_ms: filters "
                                    "sharded, per-rank async copy into one shared pinned host buffer; replicated: one "
                                    "independent inference per rank (FC excluded)"}
    if not with_fc:
        out = {f"{prefix}_he_linear_ms": total, f"{prefix}_conv_layers": sum(x[-1] for x in convs),
               f"{prefix}_conv_ms_per_shape": per_layer}
        if world == 1:
            out[f"{prefix}_he_linear_prepared_filters_ms"] = cached_total
            if prepared:
                out[f"{prefix}_he_linear_online_ms"] = online_total
                out[f"{prefix}_he_linear_offline_zeros_ms"] = offline_total
        out.update(multi)
        return out
    fc = HomFCSS()
    fc.setUp(ctx, pk_ntt=pk)
    fmeta = FcMeta((1001, 2048), (2048,), True)
    fplan = fc.plan(fmeta)
    vec_ct = rnd(0, 1 << 62, fplan.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
    vec, wmat = rnd(0, t, 2048), (rnd(-(1 << 12) + 1, 1 << 12, 1001, 2048) | 1) & (t - 1)
    frand = (rnd(-1, 2, fplan.n_out, N), rnd(-8, 9, fplan.n_out, N), rnd(0, 1 << 62, fplan.n_out, N) % ctx.q[0])
    fc_ms = wall_ms(lambda: fc.matVecMul(fc.encodeWeightMatrix(wmat, fmeta), vec_ct,
                                         fc.encodeInputVector(vec, fmeta), fmeta, frand))
    out = {"resnet50_he_linear_ms": total + fc_ms, "resnet50_conv_layers": 53, "resnet50_fc_ms": round(fc_ms, 3),
           "resnet50_conv_ms_per_shape": per_layer, "resnet50_sharding": f"output filters x{world}, gather on rank 0" if world > 1 else "none"}
    out["layer_prewarm_ms"] = LAYER_PREWARM_S * 1e3
    if world == 1:
        if cold_total:
            out["resnet50_he_linear_from_idle_ms"] = cold_total + fc_ms
            out["resnet50_he_linear_from_idle_note"] = (
                "every layer shape timed over 5 calls behind 2 warm-up calls only, as in earlier rounds: the chip has not "
                "settled and the window's start and drain weigh 2-3 % (resnet50_he_linear_ms pre-warms each shape for "
                "layer_prewarm_ms and times 20 calls)")
        out["resnet50_he_linear_prepared_filters_ms"] = cached_total + fc_ms
        if prepared:
            out["resnet50_he_linear_online_ms"] = online_total + fc_ms
            out["resnet50_he_linear_offline_zeros_ms"] = offline_total
            out["resnet50_online_note"] = ("online: static weights prepared per model AND the input-independent "
                                           "pk-encryptions of zero of the re-sharing made ahead of the client's "
                                           "ciphertexts (mmh_reshare_zero_prepare, fresh per inference: "
                                           "offline_zeros_ms); the reference does both inside conv2DSS on every call, "
                                           "which resnet50_he_linear_ms times (FC not split)")
    out.update(multi)
    if "resnet50_he_linear_d2h_ms" in out:
        out["resnet50_he_linear_d2h_ms"] += fc_ms
    return out


def host_to_host_leg():
    """The C++ host mirror end to end (tests/cpp/network_linear.cc): every ResNet50 convolution + the FC through
    Conv2DWrapper / MatMul2D on host arrays, both parties as two threads of one process on this GPU: encryption,
    ciphertexts through an in-process byte pipe (the stand-in for the network), HE compute, decryption.  Warm pass."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "network_linear")
    if not os.path.exists(exe):
        return {}
    r = subprocess.run([exe, "resnet50", "8"], capture_output=True, text=True, timeout=280)
    m = re.search(r"ResNet50: 53 convolutions.*?([0-9.]+) ms in total", r.stdout)
    w = re.search(r"in-process transport.*?: ([0-9.]+) ms for ([0-9.]+) MiB", r.stdout)
    if r.returncode != 0 or not m or not w:
        return {"resnet50_host_to_host_error": (r.stdout + r.stderr)[-300:]}
    return {"resnet50_host_to_host_ms": float(m.group(1)), "resnet50_host_to_host_wire_mib": float(w.group(2)),
            "resnet50_host_to_host_pipe_copy_ms": float(w.group(1)),
            "resnet50_host_to_host_note": "C++ host mirror, both parties in one process on one GPU, ciphertexts through "
                                          "an in-process byte pipe; warm pass (tests/cpp/network_linear.cc)"}


def max_over_ranks(x, torch, dist, world):
    if world == 1:
        return x
    dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
    w = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return float(w.item())


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, from this process,
    which has not touched the GPU (no torch import yet); rank 0 prints the line.  Under torch.distributed.run the
    ranks already exist (WORLD_SIZE is set) and this is not used."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


PREWARM_S = 0.25   # see timed()


def timed(fn, steps, warmup, torch, dist, world, prewarm_s=0.0):
    """W untimed warm-up steps, then exactly K timed steps between barrier + synchronize on both sides.
    prewarm_s: the step repeated untimed for that long BEFORE the warm-up steps.  An idle MI355X takes tens of
    milliseconds of load to settle (same launch: 662 us after 3 warm-up launches, 554 us after 50 or more),
    which no small W covers; the timed region is the same K steps either way, and the line reports the pre-warm."""
    if prewarm_s > 0:
        t_end = time.perf_counter() + prewarm_s
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall = max_over_ranks(wall, torch, dist, world)
    return wall, ev0.elapsed_time(ev1) / 1e3 / steps  # job wall seconds, per-launch device seconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--polys", type=int, default=16384, help="two-limb polynomials per GPU per step")
    ap.add_argument("--headline-only", action="store_true", help="only the timed NTT loop (for rocprofv3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo rehearses N ranks on fewer GPUs (rank r uses device r %% device_count)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python bench.py --gpus N starts them itself)")

    # The PMC passes come first, as child processes that have ended before this process touches the GPU at all
    # (nothing is started from a process that holds a GPU context).  The profiled runs (--no-cpu, --headline-only)
    # and multi-rank runs keep the recorded figure.
    live_traffic = None
    if world == 1 and not args.no_cpu and not args.headline_only:
        live_traffic = measured_traffic(args.polys * 2)

    import torch
    import torch.distributed as dist
    import openmm_amd

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local >= n_dev:
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {n_dev} visible); --backend gloo rehearses")
    local %= max(n_dev, 1)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    ctx = openmm_amd.Context(4096, (60, 49), 0, device=local)
    gen = torch.Generator(device="cuda").manual_seed(0x4D264D + rank)
    qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda").view(1, 2, 1)
    data = torch.randint(0, 2 ** 62, (args.polys, 2, 4096), dtype=torch.int64, device="cuda",
                         generator=gen) % qs
    ntts_per_step = args.polys * 2

    # first the same W + K launches with no pre-warm, from a chip that has only seen the data being made: reported beside
    # the steady-state figure (from_idle), which is what earlier rounds' lines held
    idle_wall, idle_launch_s = timed(lambda: ctx.ntt_forward(data), args.steps, args.warmup, torch, dist, world)
    wall, launch_s = timed(lambda: ctx.ntt_forward(data), args.steps, args.warmup, torch, dist, world,
                           prewarm_s=PREWARM_S)
    value = world * ntts_per_step * args.steps / wall
    achieved = ntts_per_step * BYTES_PER_NTT[4096] / launch_s

    extra = {}
    if rank == 0 and not args.headline_only:
        _, inv_s = timed(lambda: ctx.ntt_inverse(data), max(5, args.steps // 5), 2, torch, None, 1, prewarm_s=PREWARM_S / 2)
        extra["intt4096_per_s"] = ntts_per_step / inv_s
        c8 = openmm_amd.Context(8192, (60, 60, 60, 49), 0, device=local)
        q8 = torch.tensor(c8.q, dtype=torch.int64, device="cuda").view(1, 4, 1)
        d8 = torch.randint(0, 2 ** 62, (args.polys // 4, 4, 8192), dtype=torch.int64, device="cuda",
                           generator=gen) % q8
        _, f8 = timed(lambda: c8.ntt_forward(d8), max(5, args.steps // 5), 2, torch, None, 1, prewarm_s=PREWARM_S / 2)
        _, i8 = timed(lambda: c8.ntt_inverse(d8), max(5, args.steps // 5), 2, torch, None, 1, prewarm_s=PREWARM_S / 2)
        extra["ntt8192_per_s"] = d8.shape[0] * 4 / f8
        extra["intt8192_per_s"] = d8.shape[0] * 4 / i8
        extra["ntt8192_hbm_frac"] = d8.shape[0] * 4 * BYTES_PER_NTT[8192] / f8 / HBM_PEAK
        del d8, c8
        if world == 1:
            extra.update(layer_legs(torch, openmm_amd, local))
            extra.update(conversion_leg(torch, openmm_amd))

    if not args.headline_only:
        net = resnet50_leg(torch, dist, openmm_amd, local, world, rank)
        sq = resnet50_leg(torch, dist, openmm_amd, local, world, rank, convs=SQUEEZENET_CONVS, prefix="squeezenet",
                          with_fc=False, all_modes=False)
        dn = resnet50_leg(torch, dist, openmm_amd, local, world, rank, reps=10, convs=densenet121_convs(),
                          prefix="densenet121", with_fc=False, all_modes=False)
        dn.pop("densenet121_conv_ms_per_shape")
        if rank == 0:
            extra.update(net)
            extra.update(sq)
            extra.update(dn)
            if world == 1:
                torch.cuda.synchronize()
                extra.update(host_to_host_leg())
    if rank == 0:
        traffic = live_traffic
        source = ("measured at the start of this run: rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/ntt_bench, "
                  "8192 limb-NTTs per launch, scaled to this launch")
        if traffic is None:
            traffic = pmc_traffic(ntts_per_step)
            source = f"recorded, not measured in this run: {PMC_FILE} (tools/evidence.sh, round 2)"
        line = {
            "metric": "64-bit NTTs/sec (N=4096/8192) + ResNet50 HE-linear ms at 1/2/4/8 GPUs",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_ms": PREWARM_S * 1e3, "ms_per_step": wall / args.steps * 1e3,
            "from_idle": {"value": world * ntts_per_step * args.steps / idle_wall,
                          "ms_per_step": idle_wall / args.steps * 1e3, "launch_us": idle_launch_s * 1e6,
                          "note": "the same W warm-up and K timed steps with no pre-warm, run first"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "batched forward negacyclic NTT N=4096, q={60,49}-bit SEAL moduli "
                                   "(BASELINE.json configs[1])",
                       "polys_per_gpu": args.polys, "limbs": 2, "ntts_per_step_per_gpu": ntts_per_step,
                       "parallelism": f"independent batches x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": source,
                         "kernel": "ntt_fwd_kernel<12>", "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": ntts_per_step * BYTES_PER_NTT[4096],
                         "int64_valu": {"achieved": ntts_per_step * BUTTERFLIES[4096] * MADS_PER_BUTTERFLY
                                        / launch_s / 1e12, "peak": MAD_PEAK / 1e12, "unit": "T full-rate VALU ops/s (v_mad_u64_u32 | FP64)",
                                        "frac": ntts_per_step * BUTTERFLIES[4096] * MADS_PER_BUTTERFLY
                                        / launch_s / MAD_PEAK}},
            "extra": extra,
        }
        if world == 1 and not args.no_cpu and not args.headline_only:
            line["cpu_baseline"] = cpu_baseline()
            ms, threads = cpu_homconv_ms()
            extra["homconv_resnet50_3x3_cpu_port_ms"] = ms
            extra["homconv_resnet50_3x3_cpu_port_note"] = (
                f"conv2DOneFilter loop only, in the REFERENCE'S schedule (4096 multiply_plain, three transforms each), "
                f"oracle port (not HEXL) on {threads} host threads: the GPU leg runs another schedule, so the ratio of "
                f"the two is not a like-for-like speed-up")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
