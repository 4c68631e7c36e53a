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


PMC_FILE = "profiles/r03_ntt_pmc.json"


def pmc_traffic(ntts_per_launch):
    """HBM bytes per launch of ntt_fwd_kernel<12>: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of
    MI355X_MICROARCH.md) from separate rocprofv3 --pmc passes over the same kernel (tools/evidence.sh ->
    profiles/r03_ntt_pmc.json, collected on 8192 limb-NTTs per launch), scaled to this launch: the RECORDED figure,
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


def resnet50_order():
    """Every linear call of one ResNet50 inference in the order networks/main_resnet50.cpp:4981-6333 makes them with
    USE_FUSED_BN 1 (:11): ("conv", (C, H, W, fh, fw, M, SAME?, stride)) for Conv2DWrapper and for FusedBN (a convolution
    whose batch-norm is folded into its filters, :1728-1788), ("bn", H*W*C) for the 18 FusedBatchNorm4411 that stay live
    (-> ElemWiseActModelVectorMult, :1060-1093; 5,820,416 elements in all), ("fc", (1001, 2048)) for the MatMul2D."""
    c = lambda *a: ("conv", a)
    bn = lambda h, w, ch: ("bn", h * w * ch)
    seq = [c(3, 230, 230, 7, 7, 64, 0, 2), bn(56, 56, 64),
           c(64, 56, 56, 1, 1, 256, 0, 1), c(64, 56, 56, 1, 1, 64, 0, 1), c(64, 56, 56, 3, 3, 64, 1, 1),
           c(64, 56, 56, 1, 1, 256, 0, 1), bn(56, 56, 256)]
    for _ in range(2):
        seq += [c(256, 56, 56, 1, 1, 64, 0, 1), c(64, 56, 56, 3, 3, 64, 1, 1), c(64, 56, 56, 1, 1, 256, 0, 1),
                bn(56, 56, 256)]
    seq += [c(256, 56, 56, 1, 1, 512, 0, 2), c(256, 56, 56, 1, 1, 128, 0, 1), c(128, 58, 58, 3, 3, 128, 0, 2),
            c(128, 28, 28, 1, 1, 512, 0, 1), bn(28, 28, 512),
            c(512, 28, 28, 1, 1, 128, 0, 1), c(128, 28, 28, 3, 3, 128, 1, 1), bn(28, 28, 128),   # (:5376-5383: not folded)
            c(128, 28, 28, 1, 1, 512, 0, 1), bn(28, 28, 512)]
    for _ in range(2):
        seq += [c(512, 28, 28, 1, 1, 128, 0, 1), c(128, 28, 28, 3, 3, 128, 1, 1), c(128, 28, 28, 1, 1, 512, 0, 1),
                bn(28, 28, 512)]
    seq += [c(512, 28, 28, 1, 1, 1024, 0, 2), c(512, 28, 28, 1, 1, 256, 0, 1), c(256, 30, 30, 3, 3, 256, 0, 2),
            c(256, 14, 14, 1, 1, 1024, 0, 1), bn(14, 14, 1024)]
    for _ in range(5):
        seq += [c(1024, 14, 14, 1, 1, 256, 0, 1), c(256, 14, 14, 3, 3, 256, 1, 1), c(256, 14, 14, 1, 1, 1024, 0, 1),
                bn(14, 14, 1024)]
    seq += [c(1024, 14, 14, 1, 1, 2048, 0, 2), c(1024, 14, 14, 1, 1, 512, 0, 1), c(512, 16, 16, 3, 3, 512, 0, 2),
            c(512, 7, 7, 1, 1, 2048, 0, 1), bn(7, 7, 2048)]
    for _ in range(2):
        seq += [c(2048, 7, 7, 1, 1, 512, 0, 1), c(512, 7, 7, 3, 3, 512, 1, 1), c(512, 7, 7, 1, 1, 2048, 0, 1),
                bn(7, 7, 2048)]
    seq.append(("fc", (1001, 2048)))
    return seq


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


def densenet121_order():
    """Every linear call of one DenseNet121 inference in the order of networks/main_densenet121.cpp:9704-11894: per
    dense layer FusedBatchNorm4411 on the concatenated features (live: 63 calls, 10,649,856 elements, no USE_FUSED_BN
    escape for them), the 1x1 bottleneck (its own batch-norm folded: FusedBN) and the 3x3 growth convolution; one
    bottleneck (:10074-10081, 28x28, 224 channels) keeps its batch-norm as a separate call."""
    c = lambda *a: ("conv", a)
    seq = [c(3, 224, 224, 7, 7, 64, 1, 2)]
    for hw, c0, layers, trans in ((56, 64, 6, 128), (28, 128, 12, 256), (14, 256, 24, 512), (7, 512, 16, None)):
        for k in range(layers):
            ch = c0 + 32 * k
            seq += [("bn", hw * hw * ch), c(ch, hw, hw, 1, 1, 128, 0, 1)]
            if hw == 28 and ch == 224:
                seq.append(("bn", hw * hw * 128))
            seq.append(c(128, hw, hw, 3, 3, 32, 1, 1))
        seq.append(("bn", hw * hw * (c0 + 32 * layers)))
        if trans:
            seq.append(c(c0 + 32 * layers, hw, hw, 1, 1, trans, 0, 1))
    seq.append(c(1024, 1, 1, 1, 1, 1000, 0, 1))
    return seq


def squeezenet_order():
    """SqueezeNet has no live batch-norm and its calls are not interleaved with anything linear: the shape table in
    network order (networks/main_sqnet.cpp:3107-3764 walks fire modules squeeze -> expand 1x1, expand 3x3)"""
    return [("conv", sh[:8]) for sh in SQUEEZENET_CONVS for _ in range(sh[8])]


def check_orders():
    """the in-order lists hold exactly the calls of the shape tables (CPU-checkable: tests/test_bench_contract.py)"""
    from collections import Counter
    for order, table, n_bn, bn_elems in ((resnet50_order(), RESNET50_CONVS, 18, 5820416),
                                         (densenet121_order(), densenet121_convs(), 63, 10649856),
                                         (squeezenet_order(), SQUEEZENET_CONVS, 0, 0)):
        got = Counter(a for k, a in order if k == "conv")
        want = Counter()
        for sh in table:
            want[tuple(sh[:8])] += sh[8]
        assert got == want, (got - want, want - got)
        bns = [a for k, a in order if k == "bn"]
        assert len(bns) == n_bn and sum(bns) == bn_elems, (len(bns), sum(bns))
    return True


LAYER_PREWARM_S = 0.03
RING_BITS, K_SCALE = 37, 12                      # shipped ell / k (scripts/common.sh:48-56)
FIELD_PRIME = 137438822401                       # sci::default_prime_mod.at(37), SCI/src/utils/constants.h:47
I8_MAC_PEAK = 2.5e15                             # dense i8 MAC/s, MI355X_MICROARCH.md


class NetworkLegs:
    """Server-side (ALICE) device work of the linear calls of a network, inputs resident in HBM, as the reference runs
    them per inference: CheetahLinear::conv2d (cheetah-api.cpp:417-445: encodeFilters, encodeImage, conv2DSS), ::fc
    (:343-373) and, for every live batch-norm, ElemWiseActModelVectorMult (library_fixed_uniform_cheetah.cpp:567-600):
    Ring2Field on her share (pre step, COT post-processing as the sender, post step fused into encodeVector),
    CheetahLinear::bn (:560-602: encodeScales, bn + addMask) and Field2Ring_with_Truncate on her output share -- the same
    work in five launches (mmh_bn_alice_prepare / mmh_bn_alice_finish: her share is -r, so its conversion needs no
    ciphertext and runs beside Ring2Field; bit for bit the step-by-step calls, tests/test_fc_bn_gpu.py).
    EVERY call draws its own randomness on the device inside the call (u, e1 and the mask for each output ciphertext
    as conv2DSS / matVecMul do at hom_conv2d_ss.cc:275-340,180-223; addMask's r, hom_bn_ss.cc:329-334)."""

    def __init__(self, torch, openmm_amd, local):
        from openmm_amd.hom_bn_ss import HomBNSS
        from openmm_amd.hom_conv2d_ss import HomConv2DSS
        from openmm_amd.hom_fc_ss import HomFCSS
        import openmm_amd.lib as L
        self.torch, self.L, self.lib = torch, L, L.load_library()
        self.t, self.N = 1 << RING_BITS, 4096
        self.ctx = openmm_amd.Context(self.N, (60, 49), self.t, device=local)
        self.fctx = openmm_amd.Context(self.N, (60, 49), FIELD_PRIME, device=local)
        self.gen = torch.Generator(device="cuda").manual_seed(0x4D264D)  # same inputs on every rank
        self.qs = torch.tensor(self.ctx.q, dtype=torch.int64, device="cuda")
        pk = self.rnd(0, 1 << 62, 2, 2, self.N) % self.qs.view(1, 2, 1)
        self.sampler = L.Sampler(self.ctx)
        self.srv = HomConv2DSS()
        self.srv.setUp(self.ctx, pk_ntt=pk)
        self.fc = HomFCSS()
        self.fc.setUp(self.ctx, pk_ntt=pk)
        self.bn = HomBNSS()
        self.bn.setUp_Field(FIELD_PRIME, self.fctx)
        self.gid = 0

    def rnd(self, lo, hi, *shape):
        return self.torch.randint(lo, hi, shape, dtype=self.torch.int64, device="cuda", generator=self.gen)

    def draw(self, n_out):
        """(u, e1, mask) for n_out output ciphertexts, sampled now on the device (mmh_sample_*)"""
        return self.sampler.reshare(n_out)

    def conv_inputs(self, shape):
        from openmm_amd.hom_conv2d_ss import Meta
        c, h, w, fh, fw, m, same, stride = shape
        shared = c != 3  # first conv runs with kIsSharedInput=false (main_resnet50.cpp:4979)
        meta = Meta((c, h, w), (c, fh, fw), m, same, stride, shared)
        plan = self.srv.plan(meta)
        cts = self.rnd(0, 1 << 62, plan.n_in, 2, 2, self.N) % self.qs.view(1, 1, 2, 1)
        share = self.rnd(0, self.t, c, h, w) if shared else None
        filters = (self.rnd(-(1 << 12) + 1, 1 << 12, m, c, fh, fw) | 1) & (self.t - 1)
        return meta, plan, cts, share, filters

    def conv_call(self, shape):
        meta, plan, cts, share, filters = self.conv_inputs(shape)
        # (the zero-filter verdict of conv2DOneFilter is computed inside the call and read by srv.raise_pending() behind
        # the caller's synchronisation, as the C++ host reads it behind the one it needs before the send: the host does
        # not stop at every layer)
        # randomness: the call draws u, e1 and the mask itself (mmh_conv_server_tensors_sampled), as conv2DSS does
        return lambda: self.srv.conv2DSSFromTensors(cts, share, filters, meta, self.sampler, defer_verdict=True)

    def fc_call(self, rows, cols):
        from openmm_amd.hom_fc_ss import Meta as FcMeta
        fmeta = FcMeta((rows, cols), (cols,), True)
        fplan = self.fc.plan(fmeta)
        vec_ct = self.rnd(0, 1 << 62, fplan.n_in, 2, 2, self.N) % self.qs.view(1, 1, 2, 1)
        vec, wmat = self.rnd(0, self.t, cols), (self.rnd(-(1 << 12) + 1, 1 << 12, rows, cols) | 1) & (self.t - 1)
        return lambda: self.fc.matVecMul(self.fc.encodeWeightMatrix(wmat, fmeta), vec_ct,
                                         self.fc.encodeInputVector(vec, fmeta), fmeta, self.draw(fplan.n_out))

    def bn_call(self, n):
        """ALICE's side of one ElemWiseActModelVectorMult on n elements (see the class comment); the client's
        ciphertexts, Ferret's blocks for the two COTs and the model's scale vector (already getFieldElt'ed) resident"""
        import ctypes as C
        torch, L, lib, p, l, N = self.torch, self.L, self.lib, FIELD_PRIME, RING_BITS, self.N
        n_ct = (n + N - 1) // N
        x = self.rnd(0, self.t, n)
        scale = self.rnd(0, p, n)
        cts = self.rnd(0, 1 << 62, n_ct, 2, 2, N) % self.qs.view(1, 1, 2, 1)
        rcm = [self.rnd(-(1 << 62), 1 << 62, n, 2) for _ in range(2)]
        two = (C.c_uint64 * 2)
        delta, seed = two(0x0123456789ABCDEF, 0xFEDCBA9876543211), two(0x1122334455667788, 0x99AABBCCDDEEFF00)
        wires = []
        for prime in (1, 0):
            nb = C.c_size_t()
            L.check(lib.mmh_cot_wire_bytes(n, l, p, prime, C.byref(nb)))
            wires.append(torch.empty((nb.value + 7) // 8, dtype=torch.int64, device="cuda"))
        gids = C.c_uint64()
        L.check(lib.mmh_cot_gids(n, C.byref(gids)))
        h = self.bn.ctx.handle

        def call():
            s = L._stream_ptr()
            # what needs no ciphertext, two launches: Ring2Field with ALICE as the COT sender (truncation.cpp:518-581;
            # silent_ot.h:129-163) into encodeVector's plaintexts, the scales' plaintexts, addMask's r drawn in place and
            # Field2Ring_with_Truncate on her share -r (truncation.cpp:639-695; silent_ot.h:85-127), side by side; then
            # BatchEncoder's transform of the three plaintext batches
            pts3 = torch.empty(3, n_ct, N, dtype=torch.int64, device="cuda")
            out = torch.empty(n, dtype=torch.int64, device="cuda")
            out_cts = torch.empty(n_ct, 2, N, dtype=torch.int64, device="cuda")
            L.check(lib.mmh_bn_alice_prepare(h, L.dptr(rcm[0]), L.dptr(rcm[1]), delta, seed, seed, self.gid, L.dptr(x), n, l,
                                             K_SCALE, L.dptr(scale), self.sampler.key, self.sampler._sid(), L.dptr(pts3),
                                             None, L.dptr(out), L.dptr(wires[0]), L.dptr(wires[1]), s))
            self.gid += 2 * gids.value
            # HomBNSS::bn on the client's ciphertexts (three launches)
            L.check(lib.mmh_bn_alice_finish(h, L.dptr(cts), L.dptr(pts3), n, L.dptr(out_cts), s))
            return out_cts, out
        return call


def conv_work(plan, shape, prepared=False):
    """Algorithmic work of one server call of a convolution, from the plan (mmh_conv_plan_work), for the leg's roofline:
    byte multiply-accumulates of the direct route (filters x taps x live columns x 8 column digits x 2 weight digits;
    the eighth column digit is identically zero under the 49-bit limb and is not multiplied: 14), limb-NTTs of the
    transform route's minimal schedule, and the bytes that cross the API boundary (client ciphertexts
    and the server's share in, filters in, result ciphertexts and share out; randomness is made on the device)."""
    c, h, w, fh, fw, m, same, stride = shape
    wk = plan.work()
    N, L = wk.poly_degree, wk.limbs
    direct = bool(wk.direct_prepared if prepared else wk.direct_fresh)
    macs = m * wk.taps * (wk.live_columns[0] * 16 + wk.live_columns[1] * 14) if direct else 0
    # transform route: image 2L per input ct, filters L per filter polynomial, result 2L per output ct; either route
    # adds the tail's three one-limb transforms per output ct (u forward, u * pk_j inverse for both components)
    ntts = 3 * wk.n_out + (0 if direct else 2 * L * wk.n_in + L * m * wk.n_cslices + 2 * L * wk.n_out)
    out_elems = plan.out_shape[0] * plan.out_shape[1] * plan.out_shape[2]
    nbytes = 8 * (wk.n_in * 2 * L * N + (c * h * w if c != 3 else 0) + m * c * fh * fw + wk.n_out * 2 * N + out_elems)
    return {"direct": direct, "i8_macs": macs, "limb_ntts": ntts, "bytes": nbytes}


def resnet50_leg(torch, dist, openmm_amd, local, world, rank, reps=20, convs=None, prefix="resnet50", with_fc=True,
                 all_modes=True, prepared=True, order=None, passes=10):
    """Second half of BASELINE.json's metric: server-side HE compute of all 53 ResNet50 conv layers
    + FC 2048->1001 (what CheetahLinear::conv2d / ::fc run between receiving the client's
    ciphertexts and sending the result, cheetah-api.cpp:417-445, :520-548): filter encoding and
    lift, image-share encoding, multiply-accumulate, mod-switch, re-share, truncation, the randomness of every call
    sampled inside it -- plus the network's live batch-norm layers (NetworkLegs.bn_call).  Output
    filters are sharded over the ranks (hom_conv2d_ss.cc:679-690) and gathered on rank 0; each
    layer's time is the max over ranks, barrier to barrier.  FC runs replicated (520 products).
    Two kinds of figure: per shape (pre-warmed, `reps` identical calls, times its count: `*_he_linear_ms`,
    `*_he_bn_ms`) and ONE INFERENCE IN ORDER (`*_inorder_ms`, one GPU): every call of the network enqueued once, in
    network order, each on buffers of its own, drained, `passes` times."""
    from openmm_amd.sharding import ShardedConv
    net = NetworkLegs(torch, openmm_amd, local)
    srv, N = net.srv, net.N

    def wall_ms(fn, prewarm_s=LAYER_PREWARM_S, reps=reps):
        """per-call wall time in steady state: the call repeated untimed for prewarm_s first (the same count on every
        rank: the sharded forms hold a collective), since an idle MI355X needs tens of ms of load to settle (timed()
        above; one layer call: 0.219 ms over 30 calls from idle, 0.199 over 300, 0.195 over 1000)"""
        def drain():   # the stream's end, then the zero-filter verdicts of the calls behind it (inside the timed region)
            torch.cuda.synchronize()
            srv.raise_pending()

        fn()
        fn()   # warm-up: the scratch cache and torch's allocator have their blocks by now
        drain()
        if prewarm_s > 0:
            t0 = time.perf_counter()
            fn()
            drain()
            one = max(time.perf_counter() - t0, 1e-5)
            n = int(max_over_ranks(float(min(2000, max(1, round(prewarm_s / one)))), torch, dist, world))
            for i in range(n):
                fn()
                if i % 8 == 7:
                    drain()
            drain()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        drain()
        if world > 1:
            dist.barrier()
        return max_over_ranks((time.perf_counter() - t0) / reps * 1e3, torch, dist, world)

    per_layer, total, cached_total, d2h_total, repl_total, online_total, offline_total = [], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0
    cold_total = 0.0
    work = {"i8_macs": 0, "limb_ntts": 0, "bytes": 0, "direct_layers": 0, "layers": 0}
    for sh in (convs or RESNET50_CONVS):
        count = sh[8]
        meta, plan, cts, share, filters = net.conv_inputs(sh[:8])
        shared = meta.is_shared_input
        n_out = plan.n_out
        wk = conv_work(plan, sh[:8])
        for k in ("i8_macs", "limb_ntts", "bytes"):
            work[k] += count * wk[k]
        work["direct_layers"] += count * int(wk["direct"])
        work["layers"] += count

        def layer():   # encodeImage + encodeFilters + conv2DSS in one device call that draws its randomness too
            return srv.conv2DSSFromTensors(cts, share, filters, meta, net.sampler, defer_verdict=True)

        if world == 1 and not prepared:
            ms = wall_ms(layer)
        elif world == 1:
            if with_fc:   # (the ResNet50 leg) the same call timed from a chip that has only seen two warm-up calls
                cold_total += count * wall_ms(layer, prewarm_s=0, reps=5)
            ms = wall_ms(layer)
            # static weights prepared once per model (not what the reference does: reported apart)
            prep = srv.prepareFilters(filters, meta)
            cached_total += count * wall_ms(lambda: srv.conv2DSSPrepared(
                cts, None, prep, meta, net.draw(n_out), share_tensor=share))
            # ... and with the input-independent encryptions of zero of the re-sharing made ahead as well (offline
            # phase; fresh ones are needed per inference, their making is timed apart, u drawn inside it)
            zeros = srv.prepareZeros(meta, net.sampler.ternary(n_out))
            online_total += count * wall_ms(lambda: srv.conv2DSSOnline(cts, share, prep, zeros, meta,
                                                                       net.draw(n_out)[1:]))
            offline_total += count * wall_ms(lambda: srv.prepareZeros(meta, net.sampler.ternary(n_out)))
            del prep, zeros
        else:
            # output filters sharded over the ranks, one-call form per rank, assembled on rank 0 by (a) one RCCL
            # gather into a pre-sized device buffer, (b) per-rank async D2H into one shared page-locked host buffer
            times = {}
            for mode in (("gather", "d2h") if all_modes else ("gather",)):
                shd = ShardedConv(srv, meta, mode=mode)
                times[mode] = wall_ms(lambda: shd.run(cts, share, filters, lambda k: net.draw(k)))
                del shd
            ms = times["gather"]
            if all_modes:
                d2h_total += times["d2h"] * count
                # (c) replicated: every rank runs whole layers of its own inference (no sharding, no exchange)
                repl_total += wall_ms(layer) * count
        per_layer.append(round(ms, 3))
        total += ms * count
        del cts, share, filters
    out = {}
    # the live batch-norm layers, per size (replicated on every rank: ciphertext ranges could be sharded like the
    # filters, openmm_amd/sharding.py: sharded_bole; not in this leg)
    bn_sizes = {}
    for kind, arg in (order or []):
        if kind == "bn":
            bn_sizes[arg] = bn_sizes.get(arg, 0) + 1
    bn_total, bn_bytes = 0.0, 0
    for n, count in sorted(bn_sizes.items()):
        bn_total += count * wall_ms(net.bn_call(n), reps=max(5, reps // 2))
        # algorithmic bytes of ALICE's side per element: share in 8, scale in 8, Ferret blocks 2 x 16, the two COT
        # wires ceil(37/8) + 37/8, result out 8; plus the client's ciphertexts in (2 limbs) and hers out (1 limb)
        n_ct = (n + N - 1) // N
        bn_bytes += count * (n * (8 + 8 + 32 + 5 + 37 / 8 + 8) + n_ct * (4 + 2) * N * 8)
    if bn_sizes:
        out[f"{prefix}_he_bn_ms"] = bn_total
        out[f"{prefix}_bn_layers"] = sum(bn_sizes.values())
        out[f"{prefix}_bn_elements"] = sum(n * c for n, c in bn_sizes.items())
    multi = {}
    if world > 1 and all_modes:
        multi = {f"{prefix}_he_linear_d2h_ms": d2h_total,
                 f"{prefix}_replicated_ms_per_inference": repl_total,
                 f"{prefix}_replicated_inferences_per_s": world / (repl_total / 1e3),
                 f"{prefix}_modes": "he_linear_ms: filters sharded, ONE RCCL gather per layer to rank 0 (device); d2h_ms: "
                                    "filters sharded, per-rank async copy into one shared pinned host buffer; replicated: "
                                    "one independent inference per rank (FC excluded)"}
    fc_ms = 0.0
    if with_fc:
        fc_ms = wall_ms(net.fc_call(1001, 2048))
    conv_total = total
    total += fc_ms
    out.update({f"{prefix}_he_linear_ms": total, f"{prefix}_conv_layers": work["layers"],
                f"{prefix}_conv_ms_per_shape": per_layer,
                f"{prefix}_randomness": "u, e1 and the mask of every output ciphertext (and BN's r) are sampled on the "
                                        "device inside every timed call"})
    if with_fc:
        out["resnet50_fc_ms"] = round(fc_ms, 3)
        out["resnet50_sharding"] = f"output filters x{world}, gather on rank 0" if world > 1 else "none"
        out["layer_prewarm_ms"] = LAYER_PREWARM_S * 1e3
    if world == 1:
        if cold_total:
            out[f"{prefix}_he_linear_from_idle_ms"] = cold_total + fc_ms
            out[f"{prefix}_he_linear_from_idle_note"] = (
                "every layer shape timed over 5 calls behind 2 warm-up calls only, as in earlier rounds: the chip has not "
                "settled and the window's start and drain weigh 2-3 % (resnet50_he_linear_ms pre-warms each shape for "
                "layer_prewarm_ms and times 20 calls)")
        if prepared:
            out[f"{prefix}_he_linear_prepared_filters_ms"] = cached_total + fc_ms
            out[f"{prefix}_he_linear_online_ms"] = online_total + fc_ms
            out[f"{prefix}_he_linear_offline_zeros_ms"] = offline_total
            if with_fc:
                out["resnet50_online_note"] = ("online: static weights prepared per model AND the input-independent "
                                               "pk-encryptions of zero of the re-sharing made ahead of the client's "
                                               "ciphertexts (mmh_reshare_zero_prepare, fresh per inference: "
                                               "offline_zeros_ms); the reference does both inside conv2DSS on every call, "
                                               "which resnet50_he_linear_ms times (FC not split)")
    out.update(multi)
    if f"{prefix}_he_linear_d2h_ms" in out:
        out[f"{prefix}_he_linear_d2h_ms"] += fc_ms
    # ---- one inference, every call once, in network order (one GPU: the sharded forms are timed per layer above)
    if order and world == 1:
        calls = []
        for kind, arg in order:
            calls.append(net.conv_call(arg) if kind == "conv" else (net.bn_call(arg) if kind == "bn" else net.fc_call(*arg)))

        def one_pass():
            for f in calls:
                f()
            torch.cuda.synchronize()
            srv.raise_pending()   # the zero-filter verdicts of the pass's convolutions

        one_pass()
        one_pass()
        t0 = time.perf_counter()
        one_pass()
        first = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(passes):
            one_pass()
        inorder = (time.perf_counter() - t0) / passes * 1e3
        only_linear = [f for f, (kind, _) in zip(calls, order) if kind != "bn"]
        t0 = time.perf_counter()
        for _ in range(passes):
            for f in only_linear:
                f()
            torch.cuda.synchronize()
            srv.raise_pending()
        inorder_linear = (time.perf_counter() - t0) / passes * 1e3
        out[f"{prefix}_inorder_ms"] = inorder
        out[f"{prefix}_he_linear_inorder_ms"] = inorder_linear
        out[f"{prefix}_inorder_note"] = (
            f"one inference: all {len(calls)} linear calls of the network ({work['layers']} convolutions"
            f"{', the FC' if with_fc else ''}, {sum(bn_sizes.values())} live batch-norms with their ring<->field "
            f"conversions and COT post-processing) enqueued once each in network order on buffers of their own, "
            f"randomness sampled inside every call, drained after the last; mean of {passes} passes behind 3 "
            f"(the third took {first:.2f} ms); he_linear_inorder_ms: the same without the batch-norms")
        del calls, only_linear
        # ---- roofline of the leg, from the plans: what the calls have to do against what the chip can do in that time
        sec = inorder_linear / 1e3
        out[f"{prefix}_roofline"] = {
            "time_ms": inorder_linear, "time_is": f"{prefix}_he_linear_inorder_ms (convolutions + FC, sampling in)",
            "i8_macs": work["i8_macs"], "mfma_frac": work["i8_macs"] / sec / I8_MAC_PEAK, "mfma_peak": I8_MAC_PEAK,
            "direct_route_layers": work["direct_layers"], "layers": work["layers"],
            "limb_ntts": work["limb_ntts"],
            "ntt_hbm_frac": work["limb_ntts"] * BYTES_PER_NTT[4096] / sec / HBM_PEAK,
            "algorithmic_bytes": work["bytes"], "hbm_frac": work["bytes"] / sec / HBM_PEAK,
            "bn_algorithmic_bytes": bn_bytes,
            "bn_hbm_frac": (bn_bytes / ((inorder - inorder_linear) / 1e3) / HBM_PEAK) if inorder > inorder_linear else None,
            "kernel_share": recorded_kernel_share(prefix),
            "note": "i8_macs: byte multiply-accumulates of the direct-route layers (M x K x live columns x 16 | 14), "
                    "priced at the dense i8 peak; limb_ntts: transforms of the minimal schedule (the tail's three per "
                    "output ciphertext, and for transform-route layers image, filters and result) at 65,536 B each; "
                    "algorithmic_bytes: ciphertexts, share, filters in, ciphertexts and share out.  Each fraction prices "
                    "ALL of the leg's time against ONE resource, so they do not add up to a utilisation; kernel_share is "
                    "the recorded rocprofv3 split of the same leg (profiles/), not measured in this run"}
    return out


def recorded_kernel_share(prefix):
    """per-kernel share of the leg's GPU time from the committed rocprofv3 summary (profiles/r03_kernel_share.json,
    written by tools/prof_resnet.sh from a --kernel-trace --stats run of the same calls); None when absent"""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_kernel_share.json")) as f:
            return json.load(f).get(prefix)
    except (OSError, ValueError):
        return None


def host_to_host_leg():
    """The C++ host mirror end to end (tests/cpp/network_linear.cc): every ResNet50 convolution + the FC through
    Conv2DWrapper / MatMul2D on host arrays, both parties as two threads of one process on this GPU: encryption,
    ciphertexts through an in-process byte pipe (the stand-in for the network), HE compute, decryption.  Warm pass."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "network_linear")
    if not os.path.exists(exe) or any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ):
        return {}   # (under a profiler the child would inherit its preload: skipped, like measured_traffic)
    try:
        r = subprocess.run([exe, "resnet50", "8"], capture_output=True, text=True, timeout=280)
    except (subprocess.TimeoutExpired, OSError) as e:
        return {"resnet50_host_to_host_error": repr(e)[-300:]}
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

    # `value` is the contract's protocol and nothing else: W untimed warm-up steps, then exactly K timed steps, from a
    # chip that has only seen the data being made.  The steady-state figure (the same step repeated untimed for
    # PREWARM_S first, then the same W + K) is reported beside it under `steady_state`, never as `value`.
    wall, launch_s = timed(lambda: ctx.ntt_forward(data), args.steps, args.warmup, torch, dist, world)
    steady_wall, steady_launch_s = timed(lambda: ctx.ntt_forward(data), args.steps, args.warmup, torch, dist, world,
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
        net = resnet50_leg(torch, dist, openmm_amd, local, world, rank, order=resnet50_order())
        sq = resnet50_leg(torch, dist, openmm_amd, local, world, rank, convs=SQUEEZENET_CONVS, prefix="squeezenet",
                          with_fc=False, all_modes=False, order=squeezenet_order())
        dn = resnet50_leg(torch, dist, openmm_amd, local, world, rank, reps=10, convs=densenet121_convs(),
                          prefix="densenet121", with_fc=False, all_modes=False, order=densenet121_order())
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
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "steady_state": {"value": world * ntts_per_step * args.steps / steady_wall,
                             "ms_per_step": steady_wall / args.steps * 1e3, "launch_us": steady_launch_s * 1e6,
                             "prewarm_ms": PREWARM_S * 1e3,
                             "hbm_frac": ntts_per_step * BYTES_PER_NTT[4096] / steady_launch_s / HBM_PEAK,
                             "note": "the same W warm-up and K timed steps behind prewarm_ms of the same step run "
                                     "untimed (an idle MI355X takes tens of ms of load to settle); not the headline"},
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
