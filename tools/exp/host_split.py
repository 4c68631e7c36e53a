#!/usr/bin/env python3
"""Where the host time of one conv2DSSFromTensors call goes: the wrapper's statements timed one by one
(no device sync between them), then the bare C call alone.  usage: host_split.py C H W fh fw M same stride [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd
import openmm_amd.lib as L
from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta, torch_empty_flag

c, h, w, fh, fw, m, same, stride = (int(x) for x in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 300
t, N = 1 << 37, 4096
ctx = openmm_amd.Context(N, (60, 49), t, device=0)
gen = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)
qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda")
srv = HomConv2DSS()
srv.setUp(ctx, pk_ntt=rnd(0, 1 << 62, 2, 2, N) % qs.view(1, 2, 1))
meta = Meta((c, h, w), (c, fh, fw), m, same, stride, True)
p = srv.plan(meta)
cts = rnd(0, 1 << 62, p.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
share = rnd(0, t, c, h, w)
filters = (rnd(-(1 << 12) + 1, 1 << 12, m, c, fh, fw) | 1) & (t - 1)
rand = (rnd(-1, 2, p.n_out, N), rnd(-8, 9, p.n_out, N), rnd(0, 1 << 62, p.n_out, N) % ctx.q[0])
u, e1, mask = rand
for _ in range(3):
    srv.conv2DSSFromTensors(cts, share, filters, meta, rand)
torch.cuda.synchronize()

names = ["flag+check_filters", "DeferredFlag", "check_rand+empty", "server_tensors", "bool(nz)"]
acc = [0.0] * len(names)
tot0 = time.perf_counter()
for _ in range(reps):
    t0 = time.perf_counter()
    bad = torch_empty_flag(filters)
    L.check(ctx.lib.mmh_conv_check_filters(p.handle, L.dptr(filters), L.dptr(bad), L._stream_ptr()))
    t1 = time.perf_counter()
    nz = L.DeferredFlag(bad, negate=True)
    t2 = time.perf_counter()
    srv._check_rand(p, rand, ("u", "e1", "mask"))
    out_cts = ctx._empty(p.n_out, 2, 1, N)
    out_share = ctx._empty(*p.out_shape)
    t3 = time.perf_counter()
    L.check(ctx.lib.mmh_conv_server_tensors(p.handle, L.dptr(cts), L.dptr(share), L.dptr(filters), L.dptr(srv.pk_last),
                                            L.dptr(u), L.dptr(e1), L.dptr(mask), L.dptr(out_cts), L.dptr(out_share),
                                            L.dptr(None), L._stream_ptr()))
    t4 = time.perf_counter()
    ok = bool(nz)
    t5 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[i] += d
torch.cuda.synchronize()
tot = (time.perf_counter() - tot0) / reps * 1e6
print(f"shape {sys.argv[1:9]} direct={p.direct_fresh if hasattr(p, 'direct_fresh') else '?'} total {tot:.1f} us/layer")
for n, a in zip(names, acc):
    print(f"  {n:24s} {a / reps * 1e6:7.1f} us")

# inline verdict: one C call, verdict at the end
bad = torch_empty_flag(filters)
args = (p.handle, L.dptr(cts), L.dptr(share), L.dptr(filters), L.dptr(srv.pk_last), L.dptr(u), L.dptr(e1), L.dptr(mask),
        L.dptr(out_cts), L.dptr(out_share), L.dptr(bad), L._stream_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.lib.mmh_conv_server_tensors(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"  bare C call, inline verdict: host {(t1 - t0) / reps * 1e6:.1f} us/layer, with drain {(t2 - t0) / reps * 1e6:.1f} us/layer")
