#!/usr/bin/env python3
"""Server-side HomConv of one layer shape, timed with events; run under rocprofv3 for the
per-kernel split.  usage: layer_bench.py C H W fh fw M same stride [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openmm_amd
from openmm_amd.hom_conv2d_ss import HomConv2DSS, Meta

if os.environ.get("MMH_LIB"):   # A/B of library builds (tools/exp/build_variant.sh)
    import openmm_amd.lib as _L0
    _L0.LIB_PATH = os.path.abspath(os.environ["MMH_LIB"])
if os.environ.get("NO_VERDICT_WAIT"):   # host enqueue cost alone: never wait for the zero-filter verdict
    import openmm_amd.lib as _L
    _L.DeferredFlag.__bool__ = lambda self: True
c, h, w, fh, fw, m, same, stride = (int(x) for x in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
t, N = 1 << 37, 4096
ctx = openmm_amd.Context(N, (60, 49), t, device=0)
gen = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda lo, hi, *shape: torch.randint(lo, hi, shape, dtype=torch.int64, device="cuda", generator=gen)
qs = torch.tensor(ctx.q, dtype=torch.int64, device="cuda")
srv = HomConv2DSS()
srv.setUp(ctx, pk_ntt=rnd(0, 1 << 62, 2, 2, N) % qs.view(1, 2, 1))
meta = Meta((c, h, w), (c, fh, fw), m, same, stride, True)
plan = srv.plan(meta)
cts = rnd(0, 1 << 62, plan.n_in, 2, 2, N) % qs.view(1, 1, 2, 1)
share = rnd(0, t, c, h, w)
filters = (rnd(-(1 << 12) + 1, 1 << 12, m, c, fh, fw) | 1) & (t - 1)
rand = (rnd(-1, 2, plan.n_out, N), rnd(-8, 9, plan.n_out, N), rnd(0, 1 << 62, plan.n_out, N) % ctx.q[0])
if os.environ.get("TWO_CALL"):
    layer = lambda: srv.conv2DSS(cts, srv.encodeImage(share, meta), srv.encodeFilters(filters, meta), meta, rand)
else:
    layer = lambda: srv.conv2DSSFromTensors(cts, share, filters, meta, rand)
layer()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
h0 = time.perf_counter()
e0.record()
for _ in range(reps):
    layer()
e1.record()
host_ms = (time.perf_counter() - h0) / reps * 1e3   # enqueue time per layer (queue permitting)
torch.cuda.synchronize()
print(f"n_in={plan.n_in} n_cslices={plan.n_cslices} n_spatial={plan.n_spatial} n_out={plan.n_out} "
      f"filter_polys={m * plan.n_cslices}: {e0.elapsed_time(e1) / reps:.3f} ms/layer, host enqueue {host_ms:.3f} ms/layer")
