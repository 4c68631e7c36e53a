#!/usr/bin/env python3
"""Which route wins per convolution shape: every shape of a network timed with MMH_CONV_POINTWISE=0 (transform route),
=1 (direct route) and unset (the plan's cost model), each on a server object of its own (plans are cached per object and
read the switch when they are made).  usage: shapes_route.py [resnet50|densenet121|squeezenet] [reps=20]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "densenet121"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = {"resnet50": bench.RESNET50_CONVS, "densenet121": bench.densenet121_convs(), "squeezenet": bench.SQUEEZENET_CONVS}[which]
nets = {}
for v in ("0", "1", None):
    if v is None:
        os.environ.pop("MMH_CONV_POINTWISE", None)
    else:
        os.environ["MMH_CONV_POINTWISE"] = v
    nets[v] = bench.NetworkLegs(torch, openmm_amd, 0)


def timed(call, net):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    t_end = time.perf_counter() + 0.02
    n = 0
    while time.perf_counter() < t_end:
        call()
        n += 1
        if n % 64 == 0:
            torch.cuda.synchronize()
            net.srv.raise_pending()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {"0": 0.0, "1": 0.0, None: 0.0, "best": 0.0}
for sh in shapes:
    r = {}
    for v in ("0", "1", None):
        if v is None:
            os.environ.pop("MMH_CONV_POINTWISE", None)
        else:
            os.environ["MMH_CONV_POINTWISE"] = v
        try:
            r[v] = timed(nets[v].conv_call(sh[:8]), nets[v])
        except Exception as e:   # a route a plan cannot take (more than 5120 taps)
            r[v] = float("nan")
    best = min(x for x in (r["0"], r["1"]) if x == x)
    for k in ("0", "1", None):
        tot[k] += (r[k] if r[k] == r[k] else best) * sh[8]
    tot["best"] += best * sh[8]
    flag = "" if r[None] <= best * 1.05 else "   <-- model picks the slower route"
    print(f"{str(sh[:8]):44s} x{sh[8]:2d}: transform {r['0']:7.1f}  direct {r['1']:7.1f}  model {r[None]:7.1f} us{flag}", flush=True)
print(f"total: transform {tot['0'] / 1e3:.3f}  direct {tot['1'] / 1e3:.3f}  model {tot[None] / 1e3:.3f}  best-of-two {tot['best'] / 1e3:.3f} ms")
