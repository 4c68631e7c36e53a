#!/usr/bin/env python3
"""A/B of one environment switch of libmm_hip over a network's convolution shapes, in ONE process and interleaved per
shape (the library reads its switches with getenv at every call, and boxes of the pool differ by several per cent, so
two processes on two boxes do not compare).  Per shape: the server call with randomness drawn inside, pre-warmed, then
`reps` calls timed with events, for VAR=a and VAR=b alternately, twice; totals weighted by the shape's count.
usage: shapes_ab.py VAR a b [resnet50|densenet121|squeezenet] [reps=20]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd
import bench

var, a, b = sys.argv[1:4]
which = sys.argv[4] if len(sys.argv) > 4 else "resnet50"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
shapes = {"resnet50": bench.RESNET50_CONVS, "densenet121": bench.densenet121_convs(), "squeezenet": bench.SQUEEZENET_CONVS}[which]
net = bench.NetworkLegs(torch, openmm_amd, 0)


def timed(call, value):
    if value == "unset":
        os.environ.pop(var, None)
    else:
        os.environ[var] = value
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    import time
    t_end = time.perf_counter() + 0.03
    while time.perf_counter() < t_end:   # settle the clocks
        call()
        n += 1
        if n % 64 == 0:
            torch.cuda.synchronize()
            net.srv.raise_pending()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    return e0.elapsed_time(e1) / reps * 1e3


tot = {a: 0.0, b: 0.0}
for sh in shapes:
    call = net.conv_call(sh[:8])
    r = {a: [], b: []}
    for _ in range(2):
        for v in (a, b):
            r[v].append(timed(call, v))
    ta, tb = min(r[a]), min(r[b])
    tot[a] += ta * sh[8]
    tot[b] += tb * sh[8]
    print(f"{str(sh[:8]):44s} x{sh[8]}: {var}={a} {ta:7.1f} us   {var}={b} {tb:7.1f} us   {tb / ta - 1:+.1%}", flush=True)
print(f"total: {var}={a} {tot[a] / 1e3:.3f} ms   {var}={b} {tot[b] / 1e3:.3f} ms   {tot[b] / tot[a] - 1:+.1%}")
