#!/usr/bin/env python3
"""The live batch-norm calls of a network (bench.NetworkLegs.bn_call: ring->field, COT post-processing, bn with its
mask, field->ring with truncation, ALICE's side) enqueued once each in network order, a few passes, events around each
pass: the BN leg alone.  usage: bn_time.py [resnet50|densenet121] [passes=10]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
import openmm_amd
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
order = {"resnet50": bench.resnet50_order, "densenet121": bench.densenet121_order}[which]()
net = bench.NetworkLegs(torch, openmm_amd, 0)
calls = [net.bn_call(a) for k, a in order if k == "bn"]
ts, hs = [], []
for p in range(passes + 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    w0 = time.perf_counter()
    for f in calls:
        f()
    w1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    if p >= 3:
        ts.append(e0.elapsed_time(e1))
        hs.append((w1 - w0) * 1e3)
print(f"{which}: {len(calls)} batch-norm calls, {sum(ts) / len(ts):.3f} ms per inference (min {min(ts):.3f}), "
      f"{sum(ts) / len(ts) / len(calls) * 1e3:.1f} us per call; the host enqueues a pass in {sum(hs) / len(hs):.3f} ms")
