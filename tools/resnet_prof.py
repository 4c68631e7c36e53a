#!/usr/bin/env python3
"""One network's linear calls in network order (bench.py's in-order leg: convolutions, FC, live batch-norms, randomness
sampled inside every call), a few passes, for rocprofv3 --kernel-trace --stats: the per-kernel share of the leg.
usage: resnet_prof.py [resnet50|densenet121|squeezenet] [passes=5]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openmm_amd
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
order = {"resnet50": bench.resnet50_order, "densenet121": bench.densenet121_order, "squeezenet": bench.squeezenet_order}[which]()
net = bench.NetworkLegs(torch, openmm_amd, 0)
calls = [net.conv_call(a) if k == "conv" else (net.bn_call(a) if k == "bn" else net.fc_call(*a)) for k, a in order]
import time
for p in range(passes + 1):
    t0 = time.perf_counter()
    for f in calls:
        f()
    torch.cuda.synchronize()
    net.srv.raise_pending()
    print(f"pass {p}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
