#!/usr/bin/env python3
"""Host time of one server call through the Python wrapper (sampling + conv2DSSFromTensors with the verdict deferred),
no synchronisation inside the loop, against the GPU time of the same calls; then a cProfile of the loop.
usage: host_call_cost.py C H W fh fw M same stride"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd
import bench
shape = tuple(int(x) for x in sys.argv[1:9])
net = bench.NetworkLegs(torch, openmm_amd, 0)
call = net.conv_call(shape)
for _ in range(20):
    call()
torch.cuda.synchronize()
net.srv.raise_pending()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
net.srv.raise_pending()
print(f"host enqueue {(t1 - t0) / n * 1e6:.1f} us per call, until drained {(t2 - t0) / n * 1e6:.1f} us per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    call()
pr.disable()
torch.cuda.synchronize()
net.srv.raise_pending()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
