#!/usr/bin/env python3
"""The ResNet50 leg of bench.py alone (fresh filters, one GPU), for rocprofv3 --kernel-trace --stats."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openmm_amd
import bench
bench.LAYER_PREWARM_S = 0.0   # the same number of calls for every layer shape: the kernel shares are per pass over the shapes
out = bench.resnet50_leg(torch, None, openmm_amd, 0, 1, 0, reps=5, prepared=False)
print({k: v for k, v in out.items() if k != "resnet50_conv_ms_per_shape"})
print(out["resnet50_conv_ms_per_shape"])
