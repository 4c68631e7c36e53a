#!/usr/bin/env python3
"""The DenseNet121 leg of bench.py alone, for rocprofv3 --kernel-trace (tools/exp/layer_kernels.py reads the result)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import openmm_amd
import bench
out = bench.resnet50_leg(torch, None, openmm_amd, 0, 1, 0, reps=3, convs=bench.densenet121_convs(), prefix="densenet121",
                         with_fc=False, all_modes=False, prepared=False)
print({k: v for k, v in out.items() if "per_shape" not in k})
print(out["densenet121_conv_ms_per_shape"])
