"""Times the COT post-processing kernels alone (for rocprofv3): python tools/cot_bench.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import openmm_amd  # noqa: E402,F401
from bench import conversion_leg  # noqa: E402

if __name__ == "__main__":
    res = conversion_leg(torch, openmm_amd)
    for k, v in res.items():
        if "cot" in k:
            print(k, f"{v:.4g}")
