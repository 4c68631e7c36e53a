"""Runs the C++ host mirror's two-party loop-back acceptance program (tests/cpp/loopback_test.cc)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "loopback_test")


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["14", "0"], ["20", "1"]])
def test_cpp_loopback_acceptance(args):
    assert os.path.exists(BIN), "build it with __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=280)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout
    assert "FAIL" not in r.stdout.replace("FAILED", "")


@pytest.mark.gpu
def test_squeezenet_linear_layers_through_the_layer_adapters():
    """all 26 SqueezeNet convolutions through Conv2DWrapper (NHWC arrays, both parties in-process),
    sampled outputs within 1 of the cleartext convolution"""
    exe = os.path.join(ROOT, "tests", "cpp", "network_linear")
    assert os.path.exists(exe), "build it with __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([exe, "sqnet", "128"], capture_output=True, text=True, timeout=280)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout and "26 convolutions" in r.stdout


def test_host_library_links_without_gpu():
    """CPU: libmm_gemini.so loads and resolves against libmm_hip.so"""
    import ctypes
    ctypes.CDLL(os.path.join(ROOT, "openmm_amd", "libmm_hip.so"), mode=ctypes.RTLD_GLOBAL)
    ctypes.CDLL(os.path.join(ROOT, "openmm_amd", "libmm_gemini.so"))
