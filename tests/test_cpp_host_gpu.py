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


@pytest.mark.gpu
@pytest.mark.parametrize("net,count", [("resnet50", 53), ("densenet121", 121)])
def test_network_linear_layers_through_the_layer_adapters(net, count):
    """BASELINE.json configs[4] in loop-back: every convolution of ResNet50 (53, plus the FC 2048 -> 1001 through
    MatMul2D) and DenseNet121 (121) through Conv2DWrapper on NHWC shares, both parties in-process, sampled outputs
    within 1 of the cleartext convolution (library_fixed_uniform_cheetah.cpp:199,:404)"""
    exe = os.path.join(ROOT, "tests", "cpp", "network_linear")
    assert os.path.exists(exe), "build it with __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([exe, net, "96"], capture_output=True, text=True, timeout=280)
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout and f"{count} convolutions" in r.stdout
    assert "FAIL " not in r.stdout
    # the live batch-norms (ElemWiseActModelVectorMult = Ring2Field, bn, Field2Ring_with_Truncate) in network order
    assert f"{18 if net == 'resnet50' else 63} live batch-norm layers" in r.stdout and "PASS bn " in r.stdout
    if net == "resnet50":
        assert "PASS fc   2048 -> 1001" in r.stdout


@pytest.mark.gpu
def test_server_side_over_several_devices_equals_the_one_device_run():
    """CheetahLinear::Options::devices = {0,0} and {0,0,0} (workers sharing the one GPU of the box, each with its own
    stream, contexts, plans and pool): conv2d / conv2d_nhwc cut by output filters, fc / fc_in_out by row blocks, bn by
    ciphertexts.  With fixed sampler keys ALICE's bytes on the wire and both parties' shares equal the one-device
    run's, layer by layer (tests/cpp/team_test.cc)"""
    exe = os.path.join(ROOT, "tests", "cpp", "team_test")
    assert os.path.exists(exe), "build it with __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=280)
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and "team_test: all passed" in r.stdout
    assert "FAIL " not in r.stdout and "bytes on the wire equal the one-device run's (3 devices)" in r.stdout


@pytest.mark.gpu
def test_loopback_acceptance_with_two_workers():
    """the acceptance program again with MM_GEMINI_DEVICES=0,0: every layer it reconstructs is computed by two workers"""
    assert os.path.exists(BIN), "build it with __graft_entry__.build() (make -C tests/cpp)"
    r = subprocess.run([BIN, "14", "0"], capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, MM_GEMINI_DEVICES="0,0"))
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout
    assert "FAIL" not in r.stdout.replace("FAILED", "")


@pytest.mark.gpu
def test_resnet50_layers_with_two_workers():
    """network_linear resnet50 with MM_GEMINI_DEVICES=0,0: all 53 convolutions, the FC and the 18 batch-norms through the
    cut server side"""
    exe = os.path.join(ROOT, "tests", "cpp", "network_linear")
    r = subprocess.run([exe, "resnet50", "96"], capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, MM_GEMINI_DEVICES="0,0"))
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout and "53 convolutions" in r.stdout
    assert "FAIL " not in r.stdout and "PASS bn " in r.stdout and "PASS fc   2048 -> 1001" in r.stdout
    assert "server side on 2 device worker(s)" in r.stdout


def test_host_library_links_without_gpu():
    """CPU: libmm_gemini.so loads and resolves against libmm_hip.so"""
    import ctypes
    ctypes.CDLL(os.path.join(ROOT, "openmm_amd", "libmm_hip.so"), mode=ctypes.RTLD_GLOBAL)
    ctypes.CDLL(os.path.join(ROOT, "openmm_amd", "libmm_gemini.so"))
