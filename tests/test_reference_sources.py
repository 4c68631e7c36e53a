"""SURVEY.md section 8(b), upper face: the reference's own protocol sources compile UNCHANGED against the seal:: facade
of this repository (include/seal/seal.h + the forwarding headers beside it, include/eigen_stub for the one Eigen
container tensor.h uses) and run on libmm_hip.so.

CPU part: compile-only, straight from /root/reference with -I, nothing copied (skipped where the tree is absent, as on
the GPU box).  GPU part: the two programs tests/cpp/Makefile built from those sources into tests/cpp/_ref/ (git-ignored,
they travel with the snapshot like oracle/_ref): ref_gemini_test drives HomConv2DSS / HomFCSS / HomBNSS directly and
compares the reference encoders' plaintexts word for word with the device encoders; ref_cheetah_api_test runs the
reference's CheetahLinear driver, two parties over sci::NetIO on 127.0.0.1.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GEMINI = ["cheetah/shape_inference.cc", "cheetah/tensor_shape.cc", "cheetah/tensor_encoder.cc",
          "cheetah/hom_conv2d_ss.cc", "cheetah/hom_fc_ss.cc", "cheetah/hom_bn_ss.cc",
          "core/types.cc", "core/logging.cc", "core/util/math.cc"]
FLAGS = ["-std=c++17", "-fsyntax-only", "-DUSE_APPROX_RESHARE=1", f"-I{REF}/include", f"-I{ROOT}/include",
         f"-I{ROOT}/include/eigen_stub"]

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "gemini")) or
                                     shutil.which("g++") is None, reason="reference tree (or g++) not present")


@needs_reference
@pytest.mark.parametrize("src", GEMINI)
def test_reference_gemini_source_compiles_unchanged(src):
    r = subprocess.run(["g++"] + FLAGS + [os.path.join(REF, "include", "gemini", src)],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-4000:]


@needs_reference
def test_reference_cheetah_api_compiles_unchanged():
    """SCI/src/cheetah/cheetah-api.cpp: key exchange through Serializable<PublicKey>::save / PublicKey::load, the
    ciphertext wire through Ciphertext::save / load, every CheetahLinear method"""
    r = subprocess.run(["g++"] + FLAGS + ["-maes", "-msse4.1", f"-I{REF}/SCI/src",
                                          os.path.join(REF, "SCI", "src", "cheetah", "cheetah-api.cpp")],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-4000:]


def test_facade_headers_hold_no_reference_text():
    """the forwarding headers only include seal.h; seal.h declares what the reference calls and nothing from SEAL's
    own text (SEAL is not in /root/reference at all: the facade is written against the call sites)"""
    inc = os.path.join(ROOT, "include", "seal")
    for dirpath, _, files in os.walk(inc):
        for f in files:
            if f in ("seal.h", "wire.h", "zstd_decode.h"):   # the facade, its serialisation, its decoder (own text)
                continue
            body = open(os.path.join(dirpath, f)).read()
            lines = [ln for ln in body.splitlines() if ln.strip() and not ln.strip().startswith("//")]
            assert all(ln.startswith("#") for ln in lines), (f, lines[:3])


@pytest.mark.gpu
@pytest.mark.parametrize("exe,needle", [
    ("ref_gemini_test", "reference TensorEncoder plaintexts == device encoders"),
    ("ref_cheetah_api_test", "CheetahLinear::BOLE reconstructs"),
])
def test_reference_sources_run_on_the_device_library(exe, needle):
    path = os.path.join(ROOT, "tests", "cpp", "_ref", exe)
    assert os.path.exists(path), "built by make -C tests/cpp where /root/reference exists; travels with the snapshot"
    r = subprocess.run([path], capture_output=True, text=True, timeout=280)
    print(r.stdout[-6000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout
    assert "FAIL " not in r.stdout and needle in r.stdout
