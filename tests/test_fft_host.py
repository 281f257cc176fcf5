"""The register DFTs of speech_to_text_amd/csrc/fft_core.h (the same header the HIP kernels compile) and the 20 x 20 /
16 x 32 lane choreography built from them, compiled with g++ and checked against a naive O(N^2) float64 DFT on the
CPU (tests/host/fft_host_check.cpp).  No GPU needed."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_fft_core_against_naive_dft(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "fft_host_check")
    src = os.path.join(ROOT, "tests", "host", "fft_host_check.cpp")
    subprocess.run([gxx, "-O2", "-std=c++17", src, "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout, r.stdout
