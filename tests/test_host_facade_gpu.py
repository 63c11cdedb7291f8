"""The C++ facade (include/gpu_mpm.hpp) and the DeformableDriver-style loop
(drake_amd/host/mpm_driver.hpp), exercised by the mirror of the reference's cuda_mpm_test.cc."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cloth_test_binary():
    exe = os.path.join(ROOT, "drake_amd", "host", "cloth_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    out = subprocess.run([exe, "60", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cloth_test ok" in out.stdout
