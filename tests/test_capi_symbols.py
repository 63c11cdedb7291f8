"""The C-ABI library builds for gfx950, loads without a GPU, exports every
symbol that include/mpm_hip.h declares, and refuses to run without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mpm_hip.h")).read()
    return sorted(set(re.findall(r"MPM_API\s+[\w\s\*]+?\b(mpm_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from drake_amd import capi
    assert _declared() == sorted(capi.SYMBOLS)


def test_library_builds_and_exports_all_symbols():
    from drake_amd import capi
    lib = capi.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    # the shipped code object targets gfx950 only
    blob = open(capi.library_path(), "rb").read()
    assert b"gfx950" in blob
    assert b"gfx90a" not in blob and b"sm_" not in blob


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from drake_amd import GpuMpm, MpmError
    with pytest.raises(MpmError) as e:
        GpuMpm(6)
    assert e.value.code == -5


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "drake_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cc")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), os.path.join(root, f)
