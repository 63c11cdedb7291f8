"""Builds the HIP engine in-tree: drake_amd/libmpm_hip.so (gfx950 only)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpm_hip.so")
SOURCES = ["mpm_engine.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-fno-slp-vectorize",
         "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    return False


def build(force: bool = False, verbose: bool = False, extra=(), out: str | None = None) -> str:
    """`out`: write the library somewhere else (A/B variants of an experiment, loaded through
    MPM_HIP_LIBRARY; the product is always drake_amd/libmpm_hip.so)."""
    if os.environ.get("MPM_DIAG_BUILD") == "1":
        extra = tuple(extra) + ("-DMPM_DIAG=1",)
        force = True
    if out is None and os.environ.get("MPM_HIP_LIBRARY"):
        return os.environ["MPM_HIP_LIBRARY"]   # an experiment's prebuilt variant (scratch/ab_build.py)
    if out is None and not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")
    cmd = [hipcc, *FLAGS, *extra, "-o", out or LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv or "--diag" in sys.argv, verbose=True,
          extra=("-DMPM_DIAG=1",) if "--diag" in sys.argv else ())
    print(LIB)
