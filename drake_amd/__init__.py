"""drake_amd: MI355X-native cloth-MPM substep engine behind g1n0st/drake's
GpuMpmState / GpuMpmSolver interface.

The product is the C-ABI library built from drake_amd/csrc (see
include/mpm_hip.h).  This package only holds the build recipe, a ctypes binding
used by the tests and bench.py, and synthetic scene generators.  There is no
CPU fallback: every compute call goes to the HIP library or raises.
"""
from .capi import (MpmError, GpuMpm, load_library, library_path, Material, Collider, GridCollider, BC_TABLE,  # noqa: F401
                   grid_collider_preset, ARR, PHASES)
from . import scenes  # noqa: F401
