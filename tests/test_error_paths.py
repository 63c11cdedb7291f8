"""The error contract of the C ABI (include/mpm_hip.h, "Conventions"): no call throws, no call terminates the
process.  The reference's contract is settings.h:11-25 (CUDA_SAFE_CALL throws only in DEBUG builds; a release build
never takes the caller down); here every entry point ends in a catch-all, and numbers that come out of BUFFERS
(migration headers written by another rank) are checked before anything is sized from them.

CPU part: the exception barrier and the header checks are host code and run without a GPU.  GPU part: a device
allocation that fails in the middle of a slot-space re-allocation leaves the engine usable (two-phase dist_resize).
VERDICT r4, item 3."""
import ctypes as C

import numpy as np
import pytest

from drake_amd import capi

ERR_INVALID, ERR_CAPACITY, ERR_NOMEM, ERR_INTERNAL = -1, -4, -9, -10


def _msg(lib):
    return (lib.mpm_last_error() or b"").decode()


def test_a_cxx_exception_inside_an_entry_point_becomes_a_status_code():
    lib = capi.load_library()
    assert lib.mpm_debug_throw(0) == ERR_NOMEM          # std::bad_alloc
    assert "bad_alloc" in _msg(lib)
    assert lib.mpm_debug_throw(1) == ERR_INTERNAL       # std::length_error from a container asked for an absurd size
    assert "exception" in _msg(lib)
    assert lib.mpm_debug_throw(2) == ERR_INTERNAL       # not derived from std::exception
    assert lib.mpm_debug_throw(3) == ERR_INVALID
    # ... and the process is still here, the library still works
    m = capi.Material()
    assert lib.mpm_default_material(C.byref(m)) == 0 and m.density == 2000.0


def test_every_entry_point_of_the_library_has_the_barrier():
    """The source is the evidence: every `int mpm_*(...)` definition in the one translation unit is a function-try-block
    that ends in the catch-all (size_t / const char* entry points compute a size or return a pointer: nothing to throw)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(capi.__file__), "csrc", "mpm_engine.hip")).read()
    heads = re.findall(r"^int (mpm_\w+)\(", src, flags=re.M)
    assert len(heads) >= 70
    for name in heads:
        at = src.index(f"\nint {name}(")
        body_start = src.index("{", at)
        assert src[at:body_start].rstrip().endswith("try"), name
    assert src.count("} MPM_CATCH_ALL") == len(heads)


def _plan(hl, hr, cap=1000, scene=(5000, 3000), held=(2000, 1000), slots=(3000, 1500), headroom=1.5):
    lib = capi.load_library()
    out = (C.c_size_t * 6)()
    a = None if hl is None else np.asarray(hl, np.uint32)
    b = None if hr is None else np.asarray(hr, np.uint32)
    rc = lib.mpm_dist_plan_migration(None if a is None else a.ctypes.data_as(C.c_void_p),
                                     None if b is None else b.ctypes.data_as(C.c_void_p), cap, scene[0], scene[1], held[0],
                                     held[1], slots[0], slots[1], C.c_float(headroom), out)
    return rc, list(out), _msg(lib)


def test_migration_headers_are_checked_before_anything_is_sized():
    # a sound pair of headers: counts add up, nothing to re-allocate
    rc, out, _ = _plan([10, 4, 0, 0], [6, 6, 0, 0])
    assert rc == 0 and out == [10, 6, 2010, 1006, 3000, 1500]
    # no neighbour on one side
    rc, out, _ = _plan(None, [6, 1, 0, 0])
    assert rc == 0 and out[:2] == [1, 5]
    # more arrive than the slot space holds: grow to headroom x the need, never beyond the scene
    rc, out, _ = _plan([900, 900, 0, 0], [900, 900, 0, 0], held=(2900, 1000))
    assert rc == 0 and out[2] == 4700 and out[4] == 5000 and out[5] == 1756   # min(5000, 1.5 * 4700 + 256); both kinds re-sized: 1.5 * 1000 + 256
    rc, out, _ = _plan([1000, 1000, 0, 0], [1000, 1000, 0, 0], held=(3000, 1000), slots=(3000, 1500), scene=(5000, 3000))
    assert rc == 0 and out[2] == 5000 and out[4] == 5000                      # need clamped to the scene
    # corrupt headers: an absurd count (what a stale or foreign buffer holds), more faces than records, padding not zero
    rc, _, msg = _plan([0xFFFFFFFF, 0, 0, 0], None)
    assert rc == ERR_CAPACITY and "packed" in msg
    rc, _, msg = _plan([1001, 0, 0, 0], None)
    assert rc == ERR_CAPACITY
    rc, _, msg = _plan([5, 6, 0, 0], None)
    assert rc == ERR_INVALID and "corrupt" in msg
    rc, _, msg = _plan([5, 1, 0, 7], None)
    assert rc == ERR_INVALID and "corrupt" in msg
    rc, _, msg = _plan(None, [5, 1, 3, 0])
    assert rc == ERR_INVALID and "right" in msg
    # the rank's own counts are checked too (a control block that says it holds more than it has slots for)
    rc, _, msg = _plan([1, 0, 0, 0], None, held=(3001, 10))
    assert rc == ERR_INTERNAL and "inconsistent" in msg
    rc, _, msg = _plan([1, 0, 0, 0], None, cap=0)
    assert rc == ERR_INVALID


@pytest.mark.gpu
def test_a_failed_allocation_inside_a_slot_space_resize_leaves_the_rank_usable():
    """Two-phase dist_resize on the scene of test_world_gpu's cloth that slides across a cut: at every migration every
    device allocation of the re-allocation is made to fail in turn.  Each failed call returns MPM_ERR_NOMEM and leaves the
    slot space and the error flags as they were; the same call repeated without the injected failure goes through, and
    the run ends equal to a single engine's."""
    import torch
    from drake_amd import ARR, GpuMpm, scenes
    from drake_amd.dist import LocalWorld
    from tests.helpers import close
    bits, steps, dt = 6, 72, 1e-3
    sheets = scenes.cloth_stack(2, 40, bits, z0=0.5, side=0.25, seed=5, vel_amp=0.2, center=(0.36, 0.5))
    for pos, vel, idx in sheets:
        vel[:, 0] += 3.0

    def engine():
        g = GpuMpm(bits)
        scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
        return g

    ref = engine()
    ref.run_substeps(steps, dt, -1)
    ref.gpu_sync()
    rp, rv = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES)
    n = ref.n_particles
    ref.destroy()

    engines = [engine(), engine()]
    w = LocalWorld(engines, [0, 8, 16], 2, 0, 0, capacity_blocks=512, migrate_every=0, migrate_capacity=8192,
                   device=torch.device("cuda", 0))
    failures = []
    for g in engines:
        orig = g.dist_migrate_apply

        def flaky(rl, rr, cap, g=g, orig=orig):
            k = 1
            while True:
                before = g.stats()
                g.debug_fail_alloc(k)
                try:
                    orig(rl, rr, cap)
                    g.debug_fail_alloc(0)
                    return
                except capi.MpmError as exc:
                    assert exc.code == ERR_NOMEM, exc
                    assert "stays usable" in str(exc)
                    failures.append(k)
                    after = g.stats()
                    assert (after["face_slots"], after["vertex_slots"]) == (before["face_slots"], before["vertex_slots"])
                    assert after["error_flags"] == 0
                    k += 1
                    assert k < 200
        g.dist_migrate_apply = flaky
    w.run_substeps(steps, dt, -1)
    w.sync()
    assert len(failures) >= 30 and max(failures) >= 30, failures   # (a re-allocation is 34 arrays, each failed once)
    assert engines[1].dist_geometry()["slot_resizes"] >= 2
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    owners = np.zeros(n, np.int32)
    for g in engines:
        assert g.stats()["error_flags"] == 0
        own = g.dist_roles() == 1
        owners += own
        pos[own], vel[own] = g.download(ARR.POSITIONS)[own], g.download(ARR.VELOCITIES)[own]
    assert np.all(owners == 1)
    close(pos, rp, scale=1.0, rtol=1e-5, what="resize with failed allocations: positions vs single engine")
    close(vel, rv, scale=max(float(np.abs(rv).max()), 1.0), rtol=1e-4, what="resize with failed allocations: velocities vs single engine")
    for g in engines:
        g.destroy()
