"""Capacity paths of the engine: the slab pool that grows at synchronisation points, the per-block
slab list (GRID_LIST) overflow of a dense, finely split pile, and the packed item-count field."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 1e-3


def _with_env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = str(value)
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


def _spread_scene(bits=6):
    from drake_amd import scenes
    sheets = scenes.cloth_stack(3, 48, bits, z0=0.5, side=0.6, seed=3, vel_amp=0.5)
    for pos, vel, idx in sheets:
        vel[:, 0] += 0.8
    return sheets


def test_slab_pool_grows_at_sync_and_changes_nothing():
    """The slabs are allocated for the blocks in use, not for one particle per block: a pool that starts
    barely large enough is doubled at mpm_sync / mpm_get_stats, and the trajectory is the same."""
    from drake_amd import ARR as A, GpuMpm, scenes

    def run(n=30):
        g = GpuMpm(6)
        g.set_deterministic(True)
        scenes.populate(g, _spread_scene())
        st0 = g.stats()
        for _ in range(n // 10):
            g.run_substeps(10, DT, -1)
            g.gpu_sync()
        return g, st0

    ref, st0 = run()
    home = st0["home_blocks"]
    assert home > 40
    small, _ = _with_env("MPM_SLAB_CAPACITY", int(home * 1.2), run)      # < 2 x items: grown at the first sync
    assert small.stats()["error_flags"] == 0
    assert np.array_equal(small.download(A.POSITIONS), ref.download(A.POSITIONS))
    assert np.array_equal(small.download(A.VELOCITIES), ref.download(A.VELOCITIES))
    # a pool that cannot even hold the first re-sort's items is grown by mpm_finalize itself (the re-sort reports
    # the count it wanted, the host doubles the pool and repeats it before any substep has run)
    tiny, _ = _with_env("MPM_SLAB_CAPACITY", max(1, home // 4), run)
    assert tiny.stats()["error_flags"] == 0
    assert np.array_equal(tiny.download(A.POSITIONS), ref.download(A.POSITIONS))


def test_sparse_scene_needs_more_slabs_than_the_estimate():
    """Eight separate sheets on a 256^3 grid: ~110 particles per occupied block, 8.7k home blocks against a pool
    estimated at ~5k slabs (particles / 256 + 1024).  mpm_finalize must size the pool from what the first re-sort
    reports (this used to end in MPM_ERR_CAPACITY for good), with no environment override."""
    from drake_amd import ARR as A, GpuMpm, scenes
    assert "MPM_SLAB_CAPACITY" not in os.environ
    g = GpuMpm(8)
    for k in range(8):
        (pos, vel, idx), = scenes.cloth_stack(1, 200, 8, z0=0.3 + k * 4.0 / 256.0, seed=11 + k, vel_amp=0.05)
        g.add_qr_cloth(pos, vel, idx)
    g.finalize()
    st = g.stats()
    assert st["error_flags"] == 0
    assert st["home_blocks"] > 8000, st
    npart = g.n_particles
    assert npart / st["home_blocks"] < 200
    assert st["home_blocks"] > max(4096, npart // 256 + 1024 + npart // (64 * 48))   # beyond the initial estimate
    dt = 2e-4
    g.run_substeps(10, dt, -1)
    g.gpu_sync()
    v = g.download(A.VELOCITIES)
    assert g.stats()["error_flags"] == 0 and np.isfinite(v).all()
    # free fall: every sheet has gained g * t (the sheets do not touch anything)
    assert abs(float(v[:, 2].mean()) + 9.8 * 10 * dt) < 2e-3


def test_cloth_that_spreads_inside_one_batch_recovers():
    """The slab pool overflows in the middle of ONE mpm_run_substeps batch (no synchronisation in between): the
    substeps after the overflowing re-sort skip themselves, the next synchronising call grows the pool, repeats
    the re-sort and runs them; the result equals a run with a pool that was large enough from the start."""
    from drake_amd import ARR as A, GpuMpm, scenes

    def sheets():
        # eight small sheets stacked in the same blocks fly apart in eight directions
        out = scenes.cloth_stack(8, 20, 6, z0=0.5, side=0.12, seed=3, vel_amp=0.0)
        for k, (pos, vel, idx) in enumerate(out):
            vel[:, 0] = 2.5 * np.cos(2 * np.pi * k / 8)
            vel[:, 1] = 2.5 * np.sin(2 * np.pi * k / 8)
        return out

    def run():
        g = GpuMpm(6)
        g.set_deterministic(True)
        scenes.populate(g, sheets())
        h0 = g.stats()["home_blocks"]
        g.run_substeps(60, 1e-3, -1)      # one batch
        g.gpu_sync()
        return g, h0

    ref, h0 = run()
    h1 = ref.stats()["home_blocks"]
    assert h1 > 1.3 * h0, (h0, h1)        # the cloth did spread over more blocks
    small, _ = _with_env("MPM_SLAB_CAPACITY", int(h0 * 1.1), run)
    assert small.stats()["error_flags"] == 0
    assert small.stats()["substeps"] == 60
    assert np.array_equal(small.download(A.POSITIONS), ref.download(A.POSITIONS))


def test_slab_list_overflow_of_a_split_dense_pile_is_reported():
    """k_grid lists at most 160 slabs over one block (27 neighbours x their work items).  With items of
    one wave group each (MPM_ITEM_GROUPS=1) a dense pile exceeds that: MPM_ERR_CAPACITY, not a wrong grid."""
    from drake_amd import GpuMpm, MpmError, scenes

    def run():
        g = GpuMpm(6)
        # ~30 sheets over 3 blocks of height: > 7 groups of 64 particles in every block around the centre
        scenes.populate(g, scenes.cloth_stack(30, 60, 6, z0=0.45, side=0.3, seed=9, vel_amp=0.0))
        g.run_substeps(2, DT, -1)
        g.gpu_sync()
        return g

    g = run()                                   # default items (48 groups): fine
    assert g.stats()["error_flags"] == 0
    with pytest.raises(MpmError) as ei:
        _with_env("MPM_ITEM_GROUPS", 1, run)
    assert ei.value.code == -4
    # two groups per item halve the slab count: inside the list again
    g2 = _with_env("MPM_ITEM_GROUPS", 4, run)
    assert g2.stats()["error_flags"] == 0
