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
    # a pool that cannot even hold the first re-sort's items is a capacity error, not silent garbage
    from drake_amd import MpmError
    with pytest.raises(MpmError) as ei:
        _with_env("MPM_SLAB_CAPACITY", max(1, home // 4), run)
    assert ei.value.code == -4


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
