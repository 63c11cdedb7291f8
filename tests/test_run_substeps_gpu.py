"""mpm_run_substeps against the phase-by-phase calls it batches: gated re-sort launches (substeps that find a
re-sort pending are deferred and run again by the next synchronising call), GridToParticle's lean mode between the
substeps of a batch, a collider table that changes between batches, and GpuSync() without a state argument
(cuda_mpm_solver.cu:164-166, called as in cuda_mpm_test.cc:73)."""
import numpy as np
import pytest

from drake_amd import scenes

pytestmark = pytest.mark.gpu

DT = 5e-4


def _engine(deterministic=True):
    from drake_amd import GpuMpm
    g = GpuMpm(7)
    # (the re-sort orders every cell's particles by their previous slot: trajectories are then a pure function of the
    # call sequence's arithmetic, and the batched and the phase-by-phase runs can be compared to the bit)
    g.set_deterministic(deterministic)
    sheets = scenes.cloth_stack(4, 100, 7, z0=0.45, vel_amp=0.3, seed=3)
    for pos, vel, idx in sheets:
        vel[:, 0] += 2.0     # fast sideways: a re-sort every few substeps
        vel[:, 1] -= 1.0
    scenes.populate(g, sheets)
    return g


def _phase_substep(g, bc):
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(DT)
    g.particle_to_grid(DT)
    g.update_grid(bc)
    g.grid_to_particle(DT)


def test_uneven_batches_match_phase_calls():
    from drake_amd import ARR as A
    a, b = _engine(), _engine()
    n, done = 240, 0
    rng = np.random.default_rng(0)
    while done < n:
        k = int(min(n - done, rng.integers(1, 9)))
        a.run_substeps(k, DT, 0)
        done += k
        if rng.random() < 0.2:
            a.download(A.POSITIONS)          # a synchronising call in the middle
    for _ in range(n):
        _phase_substep(b, 0)
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["substeps"] == sb["substeps"] == n
    assert sa["rebuilds"] > 5, sa                      # the scene does exercise the deferral
    assert sa["rebuilds"] == sb["rebuilds"]
    # same arithmetic in the same order (fused vertex forces, lean GridToParticle and deferred substeps included)
    # (the forces too: the last substep of a batch writes them out, the others keep them inside ParticleToGrid)
    for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS, A.FORCES, A.TAUS):
        xa, xb = a.download(arr), b.download(arr)
        assert np.isfinite(xa).all()
        assert np.array_equal(xa, xb), (arr, float(np.abs(xa - xb).max()))


def test_last_substep_of_a_batch_writes_the_face_velocities():
    """GridToParticle skips the position and velocity records of face particles between the substeps of a batch
    (nothing inside a substep reads them); what a caller downloads after the batch is the last substep's, as after
    phase calls."""
    from drake_amd import ARR as A
    a, b = _engine(), _engine()
    a.run_substeps(7, DT, -1)
    for _ in range(7):
        _phase_substep(b, -1)
    nf = a.n_faces
    idx_a, idx_b = a.download(A.INDEX_MAPPINGS), b.download(A.INDEX_MAPPINGS)
    va, vb = a.download(A.VELOCITIES), b.download(A.VELOCITIES)
    ca, cb = a.download(A.AFFINE), b.download(A.AFFINE)
    xa, xb = a.download(A.POSITIONS), b.download(A.POSITIONS)
    assert np.array_equal(idx_a, idx_b)
    faces = idx_a[:nf]
    scale = float(np.abs(vb).max())
    assert float(np.abs(va[faces] - vb[faces]).max()) < 1e-4 * scale
    assert float(np.abs(ca[faces] - cb[faces]).max()) < 1e-4 * float(np.abs(cb).max())
    # (the positions of face particles are left out between the substeps of a batch as well)
    assert float(np.abs(xa[faces] - xb[faces]).max()) < 1e-6
    # one substep earlier the face velocities were different (the test would not notice a stale record otherwise)
    c = _engine()
    for _ in range(6):
        _phase_substep(c, -1)
    vc = c.download(A.VELOCITIES)
    assert float(np.abs(vc[faces] - vb[faces]).max()) > 1e-3 * scale
    assert float(np.abs(c.download(A.POSITIONS)[faces] - xb[faces]).max()) > 1e-5


def test_collider_table_changed_between_batches():
    """mpm_set_grid_colliders between two mpm_run_substeps batches: substeps deferred by the first batch are run with
    the table they were enqueued with (they used to be replayed against the new one)."""
    from drake_amd import ARR as A, BC_TABLE, GridCollider
    a, b = _engine(), _engine()
    tables = [[GridCollider(shape=0, mode=1, p=(0.5 + 0.02 * k, 0.5, 0.42), radius=0.08, v=(0.3, 0.0, 0.0), friction=0.3)]
              for k in range(12)]
    rng = np.random.default_rng(1)
    counts = [int(rng.integers(7, 16)) for _ in tables]
    for tb, k in zip(tables, counts):
        a.set_grid_colliders(tb)
        a.run_substeps(k, DT, BC_TABLE)
    for tb, k in zip(tables, counts):
        b.set_grid_colliders(tb)
        for _ in range(k):
            _phase_substep(b, BC_TABLE)
    assert a.stats()["rebuilds"] > 2     # (re-sorts happened inside the batches)
    assert np.array_equal(a.download(A.POSITIONS), b.download(A.POSITIONS))
    assert np.array_equal(a.download(A.VELOCITIES), b.download(A.VELOCITIES))


def test_gpu_sync_without_a_state_argument():
    """solver.GpuSync() (no argument: the reference's cudaDeviceSynchronize) after mpm_run_substeps: every engine of
    the device is complete afterwards -- nothing owed, all substeps done, the stream idle."""
    from drake_amd import ARR as A, GpuMpm, capi
    # GpuSync() reports the sticky errors of EVERY engine alive on the device, like the reference's device-wide call
    # would surface a fault of any of them: engines that earlier tests left behind (kept alive by the tracebacks of
    # expected failures) are destroyed first
    import gc
    gc.collect()
    capi._destroy_live()
    a, other, b = _engine(), _engine(), _engine()
    for _ in range(12):
        a.run_substeps(5, DT, -1)
        other.run_substeps(3, DT, -1)
        GpuMpm.device_synchronize()
        assert a.owed_substeps() == 0 and other.owed_substeps() == 0
    for _ in range(60):
        _phase_substep(b, -1)
    # read the raw state WITHOUT another settling call in between: owed == 0 means the download below adds nothing
    sa = a.stats()
    assert sa["substeps"] == 60 and sa["error_flags"] == 0 and sa["rebuilds"] > 2
    assert np.array_equal(a.download(A.POSITIONS), b.download(A.POSITIONS))


def test_quiet_time_spares_the_check_launches():
    """The re-sort estimates how long no particle can leave its tile if all of them keep moving ballistically
    (Ctl::quiet_time); while half of that lasts, mpm_run_substeps enqueues substeps without the launches of the
    conditional re-sort.  The estimate is a hint -- a wrong one defers substeps, it cannot change results -- so the
    test checks (a) that it is a lower bound of the time to the first re-sort of a falling cloth, (b) that the
    launches are indeed left out, (c) that the trajectory is the one of an engine that never trusts it."""
    import os
    from drake_amd import ARR as A, GpuMpm

    def engine(factor):
        old = os.environ.get("MPM_QUIET_FACTOR")
        os.environ["MPM_QUIET_FACTOR"] = factor
        try:
            g = GpuMpm(7)
        finally:
            if old is None:
                del os.environ["MPM_QUIET_FACTOR"]
            else:
                os.environ["MPM_QUIET_FACTOR"] = old
        g.set_deterministic(True)
        scenes.populate(g, scenes.cloth_stack(4, 100, 7, z0=0.6, vel_amp=0.05, seed=5))
        return g

    a, b, c = engine("0.5"), engine("0"), engine("0.5")
    # (read from c: every call other than mpm_run_substeps drops the hint -- it may change the state -- and the
    # substeps after it get their check launches; a and b go from Finalize straight into the batch)
    quiet = c.stats()["quiet_time_s"]
    assert 0.02 < quiet < 0.2, quiet          # ~ sqrt(2 * 1.9 cells * dx / g) = 0.055 s on this grid
    n = 24
    a.run_substeps(n, DT, -1)
    b.run_substeps(n, DT, -1)
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["resort_checks"] <= 2 and sb["resort_checks"] >= n // 4, (sa, sb)
    assert sa["rebuilds"] == sb["rebuilds"]
    for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS):
        assert np.array_equal(a.download(arr), b.download(arr)), arr
    # (a) phase by phase (a check with every substep): the first re-sort after Finalize's does not come earlier
    r0 = c.stats()["rebuilds"]
    first = None
    for s in range(400):
        _phase_substep(c, -1)
        if c.stats()["rebuilds"] > r0:
            first = s          # the re-sort ran at the head of substep s: s substeps had been completed before it
            break
    assert first is not None
    assert first * DT >= 0.9 * quiet, (first, quiet)
    assert first * DT <= 3.0 * quiet, (first, quiet)   # ... and it is not uselessly small either


def test_phase_calls_held_back_until_the_substep_is_complete():
    """The reference's five calls per substep launch nothing until GridToParticle arrives and then go the way of
    mpm_run_substeps(1); a call that looks at an intermediate state first launches what was held back.  Against an
    engine that launches every call at once (MPM_DEFER_PHASES=0): the same bits everywhere a caller can look."""
    import os
    from drake_amd import ARR as A, GpuMpm

    def engine(defer):
        old = os.environ.get("MPM_DEFER_PHASES")
        os.environ["MPM_DEFER_PHASES"] = defer
        try:
            g = GpuMpm(7)
        finally:
            if old is None:
                del os.environ["MPM_DEFER_PHASES"]
            else:
                os.environ["MPM_DEFER_PHASES"] = old
        g.set_deterministic(True)
        sheets = scenes.cloth_stack(4, 100, 7, z0=0.45, vel_amp=0.3, seed=3)
        for pos, vel, idx in sheets:
            vel[:, 0] += 2.0
        scenes.populate(g, sheets)
        return g

    a, b = engine("1"), engine("0")
    looks = {
        1: (A.PIDS, A.INDEX_MAPPINGS),                      # after RebuildMapping
        2: (A.FORCES, A.TAUS, A.DEFORMATION_GRADIENTS),     # after CalcFemStateAndForce
        3: (A.GRID_MASSES, A.GRID_MOMENTUM),                # after ParticleToGrid
        4: (A.GRID_MOMENTUM, A.GRID_V_STAR),                # after UpdateGrid
        5: (A.POSITIONS, A.VELOCITIES, A.AFFINE),           # after GridToParticle
    }
    rng = np.random.default_rng(2)
    for step in range(60):
        look_at = int(rng.integers(0, 8))                   # 0, 6, 7: a substep nobody looks into
        for g in (a, b):
            g.rebuild_mapping(False)
            if look_at == 1:
                got = [g.download(x) for x in looks[1]]
            g.calc_fem_state_and_force(DT)
            if look_at == 2:
                got = [g.download(x) for x in looks[2]]
            g.particle_to_grid(DT)
            if look_at == 3:
                got = [g.download(x) for x in looks[3]]
            g.update_grid(0)
            if look_at == 4:
                got = [g.download(x) for x in looks[4]]
            g.grid_to_particle(DT)
            if look_at == 5:
                got = [g.download(x) for x in looks[5]]
            if g is a:
                got_a = got if look_at in looks else None
        if look_at in looks:
            for x, ya, yb in zip(looks[look_at], got_a, got):
                assert np.array_equal(ya, yb), (step, look_at, x)
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["substeps"] == sb["substeps"] == 60 and sa["rebuilds"] == sb["rebuilds"] and sa["rebuilds"] > 2
    assert sa["resort_checks"] < sb["resort_checks"]        # (the held-back substeps went without most check launches)
    for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS):
        assert np.array_equal(a.download(arr), b.download(arr)), arr


def test_long_mixed_run_matches_an_engine_without_the_scheduling():
    """900 substeps as uneven batches and phase-by-phase substeps, a collider table that moves, downloads in between:
    the engine with the quiet time, the held-back phase calls, the lean modes and a re-sort check every fourth substep
    against one with all of that switched off (a check with every substep, every call launched at once) -- the same
    bits at every look and at the end (scratch/soak.py is the 3000-substep version)."""
    import os
    from drake_amd import ARR as A, BC_TABLE, GpuMpm, GridCollider

    def engine(on):
        env = dict(MPM_QUIET_FACTOR="0.5" if on else "0", MPM_DEFER_PHASES="1" if on else "0",
                   MPM_RESORT_EVERY="4" if on else "1")
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            g = GpuMpm(7)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        g.set_deterministic(True)
        sheets = scenes.cloth_stack(6, 120, 7, z0=0.55, vel_amp=0.4, seed=11)
        for pos, vel, idx in sheets:
            vel[:, 0] += 1.0
        scenes.populate(g, sheets)
        return g

    a, b = engine(True), engine(False)
    rng = np.random.default_rng(5)
    done = looks = 0
    while done < 900:
        mode, k = int(rng.integers(0, 10)), int(rng.integers(1, 60))
        tb = [GridCollider(shape=1, mode=1, p=(0.5, 0.5, 0.30), n=(0.0, 0.0, 1.0), v=(0.0, 0.0, 0.0), friction=0.4),
              GridCollider(shape=0, mode=1, p=(0.3 + 0.0001 * done, 0.5, 0.36), radius=0.07, v=(0.2, 0.0, 0.0), friction=0.3)]
        n = k if mode < 6 else min(k, 12)
        for g in (a, b):
            g.set_grid_colliders(tb)
            if mode < 6:
                g.run_substeps(n, DT, BC_TABLE)
            else:
                for _ in range(n):
                    _phase_substep(g, BC_TABLE)
        done += n
        if rng.random() < 0.3:
            arr = (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS, A.FORCES)[int(rng.integers(0, 5))]
            xa, xb = a.download(arr), b.download(arr)
            assert np.isfinite(xa).all(), (done, arr)
            assert np.array_equal(xa, xb), (done, arr, float(np.abs(xa - xb).max()))
            looks += 1
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["substeps"] == sb["substeps"] == done and sa["rebuilds"] == sb["rebuilds"] and sa["rebuilds"] > 5
    assert sa["resort_checks"] < sb["resort_checks"] // 2 and looks > 5
    for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS):
        assert np.array_equal(a.download(arr), b.download(arr)), arr
