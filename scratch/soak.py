"""Long mixed run: batches of uneven length, phase-by-phase substeps, downloads in between, a collider table that moves
-- on an engine with this round's scheduling (quiet time, held-back phase calls, lean modes) and on one with all of it
switched off.  Deterministic mode: the two must agree to the bit at every look and at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, BC_TABLE, GpuMpm, GridCollider, scenes
DT = 5e-4
def engine(on):
    env = dict(MPM_QUIET_FACTOR="0.5" if on else "0", MPM_DEFER_PHASES="1" if on else "0", MPM_RESORT_EVERY="4" if on else "1")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g = GpuMpm(7)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    g.set_deterministic(True)
    sheets = scenes.cloth_stack(6, 120, 7, z0=0.55, vel_amp=0.4, seed=11)
    for pos, vel, idx in sheets:
        vel[:, 0] += 1.0
    scenes.populate(g, sheets)
    return g
a, b = engine(True), engine(False)
rng = np.random.default_rng(5)
done, looks = 0, 0
t0 = time.time()
while done < 3000:
    mode = int(rng.integers(0, 10))
    k = int(rng.integers(1, 60))
    tb = [GridCollider(shape=1, mode=1, p=(0.5, 0.5, 0.30), n=(0.0, 0.0, 1.0), v=(0.0, 0.0, 0.0), friction=0.4),
          GridCollider(shape=0, mode=1, p=(0.3 + 0.0001 * done, 0.5, 0.36), radius=0.07, v=(0.2, 0.0, 0.0), friction=0.3)]
    for g in (a, b):
        g.set_grid_colliders(tb)
        if mode < 6:
            g.run_substeps(k, DT, BC_TABLE)
        else:
            for _ in range(min(k, 12)):
                g.rebuild_mapping(False); g.calc_fem_state_and_force(DT); g.particle_to_grid(DT); g.update_grid(BC_TABLE); g.grid_to_particle(DT)
    done += k if mode < 6 else min(k, 12)
    if rng.random() < 0.3:
        arr = [A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS, A.FORCES][int(rng.integers(0, 5))]
        xa, xb = a.download(arr), b.download(arr)
        assert np.isfinite(xa).all(), (done, arr)
        assert np.array_equal(xa, xb), (done, arr, float(np.abs(xa - xb).max()))
        looks += 1
sa, sb = a.stats(), b.stats()
for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS):
    assert np.array_equal(a.download(arr), b.download(arr)), arr
print("soak ok: substeps", sa["substeps"], sb["substeps"], "re-sorts", sa["rebuilds"], sb["rebuilds"], "check launches", sa["resort_checks"], sb["resort_checks"],
      "looks", looks, "errors", sa["error_flags"], sb["error_flags"], "z range", float(a.download(A.POSITIONS)[:, 2].min()), "wall %.1fs" % (time.time() - t0))
