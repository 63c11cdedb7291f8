"""Soak of the gated substeps: the same scene advanced by mpm_run_substeps in uneven batches (re-sort launches in
front of every 4th substep, skipped substeps re-run at synchronising calls) and phase by phase (re-sort check
before every substep); positions must agree and nothing may be flagged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, GpuMpm, scenes

def make():
    g = GpuMpm(7)
    sheets = scenes.cloth_stack(4, 100, 7, z0=0.45, vel_amp=0.3, seed=3)
    for pos, vel, idx in sheets:
        vel[:, 0] += 2.0
        vel[:, 1] -= 1.0
    scenes.populate(g, sheets)
    return g

DT, N = 5e-4, 360
a, b = make(), make()
rng = np.random.default_rng(0)
done = 0
while done < N:
    k = int(min(N - done, rng.integers(1, 9)))
    a.run_substeps(k, DT, 0)
    done += k
    if rng.random() < 0.2:
        a.download(A.POSITIONS)          # a synchronising call in the middle
for _ in range(N):
    b.rebuild_mapping(False); b.calc_fem_state_and_force(DT); b.particle_to_grid(DT); b.update_grid(0); b.grid_to_particle(DT)
sa, sb = a.stats(), b.stats()
xa, xb = a.download(A.POSITIONS), b.download(A.POSITIONS)
print("stats", sa, sb)
print("max |dx|", float(np.abs(xa - xb).max()), "mean drift", float(np.abs(xa - xb).mean()))
assert sa["error_flags"] == 0 and sb["error_flags"] == 0
# (the two runs re-sort at different substeps: different orders inside the cells, rounding differences that the cloth
# amplifies over 360 substeps -- 1.9e-4 .. 2.4e-4 from run to run, with the round-3 library as with this one;
# scratch/soak.py is the bit-for-bit check, in deterministic mode)
assert np.abs(xa - xb).max() < 5e-4
print("soak ok")
