"""tests/test_parity_gpu.py::test_substep_equals_phase_calls N times in one process: the distribution of the difference
between mpm_substep and the five phase calls after 5 substeps (same arithmetic, different arrival order of the re-sort's
atomics), in units of the measured one-substep float noise."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import build_pair, natural_scales, NOISE_FLOOR, RTOL
from drake_amd import ARR as A
DT = 1e-3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for k in range(n):
    o, g1 = build_pair(seed=11)
    _, g2 = build_pair(seed=11)
    sc = natural_scales(o, DT)
    noise = sc["vel"] * RTOL / NOISE_FLOOR
    errs = []
    for s in range(steps):
        g1.substep(DT, -1)
        g2.rebuild_mapping(False); g2.calc_fem_state_and_force(DT); g2.particle_to_grid(DT); g2.update_grid(-1); g2.grid_to_particle(DT)
        v1, v2 = g1.download(A.VELOCITIES), g2.download(A.VELOCITIES)
        errs.append(float(np.abs(v1 - v2).max()) / noise)
    st1, st2 = g1.stats(), g2.stats()
    e5 = np.abs(v1 - v2).max(axis=1) / noise
    print(k, "particles beyond 0.4 / 4 noises:", int((e5 > 0.4).sum()), int((e5 > 4).sum()), "of", len(e5), end="  ")
    print("err/noise per substep", " ".join(f"{e:6.2f}" for e in errs), "rebuilds", st1["rebuilds"], st2["rebuilds"], "worst particle", int(np.abs(v1 - v2).max(axis=1).argmax()), flush=True)
    g1.destroy(); g2.destroy()
