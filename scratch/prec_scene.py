"""A parity-test scene on the float oracle, the double oracle and the engine (phase-by-phase, re-synchronised to the
float oracle before every substep like the tests): distances of the velocities after the substep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, GpuMpm, scenes
from oracle import oracle as orc
bits, dt = int(sys.argv[1]), float(sys.argv[2])
for bc, z0, side, res in [tuple(float(x) if "." in x else int(x) for x in a.split(",")) for a in sys.argv[3:]]:
    sheets = scenes.cloth_stack(3, res, bits, z0=z0, side=side, seed=7, vel_amp=0.2)
    o32, o64, g = orc.OracleMpm(bits), orc.OracleMpm(bits, real=np.float64), GpuMpm(bits)
    for pos, vel, idx in sheets:
        for s in (o32, o64, g): s.add_qr_cloth(pos, vel, idx)
    for s in (o32, o64, g): s.finalize()
    for step in range(3):
        g.upload_particle_state(o32.pos, o32.vel, o32.C, None, o32.F)
        for name in ("pos", "vel", "C", "F"): setattr(o64, name, getattr(o32, name).astype(np.float64))
        for s in (o32, o64, g):
            s.rebuild_mapping(False); s.calc_fem_state_and_force(dt); s.particle_to_grid(dt); s.update_grid(bc); s.grid_to_particle(dt)
        v, v32, v64 = g.download(A.VELOCITIES).astype(np.float64), o32.vel.astype(np.float64), o64.vel
        i = int(np.abs(v - v32).max(1).argmax())
        print(f"bc {bc} step {step}: vel |g-o64| {np.abs(v-v64).max():.2e} |o32-o64| {np.abs(v32-v64).max():.2e} |g-o32| {np.abs(v-v32).max():.2e} max|v| {np.abs(v64).max():.2e}"
              f"  worst particle {i} ({'face' if i < g.n_faces else 'vertex'}) pos {o32.pos[i]}", flush=True)
