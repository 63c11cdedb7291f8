"""The 256^3 phase-by-phase test scene on the float oracle, the double oracle and the engine: is the engine as close
to double as the float oracle is, and how far apart are the two float results?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, GpuMpm, scenes
from oracle import oracle as orc
for bc, z0, side, res in ((0, 0.56, 0.16, 40), (-1, 0.5, 0.16, 40)):
    sheets = scenes.cloth_stack(3, res, 8, z0=z0, side=side, seed=7, vel_amp=0.3)
    o32, o64, g = orc.OracleMpm(8), orc.OracleMpm(8, real=np.float64), GpuMpm(8)
    for pos, vel, idx in sheets:
        for s in (o32, o64, g): s.add_qr_cloth(pos, vel, idx)
    for s in (o32, o64, g): s.finalize()
    o32.vel[:, 2] -= 0.5; o64.vel[:, 2] -= 0.5
    for step in range(3):
        g.upload_particle_state(o32.pos, o32.vel, o32.C, None, o32.F)
        for name in ("pos", "vel", "C", "F"): setattr(o64, name, getattr(o32, name).astype(np.float64))
        for s in (o32, o64, g):
            s.rebuild_mapping(False); s.calc_fem_state_and_force(2e-4); s.particle_to_grid(2e-4); s.update_grid(bc); s.grid_to_particle(2e-4)
        v, v32, v64 = g.download(A.VELOCITIES).astype(np.float64), o32.vel.astype(np.float64), o64.vel
        f, f32, f64 = g.download(A.FORCES).astype(np.float64), o32.forces.astype(np.float64), o64.forces
        print(f"bc {bc} step {step}: vel |g-o64| {np.abs(v-v64).max():.2e} |o32-o64| {np.abs(v32-v64).max():.2e} |g-o32| {np.abs(v-v32).max():.2e}   "
              f"force |g-o64| {np.abs(f-f64).max():.2e} |o32-o64| {np.abs(f32-f64).max():.2e} |g-o32| {np.abs(f-f32).max():.2e} max|f| {np.abs(f64).max():.2e}", flush=True)
