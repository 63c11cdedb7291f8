"""Soak of the coupled path on BASELINE config 3: N coupled substeps in batches, error flags and the solve counters after
every batch (refused solves, repeated overflows), contact counts and iterations; ends with the seven-call pattern on the
same engine for a few substeps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import Collider, GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
floor_z, k, d, mu, dt = 0.25, 1e6, 1e-5, 1.0, 2e-4
floor = (Collider * 1)(Collider(0, body=0, p_WB=(0.5, 0.5, floor_z)))
g = GpuMpm(bits)
sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z - 0.004)
for pos, vel, idx in sheets:
    vel[:, 2] -= 0.5
scenes.populate(g, sheets)
g.reallocate_external_bodies(1)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
batch = 50
t0 = time.perf_counter()
for b in range(total // batch):
    rs = g.run_coupled_substeps(batch, dt, floor, mu, k, d)
    g.gpu_sync()
    st = g.stats()
    assert st["error_flags"] == 0, st
    it = [r["iterations"] for r in rs]
    ct = [r["contacts"] for r in rs]
    print(f"batch {b:3d}: contacts {min(ct):6d}..{max(ct):6d} iterations mean {np.mean(it):6.1f} max {max(it):4d} reused {sum(r['setup_reused'] for r in rs):2d} "
          f"rebuilds {st['rebuilds']} counters {g.contact_counters()} {(time.perf_counter() - t0) / ((b + 1) * batch) * 1e3:.3f} ms/substep", flush=True)
for s in range(10):
    g.rebuild_mapping(False); g.calc_fem_state_and_force(dt); g.particle_to_grid(dt); g.update_grid(-1)
    g.generate_contact_pairs(floor, want_count=False)
    r = g.update_contact(dt, mu, k, d)
    g.grid_to_particle(dt)
g.gpu_sync()
assert g.stats()["error_flags"] == 0
print("soak ok")
