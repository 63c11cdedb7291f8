"""Phase stamps of one workgroup of k_ct_tile (diagnostic build, MPM_DBG=2048), config 3: mean cycles since the kernel's start
at: segments known (state + keys + constants loaded), contact velocities gathered (stencil nodes staged), records in LDS,
segment sums written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPM_DBG"] = "2048"
from drake_amd import Collider, GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
floor_z, dt = 0.25, 2e-4
g = GpuMpm(bits)
sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z - 0.004)
for pos, vel, idx in sheets:
    vel[:, 2] -= 0.5
scenes.populate(g, sheets)
g.reallocate_external_bodies(1)
floor = [Collider(0, body=0, p_WB=(0.5, 0.5, floor_z))]
for s in range(12):
    if s == 4: g.debug_counters(reset=True)
    g.rebuild_mapping(False); g.calc_fem_state_and_force(dt); g.particle_to_grid(dt); g.update_grid(-1)
    n = g.generate_contact_pairs(floor)
    r = g.update_contact(dt, 1.0, 1e6, 1e-5)
    g.grid_to_particle(dt)
c = g.debug_counters()
k = max(c[15], 1)
print("contacts", n, "launches stamped", c[15], "cycles: segments %d, velocities %d, records %d, sums %d" % tuple(x // k for x in c[:4]), flush=True)
