"""A few substeps and one contact solve with MPM_ROCTX=1, for `rocprofv3 --marker-trace --kernel-trace`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPM_ROCTX", "1")
from drake_amd import Collider, GpuMpm, scenes
g = GpuMpm(6)
scenes.populate(g, scenes.cloth_stack(4, 60, 6, z0=0.5, vel_amp=0.1))
g.run_substeps(8, 1e-3, -1)
g.reallocate_external_bodies(1)
for _ in range(2):
    g.rebuild_mapping(False); g.calc_fem_state_and_force(1e-3); g.particle_to_grid(1e-3); g.update_grid(-1)
    n = g.generate_contact_pairs([Collider(0, body=0, p_WB=(0.5, 0.5, 0.505))])
    r = g.update_contact(1e-3, 0.5, 1e5, 1e-3)
    g.grid_to_particle(1e-3)
g.gpu_sync()
print("contacts", n, r)
