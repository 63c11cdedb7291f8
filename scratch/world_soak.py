"""Long run of a partitioned domain in one process (dist.LocalWorld): 4 ranks, a quarter-density copy of the benchmark's cloth stack
falling 400 substeps with a lateral drift onto the slip plane of scene 2, against a single engine: error flags, ownership, ghosts."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drake_amd import ARR, GpuMpm, scenes
from drake_amd.dist import LocalWorld, strong_geometry
bits, steps, dt, bc = 7, int(sys.argv[1]) if len(sys.argv) > 1 else 400, 1e-3, 2
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sheets = scenes.cloth_stack(4, 145, bits, z0=0.2, seed=1234)
for pos, vel, idx in sheets:
    vel[:, 0] += 0.5
def eng():
    g = GpuMpm(bits)
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    return g
ref = eng()
ref.run_substeps(steps, dt, bc); ref.gpu_sync()
print("single engine:", {k: ref.stats()[k] for k in ("rebuilds", "error_flags")}, flush=True)
geo = strong_geometry(bits, world)
w = LocalWorld([eng() for _ in range(world)], geo["cuts"], geo["zone_blocks"], 0, 0, capacity_blocks=1024, migrate_every=0,
               migrate_capacity=1 << 16, device=torch.device("cuda", 0))
t0 = time.perf_counter()
for s in range(0, steps, 50):
    w.run_substeps(min(50, steps - s), dt, bc); w.sync()
    st = [c.e.stats() for c in w.chains]
    print(f"substep {s + 50}: migrations {w.migrations}, rebuilds {[x['rebuilds'] for x in st]}, errors {[x['error_flags'] for x in st]}, "
          f"held {[x['active_faces'] + x['active_vertices'] for x in st]}, resizes {[c.e.dist_geometry()['slot_resizes'] for c in w.chains]}", flush=True)
    assert all(x["error_flags"] == 0 for x in st)
n = ref.n_particles
owners = np.zeros(n, int); pos = np.full((n, 3), np.nan, np.float32)
for c in w.chains:
    r = c.e.dist_roles(); own = r == 1; owners += own; pos[own] = c.e.download(ARR.POSITIONS)[own]
assert np.all(owners == 1)
d = np.abs(pos - ref.download(ARR.POSITIONS)).max()
print(f"max position difference to the single engine after {steps} substeps: {d:.3e} (wall {time.perf_counter() - t0:.1f} s)")
