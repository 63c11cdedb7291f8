"""Soak of the coupled path on a partition: an in-process world of WORLD ranks, a cloth stack that lands on a floor while it
slides along x across the cuts (particles migrate between the ranks while they are in contact), STEPS coupled substeps in
batches; after every batch: error flags, exactly one owner per particle, the ranks' contacts and Newton iterations.
    python scratch/team_soak.py > profiles/r06_team_soak.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from drake_amd import Collider, GpuMpm, scenes  # noqa: E402
from drake_amd.dist import LocalWorld, strong_geometry  # noqa: E402

WORLD = int(os.environ.get("TEAM_SOAK_WORLD", "4"))
STEPS = int(os.environ.get("TEAM_SOAK_STEPS", "1000"))
BATCH = 50
bits, layers, res = scenes.CONFIGS[os.environ.get("TEAM_SOAK_CONFIG", "cloth_250k")]
floor_z, k, d, mu, dt = 0.25, 1e6, 1e-5, 0.3, 2e-4
sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z + 0.002)
for pos, vel, idx in sheets:
    vel[:, 2] -= 0.5
    vel[:, 0] += 1.5          # 0.04 cells per substep along x: 38 cells over 1000 substeps, across every cut
engines = []
for _ in range(WORLD):
    g = GpuMpm(bits)
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    g.reallocate_external_bodies(1)
    engines.append(g)
n = engines[0].n_particles
geo = strong_geometry(bits, WORLD)
w = LocalWorld(engines, geo["cuts"], geo["zone_blocks"], 0, 0, capacity_blocks=2048, migrate_every=0, migrate_capacity=1 << 17,
               device=torch.device("cuda", 0))
w.enable_team(2048)
floor = [Collider(0, body=0, p_WB=(0.5, 0.5, floor_z))]
print(f"team soak: {WORLD} ranks of {n} particles in one process, cuts at blocks {geo['cuts']}, dt {dt}, floor friction {mu}", flush=True)
done, t0 = 0, time.perf_counter()
owned0 = None
while done < STEPS:
    out = w.coupled_substeps(BATCH, dt, floor, mu, k, d)
    w.sync()
    done += BATCH
    owners = np.zeros(n, np.int32)
    held = []
    for g in engines:
        st = g.stats()
        assert st["error_flags"] == 0, (done, st)
        roles = g.dist_roles()
        owners += roles == 1
        held.append(int(np.count_nonzero(roles == 1)))
    assert np.all(owners == 1), (done, int(np.count_nonzero(owners != 1)))
    contacts = [sum(o[s]["contacts"] for o in out) for s in range(BATCH)]
    its = [out[0][s]["iterations"] for s in range(BATCH)]
    assert all(len({o[s]["iterations"] for o in out}) == 1 for s in range(BATCH))
    print(f"substep {done:5d}: owned per rank {held}, contacts {int(np.mean(contacts)):6d}, Newton iterations {np.mean(its):5.2f} (max {max(its)}), "
          f"migrations so far {w.migrations}, slot resizes {[g.dist_geometry()['slot_resizes'] for g in engines]}", flush=True)
print(f"{STEPS} coupled substeps in {time.perf_counter() - t0:.1f} s; every particle had exactly one owner after every batch, no error flag")
