"""In-process world of two ranks, coupled substeps on a floor (config-3 parameters), for a kernel trace:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 $GRAFT_REPO_ROOT/scratch/team_world_prof.py
scripts/team_trace_summary.py turns the trace into profiles/r06_team_world_trace.txt: launches per coupled substep by
kernel, and the blit copies (copyBuffer / fillBuffer) in the window -- the host reads nothing back inside a solve."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from drake_amd import Collider, GpuMpm, scenes  # noqa: E402
from drake_amd.dist import LocalWorld, strong_geometry  # noqa: E402

CONFIG = os.environ.get("TEAM_PROF_CONFIG", "cloth_250k")
WORLD = int(os.environ.get("TEAM_PROF_WORLD", "2"))
STEPS = int(os.environ.get("TEAM_PROF_STEPS", "40"))
bits, layers, res = scenes.CONFIGS[CONFIG]
floor_z, k, d, mu, dt = 0.25, 1e6, 1e-5, 1.0, 2e-4
sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z - 0.004)
for pos, vel, idx in sheets:
    vel[:, 2] -= 0.5
engines = []
for _ in range(WORLD):
    g = GpuMpm(bits)
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    g.reallocate_external_bodies(1)
    engines.append(g)
geo = strong_geometry(bits, WORLD)
w = LocalWorld(engines, geo["cuts"], geo["zone_blocks"], 0, 0, capacity_blocks=1024, migrate_every=0, migrate_capacity=1 << 16,
               device=torch.device("cuda", 0))
w.enable_team(1024)
floor = [Collider(0, body=0, p_WB=(0.5, 0.5, floor_z))]
w.coupled_substeps(5, dt, floor, mu, k, d)
w.sync()
m0 = w.migrations
t0 = time.perf_counter()
out = w.coupled_substeps(STEPS, dt, floor, mu, k, d)
w.sync()
el = time.perf_counter() - t0
its = [r["iterations"] for r in out[0]]
print(f"team world: {WORLD} ranks of {CONFIG} on one GPU, {STEPS} coupled substeps: {el / STEPS * 1e3:.3f} ms per substep, "
      f"{np.mean(its):.2f} Newton iterations, {np.mean([sum(o[s]['contacts'] for o in out) for s in range(STEPS)]):.0f} contacts, "
      f"{w.migrations - m0} migrations in the window (each a synchronisation point with device-to-device copies, outside the solves)")
for g in engines:
    assert g.stats()["error_flags"] == 0
