import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import build_pair
from tests.test_contact_gpu import floor_contacts, CONTACT_PARAMS, Z_FLOOR
from oracle import oracle as orc
from drake_amd import ARR as A
stiffness, damping, DT = CONTACT_PARAMS["config3"]
for exact in (False, True):
  for iters in (1, 2, 3, 4, 5):
    o, g = build_pair(layers=4, res=40, side=0.15, z0=Z_FLOOR - 0.02, vel_amp=0.3)
    o.vel[:, 2] -= 0.5; o.vel[:, 0] += 0.3
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    for s in (o, g):
        s.reallocate_external_bodies(1); s.rebuild_mapping(False); s.calc_fem_state_and_force(DT); s.particle_to_grid(DT); s.update_grid(-1)
    cp = floor_contacts(g.sync_particle_state_to_cpu())
    o.copy_contact_pairs(orc.ContactPairs(*cp)); g.copy_contact_pairs(*cp)
    ro = o.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_iters=iters)
    rg = g.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_newton_iterations=iters)
    cs = g.contact_stats()
    d = float(np.abs(g.download(A.GRID_DIR) - o.g_D).max() / np.abs(o.g_D).max())
    print("exact", exact, "cap", iters, "oracle", {k: ro[k] for k in ("iterations", "residual", "alpha", "E0", "E1", "ls_last")},
          "engine", rg, {k: cs[k] for k in ("alpha", "E0", "energy", "line_search_evals")}, "dir rel err %.2e" % d, flush=True)
