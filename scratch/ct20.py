"""Where do engine and oracle part ways in a capped backtracking solve (tests/test_contact_gpu.py, dense scene)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import build_pair, oracle_copy
from tests.test_contact_gpu import CONTACT_PARAMS, Z_FLOOR, floor_contacts
from drake_amd import ARR as A
from oracle import oracle as orc
stiffness, damping, DT = CONTACT_PARAMS["config3"]
o, g = build_pair(layers=4, res=40, side=0.15, z0=Z_FLOOR - 0.02, vel_amp=0.3)
o.vel[:, 2] -= 0.5
o.vel[:, 0] += 0.3
g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
for s in (o, g):
    s.reallocate_external_bodies(1)
    s.rebuild_mapping(False)
    s.calc_fem_state_and_force(DT)
    s.particle_to_grid(DT)
    s.update_grid(-1)
cp = floor_contacts(g.sync_particle_state_to_cpu())
for iters in (5, 6, 7, 8, 9, 10, 12, 14, 16, 20):
    oc = oracle_copy(o, np.float32)
    oc.copy_contact_pairs(orc.ContactPairs(*cp))
    g.update_grid(-1)
    g.copy_contact_pairs(*cp)
    ro = oc.update_contact(DT, 1.0, stiffness, damping, exact_line_search=False, max_iters=iters)
    rg = g.update_contact(DT, 1.0, stiffness, damping, exact_line_search=False, max_newton_iterations=iters)
    cs = g.contact_stats()
    dD = float(np.abs(g.download(A.GRID_DIR) - oc.g_D).max()) / float(np.abs(oc.g_D).max())
    wgt = (oc.g_m / oc.g_m.max())[:, None]
    dv = float(np.abs((g.download(A.GRID_MOMENTUM) - oc.g_mv) * wgt).max())
    print(f"iters {iters:2d}: oracle it {ro['iterations']} alpha {ro['alpha']:.4g} res {ro['residual']:.6g} E0 {ro['E0']:.8g} E1 {ro['E1']:.8g} ls {ro.get('line_search_evals')} | "
          f"engine it {rg['iterations']} alpha {cs['alpha']:.4g} res {rg['residual']:.6g} E0 {cs['E0']:.8g} E {cs['energy']:.8g} ls {cs['line_search_evals']} | dDir {dD:.2e} dv {dv:.2e}", flush=True)
