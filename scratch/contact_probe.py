import sys
sys.path.insert(0, '.')
import numpy as np
from tests.helpers import build_pair
from tests.test_contact_gpu import floor_contacts, DT, Z_FLOOR
from drake_amd import ARR as A
from oracle import oracle as orc
mu, exact = 0.5, False
ref = None
inputs = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    o, g = build_pair(layers=2, res=20, z0=Z_FLOOR - 0.004, vel_amp=0.3)
    o.vel[:, 2] -= 0.5
    o.vel[:, 0] += 0.3
    log = []
    for step in range(3):
        if rep == 0:
            inputs.append([x.copy() for x in (o.pos, o.vel, o.C, o.F)])
        ip = inputs[step]
        g.upload_particle_state(ip[0], ip[1], ip[2], None, ip[3])
        g.reallocate_external_bodies(1)
        pos_g = g.sync_particle_state_to_cpu()
        g.rebuild_mapping(False); g.calc_fem_state_and_force(DT); g.particle_to_grid(DT); g.update_grid(-1)
        gv0 = g.download(A.GRID_V_STAR).copy()
        cp = floor_contacts(pos_g)
        g.copy_contact_pairs(*cp)
        rg = g.update_contact(DT, mu, 1e5, 1e-3, exact_line_search=exact)
        v = g.download(A.CONTACT_VEL)
        log.append((rg["iterations"], float(rg["residual"]), float(np.abs(v).sum()), float(np.nansum(np.abs(gv0))), bool(np.isnan(gv0).any())))
        g.grid_to_particle(DT)
        if rep == 0:
            o.reallocate_external_bodies(1)
            o.rebuild_mapping(False); o.calc_fem_state_and_force(DT); o.particle_to_grid(DT); o.update_grid(-1)
            o.copy_contact_pairs(orc.ContactPairs(*cp))
            o.update_contact(DT, mu, 1e5, 1e-3, exact_line_search=exact)
            o.grid_to_particle(DT)
    if ref is None:
        ref = log
        print('ref', log)
    elif log != ref:
        print('DIFF rep', rep, log)
    g.destroy()
print('done')
