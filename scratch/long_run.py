import sys, time
sys.path.insert(0, '.')
import numpy as np
from drake_amd import GpuMpm, scenes, ARR as A
bits, layers, res = scenes.CONFIGS['cloth_1m']
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
vol = g.download(A.VOLUMES).astype(np.float64)
t = time.perf_counter()
for k in range(6):
    g.run_substeps(500, 1e-3, -1)
    pos = g.sync_particle_state_to_cpu()
    vel = g.download(A.VELOCITIES)
    st = g.stats()
    ke = 0.5 * 2000.0 * float((vol * (vel.astype(np.float64) ** 2).sum(1)).sum())
    print(st['substeps'], 'finite', bool(np.isfinite(pos).all() and np.isfinite(vel).all()), 'z', float(pos[:, 2].min()), float(pos[:, 2].max()),
          'xy', float(pos[:, :2].min()), float(pos[:, :2].max()), 'KE', ke, 'rebuilds', st['rebuilds'], 'blocks', st['home_blocks'], 'err', st['error_flags'],
          'elapsed', round(time.perf_counter() - t, 2))
