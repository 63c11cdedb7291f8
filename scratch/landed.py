import sys
sys.path.insert(0, '.')
import numpy as np
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS['cloth_1m']
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
for n in (220, 200, 200):
    g.run_substeps(n, 1e-3, -1)
    pos = g.sync_particle_state_to_cpu()
    st = g.stats()
    c = np.floor(pos * (1 << bits) - 0.5).astype(np.int64)
    key = (c[:, 0] << 20) | (c[:, 1] << 10) | c[:, 2]
    u, cnt = np.unique(key, return_counts=True)
    bkey = ((c[:, 0] >> 2) << 20) | ((c[:, 1] >> 2) << 10) | (c[:, 2] >> 2)
    ub, bcnt = np.unique(bkey, return_counts=True)
    print(st['substeps'], 'z range', pos[:, 2].min(), pos[:, 2].max(), 'cells', len(u), 'ppc mean', cnt.mean(), 'max', cnt.max(),
          'blocks', len(ub), 'ppb mean', bcnt.mean(), 'max', bcnt.max(), 'pct blocks<256', (bcnt < 256).mean(), st)
