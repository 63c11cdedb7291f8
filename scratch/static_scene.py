import sys, time
sys.path.insert(0, '.')
from drake_amd import GpuMpm, scenes
m = GpuMpm.default_material(); m.gravity = 0.0
g = GpuMpm(7, m)
bits, layers, res = scenes.CONFIGS['cloth_1m']
scenes.populate(g, scenes.cloth_stack(layers, res, bits, vel_amp=0.0))
g.run_substeps(20, 1e-3, -1); g.gpu_sync()
t = time.perf_counter(); g.run_substeps(300, 1e-3, -1); g.gpu_sync(); dt = time.perf_counter() - t
print('static scene us/step', dt / 300 * 1e6, g.stats()['rebuilds'])
