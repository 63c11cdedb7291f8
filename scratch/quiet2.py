import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPM_DBG"] = "32"
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
g.run_substeps(5, 1e-3, -1); g.gpu_sync()
g.run_substeps(20, 1e-3, -1); g.gpu_sync()
for _ in range(6):
    g.run_substeps(40, 1e-3, -1); g.gpu_sync()
print(g.stats())
