import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
g.run_substeps(20, 1e-3, -1)
ph, tot = g.profile_substeps(200, 1e-3, -1)
print("r01" if os.environ.get("MPM_P2G_R01") else "new", {k: round(v * 1e3, 1) for k, v in ph.items()}, round(tot * 1e3, 1))
