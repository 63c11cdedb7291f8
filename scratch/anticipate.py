"""Horizon of the anticipatory binning (MPM_ANTICIPATE, substeps; set by the caller): re-sorts and time of the
200-substep figure (after 20), and of the following 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
g.run_substeps(20, 1e-3, -1); g.gpu_sync()
out = []
for k in range(2):
    r0 = g.stats()["rebuilds"]
    t0 = time.perf_counter(); g.run_substeps(200, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    out.append((round(el / 200 * 1e6, 1), g.stats()["rebuilds"] - r0))
print("MPM_ANTICIPATE", os.environ.get("MPM_ANTICIPATE"), "MPM_RESORT_EVERY", os.environ.get("MPM_RESORT_EVERY"), "(us/substep, re-sorts) per 200 substeps:", out, "err", g.stats()["error_flags"], flush=True)
