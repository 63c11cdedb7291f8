"""What the idle re-sort launches cost: the first 25 substeps of the benchmark scene need no re-sort; MPM_RESORT_EVERY
= 1, 2, 4, 8, 1000 (set by the caller) changes only how often the four idle kernels are launched."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
out = []
for rep in range(5):
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits))
    g.run_substeps(5, 1e-3, -1); g.gpu_sync()
    t0 = time.perf_counter(); g.run_substeps(20, 1e-3, -1); g.gpu_sync()
    out.append((time.perf_counter() - t0) / 20 * 1e6)
    assert g.stats()["rebuilds"] == 1
    g.destroy()
print("MPM_RESORT_EVERY", os.environ.get("MPM_RESORT_EVERY"), "us/substep", [round(x, 1) for x in out], "min", round(min(out), 1))
