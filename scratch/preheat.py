"""Does the driver's window (20 substeps after 5, fresh process) run slower than the same window on a warm GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
def window(tag, pre=0):
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    g.run_substeps(5, 1e-3, -1); g.gpu_sync()
    t0 = time.perf_counter(); g.run_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    st = g.stats()
    print(tag, "us/substep %.1f" % (el / 20 * 1e6), "rebuilds", st["rebuilds"], flush=True)
    g.destroy()
window("cold process")
window("second engine")
window("third engine")
g = GpuMpm(bits); scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
g.run_substeps(5, 1e-3, -1); g.gpu_sync()
for k in range(6):
    t0 = time.perf_counter(); g.run_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    print("consecutive window", k, "us/substep %.1f" % (el / 20 * 1e6), "rebuilds", g.stats()["rebuilds"], flush=True)
