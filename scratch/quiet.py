"""Quiet-time hint: the driver's window with and without it, re-sort statistics, deferred substeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
def window(tag):
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    g.run_substeps(5, 1e-3, -1); g.gpu_sync()
    t0 = time.perf_counter(); g.run_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    r0 = g.stats()["rebuilds"]
    t0 = time.perf_counter()
    for _ in range(10):
        g.run_substeps(40, 1e-3, -1); g.gpu_sync()
    el2 = time.perf_counter() - t0
    st = g.stats()
    print(tag, "window us/substep %.1f" % (el / 20 * 1e6), "| 10 frames of 40: %.1f us/substep" % (el2 / 400 * 1e6), "rebuilds", st["rebuilds"] - r0, "err", st["error_flags"], flush=True)
    g.destroy()
for k in range(2):
    for f in ("0", "0.5", "0.75"):
        os.environ["MPM_QUIET_FACTOR"] = f
        window("factor " + f)
