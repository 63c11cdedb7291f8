"""Phase stamps of one wave of k_rb_count (diagnostic build, MPM_DBG=1024): cycles since the kernel's start when its loads
have arrived, after the ballot loops, when the atomics have returned and when its stores are acknowledged; two passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPM_DBG"] = "1024"
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
g.run_substeps(70, 1e-3, -1); g.gpu_sync()
for rep in range(3):
    r0 = g.stats()["rebuilds"]
    g.debug_counters(reset=True)
    while g.stats()["rebuilds"] == r0:
        g.run_substeps(4, 1e-3, -1); g.gpu_sync()
    c = g.debug_counters()
    print("cycles: pass 1 loads %d ballots %d atomics %d stores %d | pass 2 loads %d ballots %d atomics %d stores %d" % tuple(c[:8]), flush=True)
