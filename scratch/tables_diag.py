"""Phase stamps of k_rb_tables (diagnostic build, MPM_DBG=512): cycles since the kernel's start at the end of each phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPM_DBG"] = "512"
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
    names = ["forget", "A home list", "A2 ranges", "C union", "D active list", "F items", "order", "flat", "", "", "", "", "", "", "", "B (wg 1)"]
    prev = 0
    out = []
    for k in range(8):
        out.append(f"{names[k]} {c[k] - prev}")
        prev = c[k]
    print("cycles:", "; ".join(out), "| total", c[7], "| B", c[15], flush=True)
