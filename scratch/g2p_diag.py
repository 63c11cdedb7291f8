"""k_g2p: staging and particle-loop cycles of wave 0 of a few workgroups (diagnostic build, MPM_DBG=4096: unlike MPM_DBG=4, which
stamps every wave and slows the kernel several times, this leaves it alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPM_DBG"] = "4096"
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
g.run_substeps(20, 1e-3, -1); g.gpu_sync()
g.debug_counters(reset=True)
n = 20
g.run_substeps(n, 1e-3, -1)
c = g.debug_counters()
k = max(c[11], 1)
print("items stamped", c[11], "per item: staging %d cycles (descriptor + neighbour table %d, node values %d, rest = LDS writes + barrier), particle loop %d cycles, %.2f particles per thread" % (c[8] // k, c[12] // k, c[13] // k, c[9] // k, c[10] / k), flush=True)
ph, tot = g.profile_substeps(40, 1e-3, -1)
print("g2p event time with the stamps on: %.1f us" % (ph["g2p"] * 1e3))
