"""ring of one (scratch/chain_cost.py) for a kernel trace: rocprofv3 --kernel-trace --stats -- python scratch/chain_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
nb = (1 << bits) // 4
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
g.chain_init(GpuMpm.chain_unique_id(), 0, 1, nb // 4, 3 * nb // 4, nb // 2, 2, 1024, periodic=True)
g.chain_substeps(5, 1e-3, -1); g.gpu_sync()
t0 = time.perf_counter(); g.chain_substeps(40, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
print("chain (ring of one): %.1f us/substep" % (el / 40 * 1e6), g.stats()["error_flags"], flush=True)
g.chain_destroy(); g.destroy()
