"""What a rank of the weak-scaling chain runs per substep, measured on one GPU: the 1M workload as a ring of one (the
rank is its own neighbour on both sides) -- with RCCL sending to itself, and with the DIRECT exchange (stores into the
"neighbour's" buffer + sequence flags, mpm_chain_direct_*) -- against the plain batched substeps."""
import os, sys, time
import torch  # noqa: F401  (before RCCL is bound)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
nb = (1 << bits) // 4
def engine():
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    return g
g = engine()
g.chain_init(GpuMpm.chain_unique_id(), 0, 1, nb // 4, 3 * nb // 4, nb // 2, 2, 1024, periodic=True)
g.chain_substeps(5, 1e-3, -1); g.gpu_sync()
for k in range(3):
    t0 = time.perf_counter(); g.chain_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    print("chain (ring of one): %.1f us/substep" % (el / 20 * 1e6), g.stats()["error_flags"], flush=True)
g.chain_destroy(); g.destroy()
g = engine()
g.chain_init(None, 0, 1, nb // 4, 3 * nb // 4, nb // 2, 2, 1024, periodic=True)
g.chain_direct_prepare(); g.chain_direct_connect(None, None)
g.chain_substeps(5, 1e-3, -1); g.gpu_sync()
for k in range(3):
    t0 = time.perf_counter(); g.chain_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    print("chain (ring of one, direct): %.1f us/substep" % (el / 20 * 1e6), g.stats()["error_flags"], flush=True)
g.chain_destroy(); g.destroy()
g = engine()
g.run_substeps(5, 1e-3, -1); g.gpu_sync()
for k in range(3):
    t0 = time.perf_counter(); g.run_substeps(20, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    print("plain: %.1f us/substep" % (el / 20 * 1e6), flush=True)
