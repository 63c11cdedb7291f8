"""What a rank of the chain runs per substep, measured on one GPU: the 1M workload as a ring of one (the rank is its own
neighbour on both sides) -- over RCCL (send / recv to itself on the engine's stream) and over the DIRECT exchange (stores
into the "neighbour's" buffer + sequence flags, mpm_chain_direct_*) -- against the plain batched substeps, all engines alive
at once and timed in turn, several rounds (the boxes of the pool differ by +-8 us: only figures of one run compare).
(profiles/r06_ring_of_one_interior_ab.txt: the same with a fourth engine whose first grid kernel updated the blocks outside
the zones itself, so that the kernel behind the exchange touched the zone blocks only -- 112.8 / 115.5 / 117.8 / 117.5 us
against 113.5 / 115.4 / 115.8 / 117.4: nothing; the change was taken out again, DESIGN_HISTORY.md section 8.)"""
import os, sys, time
import torch  # noqa: F401  (before RCCL is bound)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
nb = (1 << bits) // 4
N = int(os.environ.get("CHAIN_COST_SUBSTEPS", "40"))
def engine(env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        g = GpuMpm(bits)   # (environment switches are read per handle)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    return g
def ring(g, direct):
    g.chain_init(None if direct else GpuMpm.chain_unique_id(), 0, 1, nb // 4, 3 * nb // 4, nb // 2, 2, 1024, periodic=True)
    if direct:
        g.chain_direct_prepare(); g.chain_direct_connect(None, None)
    return g
runs = {
    "plain (mpm_run_substeps)": (engine(), lambda g, n: g.run_substeps(n, 1e-3, -1)),
    "ring of one, RCCL": (ring(engine(), False), lambda g, n: g.chain_substeps(n, 1e-3, -1)),
    "ring of one, direct": (ring(engine(), True), lambda g, n: g.chain_substeps(n, 1e-3, -1)),
}
for name, (g, f) in runs.items():
    f(g, 5); g.gpu_sync()
for rnd in range(4):
    for name, (g, f) in runs.items():
        t0 = time.perf_counter(); f(g, N); g.gpu_sync(); el = time.perf_counter() - t0
        print(f"round {rnd}: {name}: {el / N * 1e6:.1f} us/substep (error flags {g.stats()['error_flags']}, re-sorts so far {g.stats()['rebuilds']})", flush=True)
