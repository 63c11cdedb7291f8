"""P2G / G2P cycle counters of the diagnostic build (python -m drake_amd._build --diag first).
MPM_DBG=4 must be set in the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MPM_DBG", "4")
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits))
g.run_substeps(20, 1e-3, -1)
g.gpu_sync()
g.debug_counters(reset=True)
n = 20
g.run_substeps(n, 1e-3, -1)
c = g.debug_counters()
st = g.stats()
print("stats", st)
names = ["derive+group+stage", "contraction", "mfma loop", "epilogue", "cells", "steps", "groups", "block(w0)",
         "g2p stage", "g2p gather", "g2p iters", "g2p items", "p2g item total", "p2g items", "p2g pre-loop (t0)",
         "p2g wave0 barrier wait"]
for k, v in zip(names, c):
    print(f"{k:22s} {v:>16d}  per substep {v / n:14.1f}")
groups, cells, steps = c[6], c[4], c[5]
print("per group: derive %.0f contraction %.0f (mfma %.0f epilogue %.0f) cycles; cells/group %.2f steps/cell %.2f" %
      (c[0] / groups, c[1] / groups, c[2] / groups, c[3] / groups, cells / groups, steps / cells))
print("per item: total %.0f cycles, wave0 block part %.0f; items/substep %.1f" % (c[12] / c[13], c[7] / c[13] , c[13] / n))
print("g2p per item: stage %.0f gather %.0f iters %.2f" % (c[8] / c[11], c[9] / c[11], c[10] / c[11]))
ph, tot = g.profile_substeps(50, 1e-3, -1)
print("diag-build phase times (us)", {k: round(v * 1e3, 1) for k, v in ph.items()}, round(tot * 1e3, 1))
print("p2g per item: pre-loop %.0f, wave-0 wait at the closing barrier %.0f" % (c[14] / c[13], c[15] / c[13]))
