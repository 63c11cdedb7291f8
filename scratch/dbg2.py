import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drake_amd import GpuMpm, scenes
g = GpuMpm(7)
scenes.populate(g, scenes.cloth_stack(16, 145, 7))
g.run_substeps(5, 1e-3, -1); g.gpu_sync(); g.debug_counters(True)
g.run_substeps(10, 1e-3, -1); g.gpu_sync()
c = g.debug_counters(True)
n = max(c[6], 1)
print("P2G wave-0 per group: derive+sort+stage %.0f cyc, contraction %.0f cyc (steps %.0f + epilogues %.0f); cells/group %.2f steps/group %.2f groups %d" % (c[0]/n, c[1]/n, c[2]/n, c[3]/n, c[4]/n, c[5]/n, n))
print("per step %.0f cyc, per cell epilogue %.0f cyc" % (c[2]/max(c[5],1), c[3]/max(c[4],1)))
print("wave-0 block time before final barrier %.0f cyc/block; whole block incl. slab %.0f cyc; blocks %d; groups per block (wave 0) %.2f" % (c[7]/max(c[13],1), c[12]/max(c[13],1), c[13], c[6]/max(c[13],1)))
