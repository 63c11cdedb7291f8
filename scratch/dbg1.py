import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drake_amd import GpuMpm, scenes
g = GpuMpm(7)
scenes.populate(g, scenes.cloth_stack(16, 145, 7))
import drake_amd
def sync():
    try:
        g.gpu_sync()
    except drake_amd.MpmError as e:
        print('(ignored)', str(e)[:40])
g.run_substeps(5, 1e-3, -1); sync(); g.debug_counters(True)
g.run_substeps(10, 1e-3, -1); sync()
c = g.debug_counters(True)
print("per substep: cells", c[0]/10, "steps", c[1]/10, "wave-phase cycles", c[2]/10, "wave-chunks", c[3]/10)
print("cycles per wave-chunk", c[2]/max(c[3],1), "steps per wave-chunk", c[1]/max(c[3],1), "cells per wave-chunk", c[0]/max(c[3],1))

print("stamps: particles", c[8], "steps-cycles", c[9], "epilogue-cycles", c[10], "atomics-cycles", c[11])
print("g2p: tile-load cycles/wave-block", c[4]/max(c[7],1), "particle-loop cycles/wave-block", c[5]/max(c[7],1), "iterations/block", c[6]/max(c[7],1), "wave-blocks per substep", c[7]/10)
