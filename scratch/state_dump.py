import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, GpuMpm, scenes
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = GpuMpm(7)
g.set_deterministic(True)
sheets = scenes.cloth_stack(4, 120, 7, z0=0.5, vel_amp=0.5, seed=9)
for pos, vel, idx in sheets:
    vel[:, 0] += 3.0
scenes.populate(g, sheets)
g.run_substeps(n, 5e-4, -1)
np.savez(sys.argv[1], x=g.download(A.POSITIONS), v=g.download(A.VELOCITIES), C=g.download(A.AFFINE), F=g.download(A.DEFORMATION_GRADIENTS))
