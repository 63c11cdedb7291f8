"""Hash of the particle state after 40 deterministic substeps of a 128^3 scene (two re-sorts on the way): for checking that
a rewritten kernel gives the same bits (run once per library, MPM_HIP_LIBRARY)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from drake_amd import ARR as A, GpuMpm, scenes
g = GpuMpm(7)
g.set_deterministic(True)
sheets = scenes.cloth_stack(4, 120, 7, z0=0.5, vel_amp=0.5, seed=9)
for pos, vel, idx in sheets:
    vel[:, 0] += 3.0
scenes.populate(g, sheets)
g.run_substeps(40, 5e-4, -1)
h = hashlib.sha256()
for arr in (A.POSITIONS, A.VELOCITIES, A.AFFINE, A.DEFORMATION_GRADIENTS):
    h.update(np.ascontiguousarray(g.download(arr) + 0.0).tobytes())   # (+ 0.0: a negative zero hashes like a positive one)
st = g.stats()
print(os.path.basename(os.environ.get("MPM_HIP_LIBRARY", "default")), h.hexdigest()[:16], "rebuilds", st["rebuilds"], "err", st["error_flags"], flush=True)
