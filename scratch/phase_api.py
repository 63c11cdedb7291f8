"""The reference's own call pattern (cuda_mpm_test.cc:64-72: five solver calls per substep, GpuSync per frame of 40)
against mpm_run_substeps on the 1M workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
def engine():
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    return g
g = engine()
def frame_phase(n=40):
    for s in range(n):
        g.rebuild_mapping(False); g.calc_fem_state_and_force(1e-3); g.particle_to_grid(1e-3); g.update_grid(-1); g.grid_to_particle(1e-3)
    g.gpu_sync()
frame_phase(5)
for k in range(3):
    t0 = time.perf_counter(); frame_phase(); el = time.perf_counter() - t0
    print("five calls per substep: frame of 40: %.1f us/substep" % (el / 40 * 1e6), flush=True)
g.destroy()
g = engine()
g.run_substeps(5, 1e-3, -1); g.gpu_sync()
for k in range(3):
    t0 = time.perf_counter(); g.run_substeps(40, 1e-3, -1); g.gpu_sync(); el = time.perf_counter() - t0
    print("mpm_run_substeps(40): %.1f us/substep" % (el / 40 * 1e6), flush=True)
