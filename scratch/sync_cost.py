"""Host-side cost of a synchronising call on an idle stream (what the timed region of bench.py pays once at its end)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
g = GpuMpm(bits)
scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
g.run_substeps(5, 1e-3, -1); g.gpu_sync()
def t(f, n=20):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e6
print("gpu_sync on an idle stream: %.1f us" % t(g.gpu_sync))
def run1():
    g.run_substeps(1, 1e-3, -1); g.gpu_sync()
print("1 substep + gpu_sync: %.1f us" % t(run1))
def run2():
    g.run_substeps(2, 1e-3, -1); g.gpu_sync()
print("2 substeps + gpu_sync: %.1f us" % t(run2))
def run20():
    g.run_substeps(20, 1e-3, -1); g.gpu_sync()
print("20 substeps + gpu_sync: %.1f us" % t(run20, 5))
def run40():
    g.run_substeps(40, 1e-3, -1); g.gpu_sync()
print("40 substeps + gpu_sync: %.1f us" % t(run40, 3))
