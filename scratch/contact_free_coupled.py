"""What a coupled substep costs while nothing touches the collider yet: the 1M cloth stack released 0.35 above the floor,
200 coupled substeps in one call, with the watch (contact-free substeps enqueued without pair generation) and without
(MPM_CT_NO_WATCH=1), against plain contact-free substeps (mpm_run_substeps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import Collider, GpuMpm, scenes
bits, layers, res = scenes.CONFIGS["cloth_1m"]
floor = (Collider * 1)(Collider(0, body=0, p_WB=(0.5, 0.5, 0.25)))
def engine():
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
    g.reallocate_external_bodies(1)
    return g
for mode in ("watch", "no watch", "plain"):
    if mode == "no watch": os.environ["MPM_CT_NO_WATCH"] = "1"
    else: os.environ.pop("MPM_CT_NO_WATCH", None)
    g = engine()
    run = (lambda n: g.run_substeps(n, 2e-4, -1)) if mode == "plain" else (lambda n: g.run_coupled_substeps(n, 2e-4, floor, 1.0, 1e6, 1e-5))
    run(20); g.gpu_sync()
    for k in range(3):
        t0 = time.perf_counter(); run(200); g.gpu_sync(); el = time.perf_counter() - t0
        print(f"{mode:9s}: {el / 200 * 1e3:.4f} ms per substep", g.contact_counters() if mode != "plain" else "", flush=True)
    assert g.stats()["error_flags"] == 0
    g.destroy()
