"""Same-box A/B of engine variants built by scratch/ab_build.py: phase times of the 1M benchmark scene.
    python scratch/ab_run.py name1 name2 ...      (each in its own process; MPM_AB_ROUNDS repeats, interleaved)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import time
    from drake_amd import GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS[os.environ.get("AB_CONFIG", "cloth_1m")]
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits))
    g.run_substeps(5, 1e-3, -1)
    g.gpu_sync()
    t0 = time.perf_counter()
    g.run_substeps(20, 1e-3, -1)
    g.gpu_sync()
    w20 = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    g.run_substeps(200, 1e-3, -1)
    g.gpu_sync()
    w200 = (time.perf_counter() - t0) / 200
    ph, tot = g.profile_substeps(100, 1e-3, -1)
    st = g.stats()
    print(json.dumps(dict(v=sys.argv[2], run20_us=round(w20 * 1e6, 1), run200_us=round(w200 * 1e6, 1),
                          **{k: round(v * 1e3, 1) for k, v in ph.items()}, prof_total=round(tot * 1e3, 1),
                          rebuilds=st["rebuilds"], err=st["error_flags"])), flush=True)
else:
    for r in range(int(os.environ.get("MPM_AB_ROUNDS", "2"))):
        for name in sys.argv[1:]:
            lib = os.environ.get("MPM_HIP_LIBRARY") if name == "cur" else os.path.join(ROOT, "drake_amd", "variants", f"libmpm_hip_{name}.so")
            env = dict(os.environ, MPM_HIP_LIBRARY=lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=env, timeout=300)
