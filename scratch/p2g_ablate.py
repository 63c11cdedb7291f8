"""P2G ablations of the diagnostic build (variant library built with -DMPM_DIAG=1): event time of each phase with
parts of k_p2g switched off by MPM_DBG bits (results are wrong by construction; timing only).
  2: stop after the per-particle derivation   1: stop after grouping + staging   8: no MFMA steps (cells and
  epilogues still run)   16: no LDS atomics   64: one FMA instead of each MFMA   128: no weight polynomials"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from drake_amd import GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits))
    g.run_substeps(20, 1e-3, -1)
    g.gpu_sync()
    best = None
    for _ in range(3):
        ph, tot = g.profile_substeps(40, 1e-3, -1)
        if best is None or ph["p2g"] < best["p2g"]:
            best = dict(ph)
    print(json.dumps({"dbg": int(os.environ.get("MPM_DBG", "0")), **{k: round(v * 1e3, 1) for k, v in best.items()}}))
    sys.exit(0)
for flags in (0, 128, 16, 2, 1, 8, 0):
    env = dict(os.environ, MPM_DBG=str(flags), MPM_HIP_LIBRARY=os.path.join(ROOT, "drake_amd/variants/libmpm_hip_diag.so"))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + r.stderr[-300:]), flush=True)
