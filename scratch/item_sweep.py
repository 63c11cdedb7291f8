"""Phase times of the 1M benchmark scene for different work-item sizes (MPM_ITEM_GROUPS) and P2G grid sizes."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from drake_amd import GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS[os.environ.get("SWEEP_CONFIG", "cloth_1m")]
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits))
    g.run_substeps(20, 1e-3, -1)
    ph, tot = g.profile_substeps(100, 1e-3, -1)
    st = g.stats()
    print(json.dumps(dict(ig=os.environ.get("MPM_ITEM_GROUPS"), wgs=os.environ.get("MPM_P2G_WGS"), home=st["home_blocks"],
                          p2g=round(ph["p2g"] * 1e3, 1), grid=round(ph["grid"] * 1e3, 1), g2p=round(ph["g2p"] * 1e3, 1),
                          total=round(tot * 1e3, 1))))
else:
    for ig in (48, 32, 24, 16, 12, 8):
        for wgs in (512,):
            env = dict(os.environ, MPM_ITEM_GROUPS=str(ig), MPM_P2G_WGS=str(wgs))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
