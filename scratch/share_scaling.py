"""What the substep kernels of ONE rank cost as its share of the 1M workload shrinks (DESIGN_HISTORY.md section 5.6: does P2G get
faster on a smaller share?).  The middle rank of `world` x-slabs (dist.strong_geometry), no exchange, event times of
mpm_profile_substeps (raw intervals, the empty vertex-force slot = cost of an event pair).  MPM_ITEM_GROUPS varies the
size of the work items (read when the engine is created)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drake_amd import GpuMpm, scenes
from drake_amd.dist import strong_geometry
cfg = sys.argv[1] if len(sys.argv) > 1 else "cloth_1m"
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
bits, layers, res = scenes.CONFIGS[cfg]
for world in (1, 2, 4, 8):
    for ig in (48, 16, 8):
        os.environ["MPM_ITEM_GROUPS"] = str(ig)
        g = GpuMpm(bits)
        scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=1234))
        if world > 1:
            geo = strong_geometry(bits, world)
            g.dist_init(world // 2, world, geo["cuts"], geo["zone_blocks"], 0, 0)
        g.profile_substeps(5, dt, -1)
        ph, tot = g.profile_substeps(20, dt, -1)
        st = g.stats()
        ev = ph["vforce"]
        print(f"{cfg} world {world} items<={ig:2d} groups: held {st['active_faces'] + st['active_vertices']:8d} particles, "
              f"home blocks {st['home_blocks']:5d} | net us: fem {1e3*(ph['fem']-ev):5.1f} p2g {1e3*(ph['p2g']-ev):5.1f} "
              f"grid {1e3*(ph['grid']-ev):5.1f} g2p {1e3*(ph['g2p']-ev):5.1f} resort-check {1e3*(ph['rebuild']-ev):5.1f} | "
              f"event pair {1e3*ev:4.1f}  err {st['error_flags']}", flush=True)
        g.destroy()
