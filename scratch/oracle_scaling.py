import sys, time, collections
sys.path.insert(0, '.')
from oracle import oracle as orc
from drake_amd import scenes
o = orc.OracleMpm(7); o.fast_scatter = True
for pos, vel, idx in scenes.cloth_stack(4, 145, 7): o.add_qr_cloth(pos, vel, idx)
o.finalize(); o.substep(1e-3, -1)
for th in (8, 16, 32, 64, 128):
    orc.set_threads(th)
    T = collections.defaultdict(float)
    o.substep(1e-3, -1)
    for _ in range(3):
        for name, fn in [('rebuild', lambda: o.rebuild_mapping(False)), ('fem', lambda: o.calc_fem_state_and_force(1e-3)), ('p2g', lambda: o.particle_to_grid(1e-3)), ('grid', lambda: o.update_grid(-1)), ('g2p', lambda: o.grid_to_particle(1e-3))]:
            t = time.perf_counter(); fn(); T[name] += (time.perf_counter() - t) / 3
    print(th, {k: round(v * 1e3, 2) for k, v in T.items()}, 'total ms', round(sum(T.values()) * 1e3, 2))
