#!/usr/bin/env python3
"""BASELINE.json configs[2]: the 1M-particle cloth stack dropped on a rigid floor (MPM + contact solve).

Not a bench.py line (bench.py measures configs[1]); prints a JSON record with the time split of a
contact substep so the contact kernels can be profiled at scale:

  python scripts/bench_contact.py [--steps 30] [--config cloth_1m] [--mu 0.5]

Contact pairs are produced the way the reference's DeformableDriver does it (positions to the host,
signed distance per particle, pairs back to the device, deformable_driver.h:120-194); that host
round trip is timed separately from the solve.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cloth_1m")
    ap.add_argument("--mu", type=float, default=0.5)
    ap.add_argument("--floor", type=float, default=0.5)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--stiffness", type=float, default=1e5)
    ap.add_argument("--damping", type=float, default=1e-3)
    ap.add_argument("--survey-config3", action="store_true",
                    help="SURVEY.md 8(d) config 3: floor z<0.25, k=1e6, d=1e-5, mu=1, dt=2e-4")
    ap.add_argument("--device-pairs", action="store_true",
                    help="contact pairs from mpm_generate_contact_pairs instead of the host round trip")
    args = ap.parse_args()
    if args.survey_config3:
        args.floor, args.stiffness, args.damping, args.mu, args.dt = 0.25, 1e6, 1e-5, 1.0, 2e-4
    from drake_amd import Collider, GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS[args.config]
    dt, stiffness, damping = args.dt, args.stiffness, args.damping
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, args.floor))]
    g = GpuMpm(bits)
    # the stack starts with its lowest sheets already touching the floor and moves down at 0.5 m/s
    sheets = scenes.cloth_stack(layers, res, bits, z0=args.floor - 0.004)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
    scenes.populate(g, sheets)
    g.reallocate_external_bodies(1)
    T = dict(transfer=0.0, pairs_host=0.0, copy_pairs=0.0, solve=0.0, step=0.0)
    iters, ncontacts = [], []
    for s in range(args.warmup + args.steps):
        timed = s >= args.warmup
        t0 = time.perf_counter()
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(dt)
        g.particle_to_grid(dt)
        g.update_grid(-1)
        g.gpu_sync()
        t1 = time.perf_counter()
        if args.device_pairs:
            t2 = t3 = time.perf_counter()
            n = g.generate_contact_pairs(floor)
            t4 = time.perf_counter()
        else:
            pos = g.sync_particle_state_to_cpu()
            t2 = time.perf_counter()
            z = pos[:, 2]
            idx = np.nonzero(z < args.floor)[0].astype(np.uint32)
            n = idx.size
            dist = (z[idx] - args.floor).astype(np.float32)
            normal = np.tile(np.array([0, 0, -1], np.float32), (n, 1))
            cpos = pos[idx]
            zeros = np.zeros((n, 3), np.float32)
            body = np.zeros(n, np.uint32)
            t3 = time.perf_counter()
            g.copy_contact_pairs(idx, body, dist, normal, cpos, zeros, zeros)
            t4 = time.perf_counter()
        r = g.update_contact(dt, args.mu, stiffness, damping)
        g.gpu_sync()
        t5 = time.perf_counter()
        g.grid_to_particle(dt)
        g.gpu_sync()
        t6 = time.perf_counter()
        if timed:
            T["step"] += (t1 - t0) + (t6 - t5)
            T["transfer"] += t2 - t1
            T["pairs_host"] += t3 - t2
            T["copy_pairs"] += t4 - t3
            T["solve"] += t5 - t4
            iters.append(r["iterations"])
            ncontacts.append(n)
    k = args.steps
    out = dict(config=args.config, particles=g.n_particles, steps=k, mu=args.mu, dt=dt, stiffness=stiffness,
               damping=damping, floor=args.floor, pairs="device" if args.device_pairs else "host",
               contacts_mean=float(np.mean(ncontacts)), contacts_max=int(np.max(ncontacts)),
               newton_iterations_mean=float(np.mean(iters)), newton_iterations_max=int(np.max(iters)),
               ms_per_substep={a: 1e3 * b / k for a, b in T.items()},
               solve_us_per_iteration=1e6 * T["solve"] / max(1, int(np.sum(iters))), stats=g.stats())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
