#!/usr/bin/env python3
"""Benchmark of the MPM substep hot path (BASELINE.json metric).

One "step" = one contact-free MPM substep (RebuildMapping, CalcFemStateAndForce,
ParticleToGrid, UpdateGrid, GridToParticle) of the 1M-particle cloth stack on a
128^3 grid (BASELINE.json configs[1]), state resident in HBM.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the definition
of the roofline and cpu_baseline objects.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: RCCL / cross-process device memory needs it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def kernel_names(fast_math, partitioned):
    """kernel names as rocprofv3 prints them, per phase of mpm_profile_substeps: the instantiations THIS run launches
    (the table used to hard-code the single-engine, non-deterministic ones).  k_fem<FM>: 0 correctly rounded
    (default), 1 mpm_set_fast_math; k_p2g<FORCES, EXACT>: the vertex forces from the vertex-side records (1: single and
    partitioned domains alike since round 5); EXACT = 1 with the fixed-point tile (MPM_DETERMINISTIC / MPM_P2G_FIXED);
    the grid update of a partitioned rank is two kernels (k_grid<0> before the exchange, k_grid<2> after it)."""
    exact = 1 if (os.environ.get("MPM_DETERMINISTIC") is not None or os.environ.get("MPM_P2G_FIXED") is not None) else 0
    return dict(fem=f"mpm::k_fem<{1 if fast_math else 0}>", vforce="mpm::k_vforce", p2g=f"mpm::k_p2g<1, {exact}>",
                grid="mpm::k_grid<2>" if partitioned else "mpm::k_grid<1>", g2p="mpm::k_g2p")


def algorithmic_bytes(np_, nf, nv, ncells):
    """bytes one substep has to move, fp32, one pass per phase.
    The reference's FEM kernel (200 B/face + 36 B/vertex) is two kernels here (k_fem, and k_vforce or -- in
    mpm_run_substeps -- the vertex lanes of k_p2g, which sum the triples k_fem left them)."""
    fem = 200 * nf
    vforce = 36 * nv
    p2g = 116 * np_ + 16 * ncells
    grid = 40 * ncells
    g2p = 72 * np_ + 12 * ncells
    return dict(fem=fem, vforce=vforce, p2g=p2g, grid=grid, g2p=g2p, total=fem + vforce + p2g + grid + g2p)


def measured_sq(kernel, config):
    """SQ counters per launch of `kernel` from the committed rocprofv3 PMC pass of this same command
    (profiles/sq_counters.json, written by scripts/sq_summary.py), or None."""
    path = os.path.join(ROOT, "profiles", "sq_counters.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("config") != config:
            return None
        for rec in t["kernels"]:
            if rec["kernel"] == kernel:
                return rec["counters"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def measured_traffic(kernel, config):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (profiles/pmc_traffic.json, written by scripts/pmc_summary.py: FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE, KiB -> bytes), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("config") != config:
            return None
        ks = t["kernels"]
        # (k_p2g became a template in round 4: profiles collected before that carry the plain name)
        return float((ks[kernel] if kernel in ks else ks[kernel.split("<")[0]])["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def measured_kernel_ns(kernel, config):
    """Average duration (ns) of `kernel` in the committed rocprofv3 --kernel-trace --stats run of this same command
    (profiles/rNN_kernel_stats.csv, carried into profiles/pmc_traffic.json by scripts/pmc_summary.py), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("config") != config:
            return None
        ks = t["kernels"]
        v = (ks[kernel] if kernel in ks else ks[kernel.split("<")[0]]).get("avg_duration_ns")
        return float(v) if v else None
    except (OSError, KeyError, ValueError, TypeError):
        return None


def host_description():
    """CPU model and the cores this process may use (north_star: "core count stated")."""
    model, sockets, cores_per_socket, threads_per_core = "unknown", None, None, None
    try:
        import subprocess
        for ln in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                model = v
            elif k == "Socket(s)":
                sockets = int(v)
            elif k == "Core(s) per socket":
                cores_per_socket = int(v)
            elif k == "Thread(s) per core":
                threads_per_core = int(v)
    except Exception:  # noqa: BLE001
        pass
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    # a container may also be limited by a CPU quota rather than by affinity
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:  # noqa: BLE001
        pass
    tpc = threads_per_core or 1
    phys = max(1, allowed // tpc)
    # what this process can actually run on: the affinity mask capped by the container's CPU quota
    avail = phys if quota is None else max(1, min(phys, int(round(quota))))
    return dict(model=model, logical_cpus_of_host=os.cpu_count(), logical_cpus_allowed=allowed,
                physical_cores_of_host=phys, threads_per_core=tpc, sockets=sockets, cores_per_socket=cores_per_socket,
                cgroup_cpu_quota=quota, cores_available=avail)


HOST = None   # host_description(), taken in main() before anything touches the GPU (it forks lscpu)


def cpu_baseline(domain_bits, layers, res, dt, budget_s=20.0):
    """The oracle (a port of the reference's kernels to plain C + OpenMP; the reference has no CPU
    path of its own) timed on the host cores on the FULL workload (same scene, grid and dt as the
    GPU leg), for a bounded number of substeps.  Reported: the rate at the thread count that is
    fastest on this box (`cores` = that thread count), the 1-thread rate, and the host's CPU."""
    from drake_amd import scenes
    from oracle import oracle as orc
    host = HOST or host_description()
    o = orc.OracleMpm(domain_bits)
    o.fast_scatter = True  # atomics-free multi-core scatter (same arithmetic, see oracle/mpm_oracle.c)
    for pos, vel, idx in scenes.cloth_stack(layers, res, domain_bits):
        o.add_qr_cloth(pos, vel, idx)
    o.finalize()
    t_begin = time.perf_counter()
    # thread counts: the cores this process is allowed to use, and fractions / multiples of that
    # (the port has short loops and dense-grid sweeps: it does not scale to every core of a big host)
    avail = host["cores_available"]
    cand = sorted({max(1, min(orc.max_threads(), c)) for c in (avail // 2, avail, 2 * avail)})
    best, threads = None, 1
    orc.set_threads(cand[-1])
    o.substep(dt, -1)  # warm-up (page in the dense grid arrays)
    for c in cand:
        orc.set_threads(c)
        t0 = time.perf_counter()
        o.substep(dt, -1)
        o.substep(dt, -1)
        el = (time.perf_counter() - t0) / 2
        if best is None or el < best:
            best, threads = el, c
    orc.set_threads(threads)
    n, t0 = 0, time.perf_counter()
    while True:
        o.substep(dt, -1)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.6 or n >= 50:
            break
    sps = n / el
    # the scalar figure: a few substeps on one thread, plain (non-coloured) scatter
    orc.set_threads(1)
    o.fast_scatter = False
    n1, t0 = 0, time.perf_counter()
    while True:
        o.substep(dt, -1)
        n1 += 1
        el1 = time.perf_counter() - t0
        if el1 > budget_s * 0.25 or n1 >= 5:
            break
    return dict(value=sps, unit="substeps/s", cores=threads, kind="port",
                sample=f"{n} substeps of the full workload ({o.n_particles} particles, {1 << domain_bits}^3 grid, "
                       f"dt={dt}); OpenMP oracle with {threads} threads (fastest of {cand}; {host['cores_available']} cores "
                       f"available to this process on a {host['model']}); "
                       f"1 thread: {n1} substeps",
                one_thread_value=n1 / el1, threads_tried=cand, host=host,
                wall_s=time.perf_counter() - t_begin)


def contact_leg(device, steps=20, warmup=5):
    """BASELINE.json configs[2] / SURVEY.md 8(d) config 3, outside the headline timed region: the same
    1M-particle cloth stack pressed on a rigid floor (half-space z < 0.25), contact pairs made on the
    device, UpdateContact with the bagging demo's parameters (k = 1e6, d = 1e-5, mu = 1, dt = 2e-4,
    examples/multibody/deformable/mpm_bagging.cc:9,15-17).  One coupled substep = RebuildMapping ...
    UpdateGrid, pairs, UpdateContact, GridToParticle (deformable_driver.h:244-258).
    `ms_per_substep` goes through mpm_run_coupled_substeps (the loop body of deformable_driver.h:240-258 n times in one
    call -- an entry point the reference does not have, like mpm_run_substeps for the headline); `reference_call_pattern`
    is the same schedule through the reference's seven calls per substep on a second engine."""
    from drake_amd import Collider, GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    floor_z, k, d, mu, dt = 0.25, 1e6, 1e-5, 1.0, 2e-4
    floor = (Collider * 1)(Collider(0, body=0, p_WB=(0.5, 0.5, floor_z)))   # (the ctypes array built once, as a C++ caller would)

    def engine():
        g = GpuMpm(bits, device=device)
        sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z - 0.004)
        for pos, vel, idx in sheets:
            vel[:, 2] -= 0.5
        scenes.populate(g, sheets)
        g.reallocate_external_bodies(1)
        return g

    def batched(g, first, count):
        """`count` coupled substeps in two calls; the first `first` of them are not timed"""
        if first:
            g.run_coupled_substeps(first, dt, floor, mu, k, d)
        g.gpu_sync()
        t0 = time.perf_counter()
        rs = g.run_coupled_substeps(count - first, dt, floor, mu, k, d)
        g.gpu_sync()
        return time.perf_counter() - t0, [r["iterations"] for r in rs], [r["contacts"] for r in rs], sum(r["setup_reused"] for r in rs)

    def seven_calls(g, first, count):
        """the reference's calls in the reference's order (deformable_driver.h:244-258); the pair count stays on the device
        (want_count = False) and comes back with UpdateContact's result"""
        iters, contacts, reused, t0 = [], [], 0, 0.0
        for s in range(count):
            if s == first:
                g.gpu_sync()
                t0 = time.perf_counter()
            g.rebuild_mapping(False)
            g.calc_fem_state_and_force(dt)
            g.particle_to_grid(dt)
            g.update_grid(-1)
            g.generate_contact_pairs(floor, want_count=False)
            r = g.update_contact(dt, mu, k, d)
            g.grid_to_particle(dt)
            if s >= first:
                iters.append(r["iterations"])
                contacts.append(r["contacts"])
                reused += int(r["setup_reused"])
        g.gpu_sync()
        return time.perf_counter() - t0, iters, contacts, reused

    def schedule(g, run):
        # the impact: substeps `warmup` .. `warmup + steps` after the release (the number this leg reports) ...
        el, iters, contacts, reused = run(g, warmup, warmup + steps)
        # ... and the same stack once it has settled on the floor (fewer Newton iterations per solve)
        el_s, iters_s, contacts_s, reused_s = run(g, 100, 100 + steps)
        return dict(ms_per_substep=el / steps * 1e3, substeps_per_s=steps / el, contacts=float(np.mean(contacts)),
                    newton_iterations=float(np.mean(iters)), solves_on_a_reused_setup=reused,
                    settled=dict(ms_per_substep=el_s / steps * 1e3, contacts=float(np.mean(contacts_s)),
                                 newton_iterations=float(np.mean(iters_s)), after_substeps=warmup + steps + 100,
                                 solves_on_a_reused_setup=reused_s))

    g2 = engine()
    ref_calls = schedule(g2, seven_calls)
    ref_calls["calls"] = ("RebuildMapping(false), CalcFemStateAndForce, ParticleToGrid, UpdateGrid, pairs on the device, "
                          "UpdateContact, GridToParticle per substep (deformable_driver.h:244-258)")
    assert g2.stats()["error_flags"] == 0
    g2.destroy()
    g = engine()
    main = schedule(g, batched)
    settled = main["settled"]
    # roofline of the Newton iteration (~ 2 * 76 B per contact + 300 B per cell carrying contact
    # Hessians per iteration), on the settled stack: one more coupled substep up to the solve, then the four kernels of
    # an iteration re-launched back to back on that state and timed with HIP events (mpm_profile_contact_iteration)
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(dt)
    g.particle_to_grid(dt)
    g.update_grid(-1)
    nk = g.generate_contact_pairs(floor)
    g.update_contact(dt, mu, k, d)
    ncc = g.contact_stats()["nodes"]
    kms = g.profile_contact_iteration(20)
    g.grid_to_particle(dt)
    it_bytes = 2 * 76 * nk + 300 * ncc
    it_ms = sum(kms.values())
    dom = max(kms, key=kms.get)
    ach = it_bytes / (it_ms * 1e-3) / 1e9
    roofline = dict(bound="hbm", kernel="mpm::" + dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                    traffic=None, algorithmic_bytes_per_iteration=it_bytes, contacts=nk, cells_with_contacts=ncc,
                    kernel_ms=kms, iteration_ms=it_ms,
                    note="achieved = bytes of one Newton iteration / duration of its four kernels (latency-bound: "
                         "four dependent launches of 8-16 us move ~8 MB)")
    st = g.stats()
    assert st["error_flags"] == 0, st
    g.destroy()
    # (the iteration counts of a window differ from run to run with the particle order -- 12 .. 15 at the impact --, and
    # an iteration is it_ms: what a coupled substep costs BEYOND its iterations is the figure that compares between runs)
    for leg in (main, main["settled"], ref_calls, ref_calls["settled"]):
        leg["ms_beyond_iterations"] = leg["ms_per_substep"] - leg["newton_iterations"] * it_ms
    return dict(ms_per_substep=main["ms_per_substep"], substeps_per_s=main["substeps_per_s"], contacts=main["contacts"],
                newton_iterations=main["newton_iterations"], ms_beyond_iterations=main["ms_beyond_iterations"], steps=steps, warmup=warmup,
                solves_on_a_reused_setup=main["solves_on_a_reused_setup"],
                settled=settled, reference_call_pattern=ref_calls, roofline=roofline,
                params=dict(stiffness=k, damping=d, friction_mu=mu, dt=dt, floor_z=floor_z, line_search="backtracking"),
                workload="cloth_1m on a half-space, pairs from mpm_generate_contact_pairs (device)")


def team_contact_leg(rank, world, local_rank, config, steps=10, warmup=3):
    """BASELINE.json configs[4] (config 5: a cloth and a 16-link manipulator, two-way coupled, on several GPUs) as far as
    the hot path goes: the cloth stack of `config` pressed on a floor (body 0) under 16 capsules (bodies 1..16) that move
    with prescribed velocities, partitioned like the headline run, coupled substeps through DomainChain.coupled_substeps:
    the per-substep halo over the DIRECT transport and the contact solve's exchanges over the TEAM transport
    (drake_amd/csrc/mpm_team.h) -- peer stores + sequence flags on the engines' streams, the host polls the mailbox only --,
    migrations between the batches.  Every rank calls this.  UNMEASURED ON HARDWARE with world > 1: on this pool the ranks
    share one GPU (a rehearsal of the protocol: rc 0, contacts, iterations), the milliseconds mean nothing."""
    import torch
    import torch.distributed as dist
    from drake_amd import Collider, GpuMpm, scenes
    from drake_amd.dist import DomainChain, strong_geometry
    bits, layers, res = scenes.CONFIGS[config]
    floor_z, k, d, mu, dt = 0.25, 1e6, 1e-5, 1.0, 2e-4
    geo = strong_geometry(bits, world)
    # (every wait of the peer-store transports is bounded; this leg has never met two devices, so the bound is short here: a
    # flag that does not become visible costs seconds, not minutes, and the leg reports the error instead of a number)
    os.environ.setdefault("MPM_HALO_TIMEOUT_S", "1.0")
    g = GpuMpm(bits, device=local_rank)
    sheets = scenes.cloth_stack(layers, res, bits, z0=floor_z - 0.004)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
    scenes.populate(g, sheets)
    g.reallocate_external_bodies(17)
    chain = DomainChain(g, rank, world, geo["cuts"], geo["zone_blocks"], geo["ghost_cells"], geo["ghost_margin_cells"], 1024,
                        geo["migrate_every"], 1 << 18, device=torch.device("cuda", local_rank))
    ok = chain.enable_team(1024)
    if not ok:
        return dict(error="team transport not available: " + chain.team_error)
    top = floor_z - 0.004 + layers * 0.5 / (1 << bits)
    Ry = (0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0)    # the capsules' axes along y

    def colliders(t):
        cols = [Collider(0, body=0, p_WB=(0.5, 0.5, floor_z))]
        for j in range(16):   # a row of links along x, alternately moving left and right: some straddle every cut
            vx = 0.5 if j % 2 == 0 else -0.5
            cols.append(Collider(3, body=1 + j, p_WB=(0.27 + 0.03 * j + vx * t, 0.5, top + 0.006), R_WB=Ry, dims=(0.012, 0.2, 0.0),
                                 v=(vx, 0.0, 0.0)))
        return cols

    err = None
    try:
        chain.coupled_substeps(warmup, dt, colliders(0.0), mu, k, d)
        g.gpu_sync()
    except Exception as exc:  # noqa: BLE001  (e.g. MPM_ERR_HALO: a peer's flag never arrived)
        err = repr(exc)
    bad = torch.tensor([1.0 if err else 0.0])
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if float(bad.item()) > 0:     # all ranks leave together
        try:
            g.chain_destroy()
            g.destroy()
        except Exception:  # noqa: BLE001
            pass
        return dict(error="the warm-up of the team contact leg failed on at least one rank" + (": " + err if err else ""))
    dist.barrier()
    t0 = time.perf_counter()
    rs = []
    try:
        rs = chain.coupled_substeps(steps, dt, colliders(warmup * dt), mu, k, d)
        g.gpu_sync()
    except Exception as exc:  # noqa: BLE001  (every wait inside is bounded: a rank that fails alone gets here, and so do the others)
        err = repr(exc)
    el = time.perf_counter() - t0
    # (no collective between the start of the timed calls and this one: whatever a rank met, it arrives here, and the
    # reduction is the closing barrier of the timed region)
    bad = torch.tensor([1.0 if err else 0.0])
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if float(bad.item()) > 0:
        try:
            g.chain_destroy()
            g.destroy()
        except Exception:  # noqa: BLE001
            pass
        return dict(error="the timed coupled substeps of the team contact leg failed on at least one rank" + (": " + err if err else ""))
    t = torch.tensor([el, float(sum(r["contacts"] for r in rs))], dtype=torch.float64)
    dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
    dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
    tau, f = g.external_body_force_to_host()
    fs = torch.from_numpy(np.ascontiguousarray(f, dtype=np.float64))
    dist.all_reduce(fs, op=dist.ReduceOp.SUM)
    st = g.stats()
    assert st["error_flags"] == 0, st
    out = dict(ms_per_substep=float(t[0]) / steps * 1e3, contacts=float(t[1]) / steps, newton_iterations=float(np.mean([r["iterations"] for r in rs])),
               steps=steps, warmup=warmup, bodies=17, migrations=chain.migrations,
               bodies_with_an_impulse=int(np.count_nonzero(np.abs(fs.numpy()).max(axis=1) > 0)),
               transport="halo: peer stores + flags (mpm_chain_direct); solve: zone exchange + rank-ordered sums as peer stores + flags "
                         "(mpm_team); host: mailbox polling only",
               params=dict(stiffness=k, damping=d, friction_mu=mu, dt=dt, floor_z=floor_z, line_search="backtracking"),
               workload=f"{config} on a half-space under 16 moving capsules, partitioned into {world} x slabs, pairs made on the device",
               note="unmeasured on hardware with world > 1: on this pool the ranks share one GPU (a rehearsal)")
    g.chain_destroy()
    g.destroy()
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without torchrun: start N rank processes of this same script
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) and
    exit with the worst of their codes.  The parent never initialises the GPU and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # only rank 0 prints the JSON line; the other ranks' stdout goes to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        try:
            code = p.wait(timeout=3000)
        except subprocess.TimeoutExpired:
            p.kill()
            code = 124
        rc = max(rc, abs(code))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 substeps ...
    ap.add_argument("--warmup", type=int, default=20)  # ... after 20 warm-up
    ap.add_argument("--config", default="cloth_1m")
    ap.add_argument("--dt", type=float, default=1e-3, help="substep length (SURVEY 8d: 1e-3 for config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sort-every", type=int, default=0,
                    help="also call RebuildMapping(sort=true) every N substeps (SURVEY 8d config 2 variants; "
                         "0 = never, the reference's Drake behaviour and the reported metric)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default, BASELINE.json: substeps/s at 1M particles on 1/2/4/8 GPUs): the ranks share "
                         "ONE copy of the workload; weak: every rank owns its own copy")
    ap.add_argument("--no-contact-leg", action="store_true", help="skip the config-3 (contact) record")
    ap.add_argument("--contact-only", action="store_true",
                    help="only the config-3 (contact) record: the command whose rocprofv3 kernel trace is profiles/rNN_contact_*")
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.launcher_selftest and "WORLD_SIZE" in os.environ:
        # (tests/test_bench_helpers.py: what a rank process sees, without touching torch or the GPU)
        if os.environ.get("RANK") == "0":
            print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                             | {"n_gpus": args.gpus}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    global HOST
    HOST = host_description()   # (forks lscpu: before the first GPU call of this process)
    if args.contact_only:
        print(json.dumps(dict(contact=contact_leg(int(os.environ.get("LOCAL_RANK", "0")), steps=args.steps if args.steps != 200 else 20,
                                                  warmup=args.warmup if args.warmup != 20 else 5))), flush=True)
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    # Multi-GPU transports, tried in this order (the JSON line names the one that was measured):
    #   0. only with MPM_BENCH_TRANSPORT=direct, on top of 1: peer stores + flags for the per-substep halo (mpm_chain_direct_*),
    #   1. the library's own chain: RCCL send/recv on the engine's stream (mpm_chain_*),
    #   2. torch.distributed point-to-point over RCCL ("nccl" backend, drake_amd/dist.py),
    #   3. host-staged gloo (what the one-GPU tests exercise).
    # The process group itself is gloo: it only carries the rendezvous, barriers and the timing
    # reduction.  MPM_BENCH_BACKEND=gloo forces 3 (ranks may then share one GPU: a rehearsal).
    backend = os.environ.get("MPM_BENCH_BACKEND", "rccl")
    ndev = torch.cuda.device_count()
    if world > 1 and (backend == "gloo" or os.environ.get("MPM_BENCH_SHARE_GPU")):
        local_rank = local_rank % ndev   # rehearsal on a box with fewer GPUs than ranks
    elif local_rank >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} are visible "
                         "(set MPM_BENCH_SHARE_GPU=1 to rehearse several ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    saved_stdout = None
    if world > 1:
        # (a failed communicator set-up then says why on stderr, in RCCL's own words)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        # gloo and RCCL print banners on stdout; the contract is ONE JSON line there: everything else goes
        # to stderr until the result is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        # (a collective that a rank never joins -- it failed alone in the optional team leg -- gives up after ten minutes
        # instead of gloo's thirty: the headline line is computed before that leg and still comes out)
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))

    from drake_amd import GpuMpm, scenes
    from drake_amd.dist import DomainChain, HaloChain, strong_geometry
    bits, layers, res = scenes.CONFIGS[args.config]
    dt = args.dt
    nb = (1 << bits) // 4
    stream = torch.cuda.Stream()
    strong = world > 1 and args.scaling == "strong"

    def all_ok(flag):   # every rank takes the same decision
        if world == 1:
            return flag
        t = torch.tensor([1.0 if flag else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    def make_engine(seed):
        g = GpuMpm(bits, device=local_rank)
        scenes.populate(g, scenes.cloth_stack(layers, res, bits, seed=seed))
        return g

    # ---- what every rank holds, and how the ranks are coupled -------------------------------------
    if strong:
        # ONE copy of the workload, cut into x slabs of equal width across the cloth (blocks nb/4 .. 3nb/4);
        # every rank finalises the whole scene and keeps its slab (mpm_dist_init)
        # (drake_amd/dist.py: strong_geometry -- the same function the tests of the 4- and 8-rank partitions use: zone 2
        # blocks where a slab is at least 4 blocks wide, else 1; ghost bands from the mesh; adaptive migration cadence)
        geometry = strong_geometry(bits, world)
        cuts, zone = geometry["cuts"], geometry["zone_blocks"]
        ghost, margin, mig_every = geometry["ghost_cells"], geometry["ghost_margin_cells"], geometry["migrate_every"]
        g = make_engine(1234)
        # The native chain sends a fixed capacity per substep (1 KiB per block; a RCCL send needs its size when it is
        # enqueued), so the capacity is what the ranks' zones hold right after the partition, agreed between them,
        # with a factor 2 of headroom for the blocks the cloth reaches later (an overflow is MPM_ERR_CAPACITY, checked
        # below) -- not a guess made before the block tables exist.  Filled in after mpm_dist_init, below.
        chain_args = dict(cut_lo_block=cuts[rank], cut_hi_block=cuts[rank + 1], pitch_blocks=0, zone_blocks=zone,
                          capacity_blocks=None)
        mig_cap = 65536
    else:
        # Weak scaling: every rank owns one copy of the workload.  The ranks' patches sit side by side
        # along x (rank r's local frame is shifted by r * 0.5), so neighbouring stacks share grid nodes
        # around the cut and exchange them every substep (drake_amd/dist.py).
        g = make_engine(1234 + rank)
        chain_args = dict(cut_lo_block=nb // 4, cut_hi_block=3 * nb // 4, pitch_blocks=nb // 2, zone_blocks=2,
                          capacity_blocks=1024)
        geometry = dict(patches="one per rank, pitch 0.5 along x")
    nv, nf, npart = g.n_verts, g.n_faces, g.n_particles
    chain = None
    native = False
    transport = "none"

    if world > 1:
        if strong:
            g.dist_init(rank, world, cuts, zone, ghost, margin)
            mine = max(g.halo_zone_blocks(cuts[rank] - zone, cuts[rank] + zone - 1) if rank > 0 else 0,
                       g.halo_zone_blocks(cuts[rank + 1] - zone, cuts[rank + 1] + zone - 1) if rank < world - 1 else 0)
            t = torch.tensor([float(mine)])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cap_blocks = 64
            while cap_blocks < 2.0 * float(t.item()):
                cap_blocks *= 2
            chain_args["capacity_blocks"] = cap_blocks
            geometry["exchange_capacity_blocks"] = cap_blocks
            geometry["zone_blocks_now"] = int(t.item())
            geometry.update({k: round(v, 4) for k, v in g.dist_geometry().items() if k.endswith("_cells")})
            geometry["migrate_every"] = "adaptive (half of the ranks' common quiet-time estimate)" if mig_every == 0 else mig_every
            # per-rank topology: what rank 0 allocates after the partition, against the whole scene
            s0 = g.stats()
            geometry["rank0_allocation"] = dict(
                face_slots=s0["face_slots"], vertex_slots=s0["vertex_slots"], held_faces=s0["active_faces"],
                held_vertices=s0["active_vertices"], scene_faces=nf, scene_vertices=nv,
                particle_array_MB=round(s0["particle_bytes"] / 1e6, 1), scene_index_MB=round(s0["scene_index_bytes"] / 1e6, 1))
        if backend != "gloo":
            # 1. native chain
            box = [None]
            try:
                if rank == 0:
                    box[0] = GpuMpm.chain_unique_id()
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] rank 0: no RCCL id ({exc!r})", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            ok = box[0] is not None
            if ok:
                try:
                    g.chain_init(box[0], rank, world, **chain_args)
                    if strong:
                        g.chain_enable_migration(mig_every, mig_cap)
                    g.chain_substeps(1, dt, -1)
                    g.gpu_sync()
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] rank {rank}: native RCCL chain failed: {exc}", file=sys.stderr, flush=True)
                    ok = False
            native = all_ok(ok)
            if native:
                transport = "RCCL send/recv on the engine stream (mpm_chain)"
            else:
                g.chain_destroy()
            if os.environ.get("MPM_BENCH_TRANSPORT") == "direct" and (native or not strong):
                # 0. (opt-in: never run across two devices) the per-substep halo as peer stores into the
                # neighbours' IPC-mapped buffers + flags.  On top of the RCCL chain, which keeps the migrations; without
                # RCCL (ranks sharing a GPU: a rehearsal) only where nothing migrates (weak scaling).  All ranks or none.
                ok = True
                try:
                    if not native:
                        g.chain_init(None, rank, world, **chain_args)
                    handles = [None] * world
                    dist.all_gather_object(handles, g.chain_direct_prepare())
                    g.chain_direct_connect(handles[rank - 1] if rank > 0 else None, handles[rank + 1] if rank < world - 1 else None)
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] rank {rank}: direct halo exchange not available: {exc}", file=sys.stderr, flush=True)
                    ok = False
                if all_ok(ok):
                    g.chain_substeps(2, dt, -1)   # (both parities; a neighbour that never signals is MPM_ERR_HALO after 5 s)
                    g.gpu_sync()
                    transport = ("peer stores into IPC-mapped buffers + flags (mpm_chain_direct)" +
                                 ("; migrations over RCCL" if native else ", no RCCL"))
                    native = True
                elif native:
                    try:
                        g.chain_direct_connect(None, None)   # (off on every rank: one transport for all)
                    except Exception:  # noqa: BLE001
                        pass
                else:
                    g.chain_destroy()

        def python_chain(group):
            if strong:
                # (the engine is partitioned already: no second mpm_dist_init)
                return DomainChain(g, rank, world, cuts, zone, ghost, margin, chain_args["capacity_blocks"], mig_every,
                                   mig_cap, device=torch.device("cuda", local_rank), group=group, partitioned=True)
            return HaloChain(g, rank, world, device=torch.device("cuda", local_rank), group=group, **chain_args)

        if not native and backend != "gloo":
            # 2. torch.distributed over RCCL
            ok = True
            try:
                g.set_stream(stream.cuda_stream)  # kernels and RCCL transfers ordered through one stream
                chain = python_chain(dist.new_group(backend="nccl"))
                with torch.cuda.stream(stream):
                    chain.run_substeps(1, dt, -1)
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] rank {rank}: torch RCCL halo exchange failed: {exc}", file=sys.stderr, flush=True)
                ok = False
            if all_ok(ok):
                transport = "RCCL via torch.distributed point-to-point"
            else:
                chain = None
        if not native and chain is None:
            # 3. host-staged
            g.set_stream(stream.cuda_stream)
            chain = python_chain(None)
            transport = "host-staged (gloo)"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n):
        if native:
            g.chain_substeps(n, dt, -1)
        elif chain is None and args.sort_every > 0:
            done = 0
            while done < n:
                g.rebuild_mapping(True)   # slot-order sort (device radix sort); particle data does not move
                k = min(args.sort_every, n - done)
                g.run_substeps(k, dt, -1)
                done += k
        elif chain is None:
            g.run_substeps(n, dt, -1)
        else:
            with torch.cuda.stream(stream):
                chain.run_substeps(n, dt, -1)

    run(args.warmup)
    g.gpu_sync()
    rebuilds_before = g.stats()["rebuilds"]   # (a synchronising call: the window starts from an idle, settled engine)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    t_enq = time.perf_counter()
    g.gpu_sync()
    t_sync = time.perf_counter()
    barrier()
    el = time.perf_counter() - t0
    # (where the wall time of the timed region went: enqueueing, the engine's own completion, the contract's barrier)
    region = dict(enqueue_ms=(t_enq - t0) * 1e3, engine_complete_ms=(t_sync - t0) * 1e3, with_barrier_ms=el * 1e3)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    st = g.stats()
    assert st["error_flags"] == 0, st
    region["resorts_in_timed_window"] = st["rebuilds"] - rebuilds_before
    if strong:
        dg = g.dist_geometry()
        region["migrations_since_partition"] = dg["migrations"]
        region["slot_space_resizes"] = dg["slot_resizes"]
    # (substeps of warm-up + timed region that carried the launches of the conditional re-sort; the others went without,
    # inside the quiet time the last re-sort estimated)
    region["resort_check_launches_since_finalize"] = st["resort_checks"] - 1   # (Finalize's own sort is one)
    region["substeps_since_finalize"] = args.warmup + args.steps
    # The steady state next to the driver's window: SURVEY 8(d)'s 200 substeps after 20 (the cloth has picked up
    # speed, re-sorts included), whatever --steps / --warmup were.  Not `value`.
    steady = None
    if world == 1 and chain is None and args.sort_every == 0:
        g.destroy()
        g = make_engine(1234)
        run(20)
        g.gpu_sync()
        r0 = g.stats()["rebuilds"]
        ts = time.perf_counter()
        run(200)
        g.gpu_sync()
        el_s = time.perf_counter() - ts
        steady = dict(ms_per_step=el_s / 200 * 1e3, substeps_per_s=200 / el_s, steps=200, warmup=20,
                      rebuilds=g.stats()["rebuilds"] - r0)

    # The reference's own call pattern (cuda_mpm_test.cc:64-74): five GpuMpmSolver calls per substep, GpuSync() after every
    # frame of 40 substeps.  Not `value` (which goes through mpm_run_substeps, an entry point the reference does not have).
    ref_pattern = None
    if world == 1 and chain is None and args.sort_every == 0:
        g.destroy()
        g = make_engine(1234)

        def frame(n=40):
            for _ in range(n):
                g.rebuild_mapping(False)
                g.calc_fem_state_and_force(dt)
                g.particle_to_grid(dt)
                g.update_grid(-1)
                g.grid_to_particle(dt)
            GpuMpm.device_synchronize()   # GpuSync(), no state argument

        frame(20)
        r0 = g.stats()["rebuilds"]
        frames = 4
        ts = time.perf_counter()
        for _ in range(frames):
            frame()
        el_r = time.perf_counter() - ts
        st_r = g.stats()
        assert st_r["error_flags"] == 0, st_r   # (GpuSync() reports no simulation errors: ADVICE r4)
        ref_pattern = dict(ms_per_step=el_r / (40 * frames) * 1e3, substeps_per_s=40 * frames / el_r, frames=frames,
                           substeps_per_frame=40, warmup_substeps=20, rebuilds=st_r["rebuilds"] - r0,
                           calls="RebuildMapping(false), CalcFemStateAndForce, ParticleToGrid, UpdateGrid, GridToParticle per "
                                 "substep; GpuSync() per frame (cuda_mpm_test.cc:64-74)")

    # per-kernel timing with HIP events on the engine's stream: a separate, un-timed pass over the
    # SAME substeps (fresh engine, same scene, same warm-up), so that the kernel durations describe
    # the timed region and not whatever the cloth does after it.  (Several ranks: the rank's own
    # engine as it stands after the run, without the exchange -- kernel durations only.)
    if world == 1:
        g.destroy()
        g = make_engine(1234)
        g.run_substeps(args.warmup, dt, -1)
        g.gpu_sync()
    phases, tot_ms = g.profile_substeps(min(args.steps, 4096 if world == 1 else 20), dt, -1)
    g.gpu_sync()
    st = g.stats()
    ncells = 64 * st["touched_blocks"]
    KERNEL_OF = kernel_names(g.fast_math, strong)
    # the particles this rank's kernels worked on (all of them unless the domain is partitioned)
    ab = algorithmic_bytes(st["active_faces"] + st["active_vertices"], st["active_faces"], st["active_vertices"], ncells)
    # mpm_run_substeps / mpm_profile_substeps: k_p2g also sums the vertex forces of its work items (no k_vforce launch).
    # Its algorithmic bytes stay the P2G row of SURVEY 8(d), 116 B per particle + 16 B per cell: the vertices' x, v and f
    # are IN those 116 bytes (rounds 3 - 4 added the 36 B per vertex of the FEM row on top, which counted the
    # vertices' x, v twice -- the PMC traffic came out BELOW the "algorithmic" figure).  The old figure is kept beside it.
    p2g_with_vertex_row = ab["p2g"] + ab["vforce"]
    if world == 1:
        ab["vforce"] = 0
    dom = max(KERNEL_OF, key=lambda k: phases[k])
    # An interval between two events holds the kernel AND the two event packets' own processing.  `achieved` / `frac`
    # are taken on the RAW interval -- the conservative figure: the kernel cannot have taken longer.  mpm_profile_substeps
    # also records one explicit back-to-back event pair per substep (the vertex-force slot: k_p2g does that work, nothing
    # is launched between its two events), i.e. the cost of an event pair measured live; `*_net` are the same figures
    # with it subtracted.  The rocprofv3 kernel trace of this command (profiles/rNN_kernel_stats.csv) lies between the two.
    ev_ms = phases["vforce"]
    dom_ms = phases[dom]
    dom_net_ms = max(dom_ms - ev_ms, 1e-6)
    ach_raw = ab[dom] / (dom_ms * 1e-3) / 1e9
    ach_net = ab[dom] / (dom_net_ms * 1e-3) / 1e9
    # ONE fraction for the dominant kernel: the one on the kernel's average duration in the committed
    # rocprofv3 kernel trace of this command (profiles/rNN_kernel_stats.csv).  The live HIP-event interval of this run
    # brackets it -- the raw interval holds two event packets on top of the kernel, the net one subtracts a measured event
    # pair, which over-corrects -- and is kept under names of its own (`live_event_interval`); without a committed trace for
    # this configuration the raw interval, the conservative figure, stands in.
    prof_ns = measured_kernel_ns(KERNEL_OF[dom], args.config) if world == 1 else None
    kernel_ms = prof_ns * 1e-6 if prof_ns else dom_ms
    ach = ab[dom] / (kernel_ms * 1e-3) / 1e9
    # the whole job: every particle once per substep (strong: one copy; weak: one copy per rank)
    copies = 1 if (strong or world == 1) else world
    # (cells: rank 0's count; a partitioned domain has about `world` times as many, a 1% term)
    job_bytes = copies * algorithmic_bytes(npart, nf, nv, ncells * world if strong else ncells)["total"]
    # which bound the dominant kernel sits on, from the committed SQ counters of this command: vector instructions per
    # launch x 4 cycles (one wave's issue cost, MI355X_MICROARCH.md) / 1024 SIMDs / 2.4 GHz = the time the vector ALUs
    # need for them if all 1024 issue all the time
    sq = measured_sq(KERNEL_OF[dom], args.config) if world == 1 else None
    valu_issue_ms = sq["SQ_INSTS_VALU"] * 4.0 / 1024.0 / 2.4e9 * 1e3 if sq and sq.get("SQ_INSTS_VALU") else None
    roofline = dict(bound="hbm", kernel=KERNEL_OF[dom], achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=ach / HBM_PEAK_GBS, traffic=measured_traffic(KERNEL_OF[dom], args.config) if world == 1 else None,
                    valu_issue_ms=valu_issue_ms,
                    valu_issue_note="SQ_INSTS_VALU of profiles/sq_counters.json (replayed) x 4 cycles / 1024 SIMDs / 2.4 GHz (the "
                                    "kernel runs nearer 2.05 GHz: SQ_WAVE_CYCLES): against kernel_ms it says how much of the "
                                    "kernel is vector-instruction issue -- k_p2g spends about half of its time issuing its "
                                    "~580 vector instructions per 64 particles and the rest on the MFMA pipe and on latencies "
                                    "that four waves per SIMD do not hide; it is not HBM-bound, although the path's roofline is "
                                    "HBM's",
                    algorithmic_bytes_per_launch_with_vertex_row=p2g_with_vertex_row if dom == "p2g" else None,
                    traffic_source="profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command, collected by scripts/collect_profiles.sh and committed (replayed, not "
                                   "measured in this run)",
                    algorithmic_bytes_per_launch=ab[dom], kernel_ms=kernel_ms,
                    kernel_ms_source=("average duration of this kernel in the committed rocprofv3 --kernel-trace --stats run of this "
                                      "command (profiles/pmc_traffic.json <- profiles/rNN_kernel_stats.csv)") if prof_ns else
                                     "HIP-event interval around the kernel in this run, raw (no committed trace for this configuration)",
                    live_event_interval=dict(
                        raw_ms=dom_ms, net_ms=dom_net_ms, event_pair_ms=ev_ms, frac_on_raw=ach_raw / HBM_PEAK_GBS,
                        frac_on_net=ach_net / HBM_PEAK_GBS,
                        brackets_the_trace=(dom_net_ms <= kernel_ms * 1.05 and kernel_ms <= dom_ms * 1.05) if prof_ns else None,
                        note="HIP events on the engine's stream around the kernel, measured in THIS run (mpm_profile_substeps): "
                             "raw = with the two event packets, net = minus one measured back-to-back event pair"),
                    substep_achieved=job_bytes / (el / args.steps) / 1e9,
                    substep_frac=job_bytes / (el / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
                    steady_state_frac=(job_bytes / (steady["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if steady else None,
                    reference_call_pattern_frac=(job_bytes / (ref_pattern["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if ref_pattern else None,
                    phase_ms=phases,
                    phase_note="separate pass with HIP events around every phase (mpm_profile_substeps); it launches the "
                               "re-sort kernels with every substep; the timed run launches none while the quiet time of the "
                               "last re-sort lasts (the stats() call in front of the window re-arms it: normally no check "
                               "launch in a 20-substep window), then with every fourth substep (gated substeps); "
                               "the vertex-force slot is empty (the vertex lanes of k_p2g do that work): its interval is "
                               "the cost of an event pair; phase_ms are raw intervals")

    team_contact = None
    engine_fast_math = g.fast_math
    if world > 1 and strong and not args.no_contact_leg:
        if native:
            g.chain_destroy()
        g.destroy()
        try:
            team_contact = team_contact_leg(rank, world, local_rank, args.config)
        except Exception as exc:  # noqa: BLE001  (the headline line must come out whatever this leg does)
            team_contact = dict(error=repr(exc))
    if rank == 0:
        if world == 1:
            par = "single GPU"
        elif strong:
            par = (f"{world} GPUs share ONE domain: x slabs cut at blocks {cuts}, ghost bands "
                   f"{geometry.get('face_band_cells')} / {geometry.get('vertex_band_cells')} cells (faces / vertices), "
                   f"zone {zone} blocks, migration {geometry['migrate_every']}, 1 rank/GPU, {transport} per substep")
        else:
            par = f"{world} GPUs: x-tiled patches, 1 rank/GPU, {transport} halo of grid-block sums per substep"
        out = dict(metric="mpm_substeps_per_sec_1M_particles", value=copies * args.steps / el, unit="substeps/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=el / args.steps * 1e3,
                   higher_is_better=True, scaling="strong" if strong or world == 1 and args.scaling == "strong" else "weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   arithmetic=("fast math (mpm_set_fast_math: hardware reciprocal / rsqrt + one Newton step in k_fem)" if engine_fast_math
                               else "correctly rounded divisions and square roots in k_fem (the default; MPM_FAST_MATH=1 / "
                                    "mpm_set_fast_math select the approximations: k_fem 1.7 us shorter in the kernel traces)"),
                   config=dict(workload=f"{args.config}: {npart} particles ({nf} faces + {nv} vertices), "
                                        f"{1 << bits}^3 grid, corotated cloth, no contact, dt={dt}",
                               particles_total=npart * copies, particles_rank0=st["active_faces"] + st["active_vertices"],
                               grid=f"{1 << bits}^3", touched_blocks=st["touched_blocks"], rebuilds=st["rebuilds"],
                               resorts_in_timed_window=region["resorts_in_timed_window"],
                               slot_sort_every=args.sort_every, parallelism=par, geometry=geometry if world > 1 else None),
                   transport=transport, roofline=roofline)
        out["timed_region"] = region
        if steady is not None:
            out["steady_state"] = steady
        if ref_pattern is not None:
            out["reference_call_pattern"] = ref_pattern
        if team_contact is not None:
            out["contact"] = team_contact
        if not args.no_contact_leg and world == 1:
            g.destroy()
            out["contact"] = contact_leg(local_rank)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(bits, layers, res, dt, args.cpu_budget)
        sys.stdout.flush()
        if saved_stdout is not None:
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout is not None:
            os.dup2(2, 1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
