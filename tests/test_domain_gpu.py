"""One domain cut into x slabs (mpm_dist_*, drake_amd/dist.py: DomainChain): two and three ranks
(processes sharing the test box's one GPU, gloo transport staged through host memory) together hold
ONE cloth stack that straddles the cuts and drifts across them; the union of what the ranks own
must equal a single engine that holds everything."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
BITS, DT, STEPS = 6, 1e-3, 48


def _scene():
    from drake_amd import scenes
    sheets = scenes.cloth_stack(3, 40, BITS, z0=0.5, side=0.4, seed=21, vel_amp=0.3)
    for pos, vel, idx in sheets:
        vel[:, 0] += 1.2      # 0.077 cells per substep: 3.7 cells over the run, across the cuts
        vel[:, 2] -= 0.4
    return sheets


def _worker(rank, world, cuts, port, q, migrate_every):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd import ARR, GpuMpm, scenes
    from drake_amd.dist import DomainChain
    g = GpuMpm(BITS)
    scenes.populate(g, _scene())     # every rank: the whole scene
    st_whole = g.stats()
    chain = DomainChain(g, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=512,
                        migrate_every=migrate_every, migrate_capacity=4096, device=torch.device("cuda", 0))
    st_init = g.stats()
    roles0 = g.dist_roles()
    sent_ok = True
    for _ in range(STEPS):
        chain.substep(DT, -1)
        # the staged exchange sends only what the pack kernel filled: 16 + 4 count + 1024 count bytes per direction,
        # count = the zone's active blocks (the capacity, 512 blocks = 526 KB, never travels)
        for n, zone in ((chain.left, chain.zone_lo), (chain.right, chain.zone_hi)):
            if n is None:
                continue
            count = chain.blocks_sent[n]
            sent_ok &= chain.bytes_sent[n] == 16 + 4 * count + 1024 * count
            sent_ok &= 0 < count < 512
    # (the set only changes with re-sorts: compare the last exchange with the engine's own count of the zone)
    for n, zone in ((chain.left, chain.zone_lo), (chain.right, chain.zone_hi)):
        if n is not None:
            sent_ok &= chain.blocks_sent[n] == g.halo_zone_blocks(zone[0], zone[1])
    g.gpu_sync()
    st = dict(g.stats(), count_sized_messages=bool(sent_ok), stats_after_init=st_init, stats_whole_scene=st_whole)
    q.put((rank, g.dist_roles(), roles0, g.download(ARR.POSITIONS), g.download(ARR.VELOCITIES),
           g.download(ARR.DEFORMATION_GRADIENTS), st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts,migrate_every", [(2, [0, 8, 16], 4), (3, [0, 6, 10, 16], 3)])
def test_partitioned_domain_matches_single_engine(world, cuts, migrate_every):
    import torch.multiprocessing as mp
    from drake_amd import ARR, GpuMpm, scenes
    from tests.helpers import close
    ref = GpuMpm(BITS)
    scenes.populate(ref, _scene())
    x0 = ref.download(ARR.POSITIONS)
    ref.run_substeps(STEPS, DT, -1)
    ref.gpu_sync()
    rp, rv, rF = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES), ref.download(ARR.DEFORMATION_GRADIENTS)
    n, nf = ref.n_particles, ref.n_faces

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 200) + 11 * world
    procs = [ctx.Process(target=_worker, args=(r, world, cuts, port, q, migrate_every)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=600)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owners = np.zeros(n, np.int32)
    owners0 = np.zeros(n, np.int32)
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    F = np.full((nf, 9), np.nan, np.float32)
    for r in range(world):
        roles, roles0, p_r, v_r, F_r, st = got[r]
        assert st["error_flags"] == 0, (r, st)
        assert st["count_sized_messages"], (r, st)
        # per-rank topology: after mpm_dist_init the rank's particle arrays hold 1.5 x its share + ghost bands (+ head
        # room), not the whole scene, and the scene-sized topology tables are gone (13 bytes per particle of the scene
        # remain: the id -> slot map, the slot-order bookkeeping, one byte of migration state)
        si, sw = st["stats_after_init"], st["stats_whole_scene"]
        held_f, held_v = si["active_faces"], si["active_vertices"]
        assert si["face_slots"] <= held_f + held_f // 2 + 256 and si["vertex_slots"] <= held_v + held_v // 2 + 256, si
        assert si["face_slots"] <= sw["face_slots"] and si["vertex_slots"] <= sw["vertex_slots"], (si, sw)
        if world == 2:   # (this small scene's ghost bands are a third of a three-rank slab: only here is there room to give back)
            assert si["face_slots"] < sw["face_slots"], (si, sw)
        bytes_per_slot = sw["particle_bytes"] / (sw["face_slots"] + sw["vertex_slots"])
        assert si["particle_bytes"] <= 1.6 * bytes_per_slot * 1.5 * (held_f + held_v) + (1 << 20), (si, sw)
        assert si["scene_index_bytes"] <= 13 * n + 64 and sw["scene_index_bytes"] > 40 * n * 0.6, (si, sw)
        own = roles == 1
        owners += own
        owners0 += roles0 == 1
        pos[own], vel[own] = p_r[own], v_r[own]
        F[own[:nf]] = F_r[own[:nf]]
        # a ghost copy is the same particle, advanced redundantly: bit-identical to the owner's
        got[r] = (roles, p_r, v_r)
    # every particle has exactly one owner, before and after
    assert np.all(owners0 == 1) and np.all(owners == 1)
    for r in range(world):
        roles, p_r, v_r = got[r]
        gh = roles == 2
        assert gh.any()
        assert np.array_equal(p_r[gh], pos[gh]) and np.array_equal(v_r[gh], vel[gh])
    # ownership followed the motion: particles did change hands
    cell0 = np.minimum((x0[:, 0] * (1 << BITS) - 0.5).astype(np.int64), (1 << BITS) - 3)
    start_owner = np.searchsorted(np.array(cuts[1:-1]) * 4, cell0, side="right")
    end_owner = np.array([int(np.argmax([got[r][0][i] == 1 for r in range(world)])) for i in range(0, n, 97)])
    assert np.count_nonzero(end_owner != start_owner[::97]) > 0
    vs = max(float(np.abs(rv).max()), 1.0)
    close(pos, rp, scale=1.0, rtol=1e-5, what=f"domain x{world}: positions vs single engine")
    close(vel, rv, scale=vs, rtol=1e-4, what=f"domain x{world}: velocities vs single engine")
    close(F, rF, scale=1.0, rtol=1e-4, what=f"domain x{world}: F vs single engine")


# ---- distributed contact solve ---------------------------------------------------------------
FLOOR_Z = 0.5


def _contact_scene():
    from drake_amd import scenes
    sheets = scenes.cloth_stack(2, 36, BITS, z0=FLOOR_Z - 0.004, side=0.4, seed=33, vel_amp=0.3)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
        vel[:, 0] += 0.3
    return sheets


def _floor():
    from drake_amd import Collider
    return [Collider(0, body=0, p_WB=(0.5, 0.5, FLOOR_Z))]


def _contact_worker(rank, world, cuts, port, q, exact):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd import ARR, GpuMpm, scenes
    from drake_amd.dist import DomainChain
    g = GpuMpm(BITS)
    scenes.populate(g, _contact_scene())
    chain = DomainChain(g, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=512,
                        migrate_every=4, migrate_capacity=4096, device=torch.device("cuda", 0))
    chain.install_contact_transport(512)
    out = []
    for step in range(3):
        # one coupled substep: the halves of the substep around the exchange, with the contact solve between
        # UpdateGrid and GridToParticle (deformable_driver.h:244-258)
        g.reallocate_external_bodies(1)
        if chain.migrate_every and step and step % chain.migrate_every == 0:
            chain.migrate()
        g.substep_begin(DT)
        chain.exchange()
        g.update_grid_from_sums(-1)
        n = g.generate_contact_pairs(_floor())
        r = g.update_contact(DT, 0.5, 1e5, 1e-3, exact_line_search=exact)
        cs = g.contact_stats()
        tau, f = g.external_body_force_to_host()
        g.grid_to_particle(DT)
        out.append((n, r, cs, f.copy()))
    g.gpu_sync()
    # one HIP runtime in the process: the staged transport copies through the engine, it does not open libamdhip64
    # itself (two runtimes tear each other's state down at exit)
    with open("/proc/self/maps") as f:
        hip_copies = {ln.split()[-1] for ln in f if "libamdhip64" in ln}
    st = dict(g.stats(), hip_runtime_copies=len(hip_copies))
    q.put((rank, g.dist_roles(), g.download(ARR.POSITIONS), g.download(ARR.VELOCITIES), out, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [False, True])
def test_distributed_contact_solve_matches_single_engine(exact):
    """Floor contact under a cloth that straddles the cut: two ranks, each with the contacts of the
    particles it owns; zone exchange of the per-node Hessian / gradient sums and all-reduce of the
    line-search scalars per Newton iteration.  Same iterations, same velocities, and the per-body
    impulses add up to the single engine's."""
    import torch.multiprocessing as mp
    from drake_amd import ARR, GpuMpm, scenes
    from tests.helpers import IMPULSE_RTOL, close, solve_tolerance
    world, cuts = 2, [0, 8, 16]
    ref = GpuMpm(BITS)
    scenes.populate(ref, _contact_scene())
    ref_out = []
    for step in range(3):
        ref.reallocate_external_bodies(1)
        ref.rebuild_mapping(False)
        ref.calc_fem_state_and_force(DT)
        ref.particle_to_grid(DT)
        ref.update_grid(-1)
        n = ref.generate_contact_pairs(_floor())
        r = ref.update_contact(DT, 0.5, 1e5, 1e-3, exact_line_search=exact)
        cs = ref.contact_stats()
        tau, f = ref.external_body_force_to_host()
        ref.grid_to_particle(DT)
        ref_out.append((n, r, cs, f.copy()))
    rp, rv = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30200 + (os.getpid() % 200) + (7 if exact else 0)
    procs = [ctx.Process(target=_contact_worker, args=(r, world, cuts, port, q, exact)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=600)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = ref.n_particles
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    for r in range(world):
        roles, p_r, v_r, out, st = got[r]
        assert st["error_flags"] == 0
        assert st["hip_runtime_copies"] == 1, st
        own = roles == 1
        pos[own], vel[own] = p_r[own], v_r[own]
    for step in range(3):
        n_ref, r_ref, cs_ref, f_ref = ref_out[step]
        n_sum = sum(got[r][3][step][0] for r in range(world))
        assert n_sum == n_ref and all(got[r][3][step][0] > 20 for r in range(world))   # both ranks have contacts
        its = [got[r][3][step][1]["iterations"] for r in range(world)]
        assert its[0] == its[1]                       # the ranks take the same decisions
        assert abs(its[0] - r_ref["iterations"]) <= max(1, r_ref["iterations"] // 10), (its, r_ref)
        dofs = sum(got[r][3][step][2]["dofs"] for r in range(world)) / world
        assert dofs == cs_ref["dofs"]                 # shared nodes counted once (both ranks hold the global count)
        f_sum = sum(got[r][3][step][3] for r in range(world))
        close(f_sum, f_ref, scale=float(np.abs(f_ref).max()), rtol=IMPULSE_RTOL, what=f"distributed contact: body impulse (exact={exact})")
    tol = solve_tolerance(ref_out[-1][2]["dofs"])
    close(pos, rp, scale=1.0, rtol=1e-5, what=f"distributed contact: positions (exact={exact})")
    close(vel, rv, scale=1.0, rtol=tol, what=f"distributed contact: velocities (exact={exact})")
