"""Coupled substeps on a PARTITIONED domain, device resident (BASELINE config 5's path: a cloth and
rigid bodies with prescribed motion on several ranks).  The reference has one device (multibody/gpu_mpm/settings.h:40);
its coupled loop is multibody/plant/deformable_driver.h:240-258 around cuda_mpm_solver.cu:274-570, which reads the
solver's global scalars back to the host several times per Newton iteration.  Round 5's partitioned solve re-created those
round trips across ranks; round 6's does not: the zone exchange of the per-node (H, G) sums and the rank-ordered sums of
the line-search rows are peer stores + sequence flags on the engines' streams (drake_amd/csrc/mpm_team.h), the host polls
the mailbox only, and mpm_run_coupled_substeps / mpm_world_coupled_substeps batch whole coupled substeps.

What one GPU can check: the protocol, the indexing, the rank-order sums and the decisions -- in in-process worlds of 2 and
4 ranks (regions named by pointer, every phase enqueued for all ranks in turn on one stream) and between two PROCESSES
that share the card and map each other's regions through HIP IPC handles.  Not the ordering of peer stores across two
devices over xGMI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
BITS, DT = 6, 1e-3
MU, K, D = 0.5, 1e5, 1e-3
FLOOR_Z = 0.5


def _scene():
    """three sheets over the middle of the domain (x blocks 5..11 of 16: across the cuts of 2 and 4 ranks), the lowest
    pressed into the floor, all drifting along x"""
    from drake_amd import scenes
    sheets = scenes.cloth_stack(3, 60, BITS, z0=FLOOR_Z - 0.004, side=0.44, seed=33, vel_amp=0.2)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
        vel[:, 0] += 0.8
    return sheets


def _colliders(t):
    """body 0: the floor; bodies 1, 2: capsules lying along y on the cloth, moving along x ACROSS the cuts (x = 0.5 is the
    cut of two ranks, 0.375 / 0.625 the outer cuts of four), with the rigid velocity they move with"""
    from drake_amd import Collider
    Ry = (0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0)    # row-major R_WB: world x = body y, world y = body z (the capsule's axis), world z = body x
    out = [Collider(0, body=0, p_WB=(0.5, 0.5, FLOOR_Z))]
    for b, (x0, vx) in enumerate(((0.485, 2.0), (0.64, -1.5)), start=1):
        out.append(Collider(3, body=b, p_WB=(x0 + vx * t, 0.5, FLOOR_Z + 0.028), R_WB=Ry, dims=(0.02, 0.12, 0.0), v=(vx, 0.0, 0.0)))
    return out


def _engine(sheets, bodies=3):
    from drake_amd import GpuMpm, scenes
    g = GpuMpm(BITS)
    g.set_deterministic(True)
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    g.reallocate_external_bodies(bodies)
    return g


CHUNKS = (1, 3, 4, 4, 4, 4, 4)     # substeps per call; the colliders are re-posed between the calls


def _single_engine_run(sheets):
    """the reference run: one engine, mpm_run_coupled_substeps per chunk, colliders re-posed between chunks"""
    from drake_amd import ARR
    g = _engine(sheets)
    res, logs, done = [], [], 0
    for k in CHUNKS:
        res += g.run_coupled_substeps(k, DT, _colliders(done * DT), MU, K, D)
        done += k
        logs.append(g.contact_log().copy())
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0
    tau, f = g.external_body_force_to_host()
    return dict(res=res, logs=logs, pos=g.download(ARR.POSITIONS), vel=g.download(ARR.VELOCITIES), tau=tau, f=f,
                dofs=g.contact_stats()["dofs"], n=g.n_particles)


def _check_against_single_engine(ref, per_rank_res, per_rank_logs, owned, pos, vel, f_sum, tau_sum, what):
    from tests.helpers import IMPULSE_RTOL, close, solve_tolerance
    world = len(per_rank_res)
    n_sub = len(ref["res"])
    assert all(len(r) == n_sub for r in per_rank_res)
    for s in range(n_sub):
        # every contact belongs to exactly one rank: the counts add up to the single engine's
        assert sum(per_rank_res[r][s]["contacts"] for r in range(world)) == ref["res"][s]["contacts"], (what, s)
        its = {per_rank_res[r][s]["iterations"] for r in range(world)}
        assert len(its) == 1, (what, s, its)          # all ranks take the same decisions
    assert max(r["contacts"] for r in ref["res"]) > 300
    assert sum(1 for s in range(n_sub) if all(per_rank_res[r][s]["contacts"] > 0 for r in range(world))) > 0   # every rank solves contacts of its own
    # the decisions of the solves whose logs were kept (the last of every chunk): every accepted step and every
    # evaluation count of the FIRST chunk's solve equal to the single engine's (the states are the same up to the
    # regrouping of the partition's float sums; later chunks are compared through iteration counts and fields)
    L_ref, L0 = ref["logs"][0], per_rank_logs[0][0]
    assert L0.shape == L_ref.shape and L_ref.shape[0] >= 2, (L0.shape, L_ref.shape)
    assert np.array_equal(L0[:, 3], L_ref[:, 3]), (what, L0[:, 3], L_ref[:, 3])       # alpha sequence
    assert np.array_equal(L0[:, 1], L_ref[:, 1]), (what, L0[:, 1], L_ref[:, 1])       # energy evaluations
    for r in range(1, world):   # ... and the same rows on every rank, to the bit (rank-ordered sums)
        for ch in range(len(ref["logs"])):
            assert np.array_equal(per_rank_logs[r][ch], per_rank_logs[0][ch]), (what, r, ch)
    for s in range(n_sub):
        it_ref = ref["res"][s]["iterations"]
        assert abs(per_rank_res[0][s]["iterations"] - it_ref) <= max(1, it_ref // 8), (what, s, per_rank_res[0][s], ref["res"][s])
    assert np.all(owned == 1)
    tol = solve_tolerance(ref["dofs"])
    close(pos, ref["pos"], scale=1.0, rtol=1e-5, what=f"{what}: positions vs single engine")
    close(vel, ref["vel"], scale=1.0, rtol=tol, what=f"{what}: velocities vs single engine")
    fscale = float(np.abs(ref["f"]).max())
    close(f_sum, ref["f"], scale=fscale, rtol=IMPULSE_RTOL, what=f"{what}: per-body impulses (summed over the ranks)")
    close(tau_sum, ref["tau"], scale=max(float(np.abs(ref["tau"]).max()), fscale), rtol=IMPULSE_RTOL,
          what=f"{what}: per-body angular impulses (summed over the ranks)")
    assert np.abs(ref["f"][1:]).max() > 1e-3 * fscale      # the capsules do push


@pytest.mark.parametrize("world,cuts", [(2, [0, 8, 16]), (4, [0, 6, 8, 10, 16])])
def test_in_process_world_coupled_substeps_match_single_engine(world, cuts):
    import torch
    from drake_amd import ARR
    from drake_amd.dist import LocalWorld
    sheets = _scene()
    ref = _single_engine_run(sheets)
    engines = [_engine(sheets) for _ in range(world)]
    w = LocalWorld(engines, cuts, zone_blocks=1 if world == 4 else 2, ghost_cells=0, ghost_margin_cells=0, capacity_blocks=512,
                   migrate_every=0, migrate_capacity=1 << 14, device=torch.device("cuda", 0))
    w.enable_team(512)
    res = [[] for _ in range(world)]
    logs = [[] for _ in range(world)]
    done = 0
    for k in CHUNKS:
        out = w.coupled_substeps(k, DT, _colliders(done * DT), MU, K, D)
        done += k
        for r in range(world):
            res[r] += out[r]
            logs[r].append(engines[r].contact_log().copy())
    w.sync()
    n = ref["n"]
    owned = np.zeros(n, np.int32)
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    f_sum, tau_sum = np.zeros_like(ref["f"]), np.zeros_like(ref["tau"])
    for g in engines:
        st = g.stats()
        assert st["error_flags"] == 0, st
        roles = g.dist_roles()
        own = roles == 1
        owned += own
        pos[own], vel[own] = g.download(ARR.POSITIONS)[own], g.download(ARR.VELOCITIES)[own]
        tau, f = g.external_body_force_to_host()
        f_sum += f
        tau_sum += tau
    assert w.migrations >= 1
    _check_against_single_engine(ref, res, logs, owned, pos, vel, f_sum, tau_sum, f"in-process world of {world}")


def test_a_rank_that_never_arrives_is_an_error_code_not_a_hang(monkeypatch):
    """Every wait of the team transport is bounded: a world of two of which only rank 0 enters the solve -- rank 0 waits for
    rank 1's status, gives up after MPM_HALO_TIMEOUT_S, finishes its solve without touching the grid and reports
    MPM_ERR_HALO at its next synchronisation.  The device is alive afterwards."""
    import time
    import torch
    from drake_amd import MpmError
    from drake_amd.dist import LocalWorld
    monkeypatch.setenv("MPM_HALO_TIMEOUT_S", "0.2")
    sheets = _scene()
    engines = [_engine(sheets) for _ in range(2)]
    w = LocalWorld(engines, [0, 8, 16], zone_blocks=2, capacity_blocks=512, migrate_every=0, migrate_capacity=1 << 14,
                   device=torch.device("cuda", 0))
    w.enable_team(512)
    g = engines[0]
    with torch.cuda.stream(w.stream):
        w._substep(DT, -1)                  # (a grid to solve on: the contact-free substep of the whole world)
        g.generate_contact_pairs(_colliders(0.0), want_count=False)
        t0 = time.perf_counter()
        r = g.update_contact(DT, MU, K, D)
        el = time.perf_counter() - t0
    assert r["iterations"] == 0
    assert 0.15 < el < 10.0, el
    with pytest.raises(MpmError) as err:
        g.gpu_sync()
    assert err.value.code == -8, err.value      # MPM_ERR_HALO
    assert engines[1].stats()["error_flags"] == 0


def _worker(rank, world, cuts, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd import ARR
    from drake_amd.dist import DomainChain
    g = _engine(_scene())
    chain = DomainChain(g, rank, world, cuts, zone_blocks=2, ghost_cells=0, ghost_margin_cells=0, capacity_blocks=512,
                        migrate_every=0, migrate_capacity=1 << 14, device=torch.device("cuda", 0))
    ok = chain.enable_team(512)
    res, logs = [], []
    if ok:
        done = 0
        for k in CHUNKS:
            res += chain.coupled_substeps(k, DT, _colliders(done * DT), MU, K, D)
            done += k
            logs.append(g.contact_log().copy())
        g.gpu_sync()
    tau, f = g.external_body_force_to_host()
    q.put((rank, ok, chain.team_error, g.dist_roles(), g.download(ARR.POSITIONS), g.download(ARR.VELOCITIES), res, logs, tau, f,
           g.stats(), chain.migrations))
    dist.barrier()
    g.chain_destroy()
    dist.destroy_process_group()


def test_two_processes_sharing_the_gpu_run_coupled_substeps_over_ipc_mapped_regions():
    import torch.multiprocessing as mp
    world, cuts = 2, [0, 8, 16]
    ref = _single_engine_run(_scene())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30800 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, cuts, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=600)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(got[r][0] for r in range(world)), [got[r][1] for r in range(world)]
    n = ref["n"]
    owned = np.zeros(n, np.int32)
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    f_sum, tau_sum = np.zeros_like(ref["f"]), np.zeros_like(ref["tau"])
    res, logs = [], []
    for r in range(world):
        ok, why, roles, p_r, v_r, res_r, logs_r, tau, f, st, migrations = got[r]
        assert st["error_flags"] == 0, st
        own = roles == 1
        owned += own
        pos[own], vel[own] = p_r[own], v_r[own]
        f_sum += f
        tau_sum += tau
        res.append(res_r)
        logs.append(logs_r)
        assert migrations >= 1
    _check_against_single_engine(ref, res, logs, owned, pos, vel, f_sum, tau_sum, "two processes over IPC")


def test_exact_line_search_and_a_buffer_overflow_on_a_team(monkeypatch):
    """The device-resident EXACT line search over the team transport (every probe of the root finder: this rank's sums into
    every rank's slot, all ranks' sums added in rank order, the state machine of cuda_mpm_solver.cu:383-471 advanced alike
    on every rank), and a refusal that only ONE rank has a reason for: pair buffers too small on the ranks (64 pairs at
    first) -- k_team_status makes every rank refuse together, every host grows its buffers, makes the pairs again and
    repeats; the result is that of buffers that were large enough."""
    import torch
    from drake_amd import ARR
    from drake_amd.dist import LocalWorld
    from tests.helpers import IMPULSE_RTOL, close, solve_tolerance
    sheets = _scene()
    ref = _engine(sheets)
    ref_res = ref.run_coupled_substeps(6, DT, _colliders(0.0), MU, K, D, exact_line_search=True)
    ref.gpu_sync()
    rp, rv = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES)
    _, rf = ref.external_body_force_to_host()
    monkeypatch.setenv("MPM_CT_INITIAL_CAPACITY", "64")
    engines = [_engine(sheets) for _ in range(2)]
    monkeypatch.delenv("MPM_CT_INITIAL_CAPACITY")
    w = LocalWorld(engines, [0, 8, 16], zone_blocks=2, capacity_blocks=512, migrate_every=0, migrate_capacity=1 << 14,
                   device=torch.device("cuda", 0))
    w.enable_team(512)
    out = w.coupled_substeps(6, DT, _colliders(0.0), MU, K, D, exact_line_search=True)
    w.sync()
    assert sum(g.contact_counters()["repeated_overflow"] for g in engines) >= 2      # both ranks repeated, together
    n = ref.n_particles
    pos, vel = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32)
    f_sum = np.zeros_like(rf)
    for g in engines:
        assert g.stats()["error_flags"] == 0
        own = g.dist_roles() == 1
        pos[own], vel[own] = g.download(ARR.POSITIONS)[own], g.download(ARR.VELOCITIES)[own]
        f_sum += g.external_body_force_to_host()[1]
    for s in range(6):
        assert out[0][s]["contacts"] + out[1][s]["contacts"] == ref_res[s]["contacts"]
        assert out[0][s]["iterations"] == out[1][s]["iterations"]
        assert abs(out[0][s]["iterations"] - ref_res[s]["iterations"]) <= max(1, ref_res[s]["iterations"] // 8), (s, out[0][s], ref_res[s])
    assert max(r["iterations"] for r in ref_res) >= 2
    tol = solve_tolerance(ref.contact_stats()["dofs"])
    close(pos, rp, scale=1.0, rtol=1e-5, what="team, exact search: positions vs single engine")
    close(vel, rv, scale=1.0, rtol=tol, what="team, exact search: velocities vs single engine")
    close(f_sum, rf, scale=float(np.abs(rf).max()), rtol=IMPULSE_RTOL, what="team, exact search: per-body impulses")
