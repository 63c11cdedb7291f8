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
    chain = DomainChain(g, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=512,
                        migrate_every=migrate_every, migrate_capacity=4096, device=torch.device("cuda", 0))
    roles0 = g.dist_roles()
    for _ in range(STEPS):
        chain.substep(DT, -1)
    g.gpu_sync()
    q.put((rank, g.dist_roles(), roles0, g.download(ARR.POSITIONS), g.download(ARR.VELOCITIES),
           g.download(ARR.DEFORMATION_GRADIENTS), g.stats()))
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
