"""Two and three ranks (processes sharing the one GPU of the test box, gloo transport staged
through host memory) against a single engine that holds all cloth patches in one grid.  With three
ranks the middle one has two neighbours: both zones packed / both buffers added in one launch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
BITS, LAYERS, RES, SIDE, DT, STEPS = 6, 3, 18, 0.25, 1e-3, 8


def _patch(rank, center_x):
    from drake_amd import scenes
    return scenes.cloth_stack(LAYERS, RES, BITS, z0=0.5, side=SIDE, seed=40 + rank, vel_amp=0.5, center=(center_x, 0.5))


def _centre(rank, world):
    return 0.5 - 0.125 * (world - 1) + 0.25 * rank


def _worker(rank, world, port, q, split):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd import ARR, GpuMpm, scenes
    from drake_amd.dist import HaloChain
    g = GpuMpm(BITS)
    # same patch as the reference's, translated to the rank's local frame (centre x = 0.5)
    gx = _centre(rank, world)
    sheets = [(p + np.array([0.5 - gx, 0, 0], np.float32), v, i) for p, v, i in _patch(rank, gx)]
    scenes.populate(g, sheets)
    chain = HaloChain(g, rank, world, cut_lo_block=6, cut_hi_block=10, pitch_blocks=4, zone_blocks=2,
                      capacity_blocks=256, device=torch.device("cuda", 0), split=split)
    for _ in range(STEPS):
        chain.substep(DT, -1)
    g.gpu_sync()
    pos = g.download(ARR.POSITIONS)
    pos[:, 0] += gx - 0.5
    q.put((rank, pos, g.download(ARR.VELOCITIES), g.stats()["error_flags"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split", [(2, False), (3, False), (3, True)])
def test_chain_matches_single_engine(world, split):
    import torch.multiprocessing as mp
    from drake_amd import ARR, GpuMpm, scenes
    from tests.helpers import close
    # reference: all patches in one grid
    ref = GpuMpm(BITS)
    patches = [_patch(r, _centre(r, world)) for r in range(world)]
    scenes.populate(ref, [s for pt in patches for s in pt])
    for _ in range(STEPS):
        ref.substep(DT, -1)
    ref.gpu_sync()
    rp, rv = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES)
    # Finalize order is [faces | verts] over the concatenated cloths: split it back per patch
    nf = ref.n_faces
    idx, f0, v0 = [], 0, 0
    for pt in patches:
        nfp = sum(s[2].size // 3 for s in pt)
        nvp = sum(s[0].shape[0] for s in pt)
        idx.append(np.r_[f0:f0 + nfp, nf + v0:nf + v0 + nvp])
        f0 += nfp
        v0 += nvp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 200) + 7 * world + (3 if split else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, pos, vel, err = q.get(timeout=300)
        got[r] = (pos, vel, err)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vs = max(float(np.abs(rv).max()), 1.0)
    for r in range(world):
        pos, vel, err = got[r]
        assert err == 0
        close(pos, rp[idx[r]], scale=1.0, rtol=1e-5, what=f"rank {r}/{world}{' split' if split else ''} positions vs single engine")
        close(vel, rv[idx[r]], scale=vs, rtol=1e-4, what=f"rank {r}/{world}{' split' if split else ''} velocities vs single engine")
    # the patches do interact through the shared nodes: without the exchange the result differs
    solo = GpuMpm(BITS)
    scenes.populate(solo, patches[0])
    for _ in range(STEPS):
        solo.substep(DT, -1)
    assert np.abs(solo.download(ARR.VELOCITIES) - rv[idx[0]]).max() > 1e-3 * vs
