"""The DIRECT halo exchange (mpm_chain_direct_*: the pack kernel stores the zone sums straight into the neighbour's
receive buffer -- peer memory mapped through a HIP IPC handle --, sequence flags instead of RCCL's kernel and staging copy;
VERDICT r4, item 4a).  The reference has no multi-GPU path (multibody/gpu_mpm/settings.h:40).

What one GPU can check, and what it cannot: the protocol (two parities of receive buffers, flags that only grow, the
bounded wait), the indexing (which zone goes into which of the neighbour's buffers) and the time-out run here -- in a ring
of one, and between two and three PROCESSES that share the card and map each other's buffers through IPC handles.  The
ordering of peer stores across two devices over xGMI cannot: on one device every store lands in the same memory."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_ring_of_one_direct_equals_rccl_ring_of_one():
    import torch  # noqa: F401  (before RCCL is bound: one copy of its dependencies per process, see csrc/mpm_chain.h)
    from drake_amd import ARR, GpuMpm
    from tests.test_chain_native_gpu import _engine, DT, STEPS
    cut_lo, cut_hi, pitch, zone, cap = 6, 10, 4, 2, 256
    a = _engine()
    a.chain_init(GpuMpm.chain_unique_id(), 0, 1, cut_lo, cut_hi, pitch, zone, cap, periodic=True)
    a.chain_substeps(STEPS, DT, -1)
    a.gpu_sync()
    b = _engine()
    b.chain_init(None, 0, 1, cut_lo, cut_hi, pitch, zone, cap, periodic=True)     # geometry only: no RCCL
    b.chain_direct_prepare()
    b.chain_direct_connect(None, None)                                              # its own neighbour: no handle
    b.chain_substeps(STEPS // 2, DT, -1)
    b.chain_substeps(STEPS - STEPS // 2, DT, -1)
    b.gpu_sync()
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0
    solo = _engine()
    solo.run_substeps(STEPS, DT, -1)
    assert np.abs(a.download(ARR.VELOCITIES) - solo.download(ARR.VELOCITIES)).max() > 1e-3   # (the exchange matters)
    # deterministic engines, the same sums added in the same order: to the bit
    assert np.array_equal(a.download(ARR.VELOCITIES), b.download(ARR.VELOCITIES))
    assert np.array_equal(a.download(ARR.POSITIONS), b.download(ARR.POSITIONS))
    a.chain_destroy()
    b.chain_destroy()


def test_a_neighbour_that_never_signals_is_an_error_not_a_hang(monkeypatch):
    """A ring of one whose signal kernel is left out (MPM_HALO_DEBUG_MUTE): the one-wave wait gives up after
    MPM_HALO_TIMEOUT_S, the substep completes on whatever the buffers held, and the engine reports MPM_ERR_HALO at its
    next synchronisation -- an error code, not a hung device."""
    import time
    from drake_amd import MpmError
    from tests.test_chain_native_gpu import _engine, DT
    monkeypatch.setenv("MPM_HALO_TIMEOUT_S", "0.2")
    monkeypatch.setenv("MPM_HALO_DEBUG_MUTE", "1")
    g = _engine()
    g.chain_init(None, 0, 1, 6, 10, 4, 2, 256, periodic=True)
    g.chain_direct_prepare()
    g.chain_direct_connect(None, None)
    t0 = time.perf_counter()
    g.chain_substeps(2, DT, -1)
    with pytest.raises(MpmError) as err:
        g.gpu_sync()
    el = time.perf_counter() - t0
    assert 0.3 < el < 20.0, el          # two substeps, 0.2 s each
    assert err.value.code == -8, err.value    # MPM_ERR_HALO
    g.chain_destroy()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd import ARR, GpuMpm, scenes
    from tests.test_dist_gpu import BITS, DT, STEPS, _centre, _patch
    g = GpuMpm(BITS)
    gx = _centre(rank, world)
    sheets = [(p + np.array([0.5 - gx, 0, 0], np.float32), v, i) for p, v, i in _patch(rank, gx)]
    scenes.populate(g, sheets)
    g.chain_init(None, rank, world, 6, 10, 4, 2, 256)
    handles = [None] * world
    dist.all_gather_object(handles, g.chain_direct_prepare())
    g.chain_direct_connect(handles[rank - 1] if rank > 0 else None, handles[rank + 1] if rank < world - 1 else None)
    dist.barrier()
    for _ in range(STEPS):        # (one call per substep: the ranks drift apart in time, the flags keep them in step)
        g.chain_substeps(1, DT, -1)
    g.gpu_sync()
    pos = g.download(ARR.POSITIONS)
    pos[:, 0] += gx - 0.5
    q.put((rank, pos, g.download(ARR.VELOCITIES), g.stats()["error_flags"]))
    dist.barrier()
    g.chain_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_processes_sharing_the_gpu_exchange_through_ipc_mapped_buffers(world):
    import torch.multiprocessing as mp
    from drake_amd import ARR, GpuMpm, scenes
    from tests.helpers import close
    from tests.test_dist_gpu import BITS, DT, STEPS, _centre, _patch
    ref = GpuMpm(BITS)
    patches = [_patch(r, _centre(r, world)) for r in range(world)]
    scenes.populate(ref, [s for pt in patches for s in pt])
    for _ in range(STEPS):
        ref.substep(DT, -1)
    ref.gpu_sync()
    rp, rv = ref.download(ARR.POSITIONS), ref.download(ARR.VELOCITIES)
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
    port = 29450 + (os.getpid() % 200) + 7 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
        close(pos, rp[idx[r]], scale=1.0, rtol=1e-5, what=f"direct halo, rank {r}/{world}: positions vs single engine")
        close(vel, rv[idx[r]], scale=vs, rtol=1e-4, what=f"direct halo, rank {r}/{world}: velocities vs single engine")
