"""The library's own multi-GPU chain (mpm_chain_*: RCCL send/recv on the engine's stream).  One GPU
is all the test box has, so the chain is closed into a ring of one rank: the rank is its own left
and right neighbour and RCCL sends to itself.  The reference result is the same substep with the
two transfers done as plain device copies."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BITS, DT, STEPS = 6, 1e-3, 6


def _engine():
    from drake_amd import GpuMpm, scenes
    g = GpuMpm(BITS)
    # (canonical particle order after every re-sort: the two engines of a comparison then sum in the same order, and
    # what is left between them is the transport -- not the arrival order of the counting sort's atomics, which alone
    # is worth ~3e-6 m/s after six substeps of this scene)
    g.set_deterministic(True)
    # one stack wide enough to reach both zones (cuts at blocks 6 and 10 of 16)
    scenes.populate(g, scenes.cloth_stack(3, 30, BITS, z0=0.5, side=0.45, seed=7, vel_amp=0.5))
    return g


def test_ring_of_one_equals_device_copies():
    import torch
    from drake_amd import ARR, GpuMpm
    cut_lo, cut_hi, pitch, zone, cap = 6, 10, 4, 2, 256
    # reference: pack -> copy (what I send left arrives from the right, and vice versa) -> add
    ref = _engine()
    nbytes = ref.halo_buffer_bytes(cap)
    dev = torch.device("cuda", 0)
    send_l, send_r, recv_l, recv_r = (torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(4))
    zones = [(cut_lo - zone, cut_lo + zone - 1, +pitch), (cut_hi - zone, cut_hi + zone - 1, -pitch)]
    za = ref.halo_zone_args(zones, [send_l.data_ptr(), send_r.data_ptr()])
    ra = ref.halo_buffer_args([recv_l.data_ptr(), recv_r.data_ptr()])
    stream = torch.cuda.Stream()
    ref.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        for _ in range(STEPS):
            ref.substep_begin_halo(DT, za, cap)
            recv_r.copy_(send_l, non_blocking=True)
            recv_l.copy_(send_r, non_blocking=True)
            ref.substep_end_halo(DT, -1, ra, cap)
    ref.gpu_sync()
    assert ref.stats()["error_flags"] == 0
    # native: RCCL to self
    g = _engine()
    g.chain_init(GpuMpm.chain_unique_id(), 0, 1, cut_lo, cut_hi, pitch, zone, cap, periodic=True)
    g.chain_substeps(STEPS, DT, -1)
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0
    # the exchanged sums matter (otherwise this test proves nothing)
    solo = _engine()
    solo.run_substeps(STEPS, DT, -1)
    v_ref, v_nat, v_solo = (x.download(ARR.VELOCITIES) for x in (ref, g, solo))
    assert np.abs(v_ref - v_solo).max() > 1e-3
    np.testing.assert_allclose(v_nat, v_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(g.download(ARR.POSITIONS), ref.download(ARR.POSITIONS), rtol=0, atol=1e-7)
    g.chain_destroy()
