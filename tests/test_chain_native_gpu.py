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


def test_a_received_block_outside_its_zone_raises_a_flag():
    """The grid update of a chain substep looks every zone block up in ITS zone's received buffer (the add folded into
    k_grid<2>); a received list that names a block of this rank's grid outside that zone must not be dropped without a
    trace (k_halo_add2 reported it, the folded add did not).  The left buffer gets the right buffer's ids."""
    import torch
    cut_lo, cut_hi, pitch, zone, cap = 6, 10, 4, 2, 256
    MPM_ERR_CAPACITY_BIT, MPM_ERR_HALO_BIT = 2, 16   # (mpm_device.h: ERR_CAPACITY, ERR_HALO)

    def run(tamper):
        g = _engine()
        nbytes = g.halo_buffer_bytes(cap)
        dev = torch.device("cuda", 0)
        send_l, send_r, recv_l, recv_r = (torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(4))
        zones = [(cut_lo - zone, cut_lo + zone - 1, +pitch), (cut_hi - zone, cut_hi + zone - 1, -pitch)]
        za = g.halo_zone_args(zones, [send_l.data_ptr(), send_r.data_ptr()])
        ra = g.halo_buffer_args([recv_l.data_ptr(), recv_r.data_ptr()])
        stream = torch.cuda.Stream()
        g.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            g.substep_begin_halo(DT, za, cap)
            recv_r.copy_(send_l, non_blocking=True)
            recv_l.copy_(send_r, non_blocking=True)
            if tamper == "zone":      # ids (and count) of the other zone's list: blocks 8..11 in the buffer of zone 4..7
                words = 4 + cap
                recv_l.view(torch.int32)[:words].copy_(recv_r.view(torch.int32)[:words])
            elif tamper == "corrupt":  # an id no sender writes
                recv_l.view(torch.int32)[4] = 0x7FFFFFF0
            g.substep_end_halo(DT, -1, ra, cap)
        try:
            g.gpu_sync()
        except Exception:  # noqa: BLE001  (the sticky flag is what is looked at)
            pass
        flags = g.stats()["error_flags"]
        g.destroy()
        return flags

    assert run(None) == 0
    assert run("zone") & MPM_ERR_CAPACITY_BIT
    assert run("corrupt") & MPM_ERR_HALO_BIT


def test_partitioned_exact_line_search_is_device_resident_over_rccl(monkeypatch):
    """UpdateContact with the exact line search on a partitioned engine whose transport is the native chain (VERDICT r3,
    item 7): the device-resident pattern of the single-GPU solve, with ncclAllReduce on the engine's stream between the two
    halves of every decision and the zone exchange inside the direction -- no host loop per probe.  One GPU is all the box
    has and RCCL refuses two ranks on one device, so the partition has ONE rank (MPM_CT_FORCE_DIST: it takes the
    distributed paths all the same -- split direction kernels, sums through the all-reduce); the result must be the
    single engine's."""
    from drake_amd import ARR, Collider, GpuMpm, scenes
    floor_z = 0.5

    def engine(partitioned):
        if partitioned:
            monkeypatch.setenv("MPM_CT_FORCE_DIST", "1")
        else:
            monkeypatch.delenv("MPM_CT_FORCE_DIST", raising=False)
        g = GpuMpm(BITS)
        g.set_deterministic(True)
        sheets = scenes.cloth_stack(2, 36, BITS, z0=floor_z - 0.004, side=0.4, seed=33, vel_amp=0.3)
        for pos, vel, idx in sheets:
            vel[:, 2] -= 0.5
        scenes.populate(g, sheets)
        if partitioned:
            nb = (1 << BITS) // 4
            g.dist_init(0, 1, [0, nb], 2, 2, 2)
            g.chain_init(GpuMpm.chain_unique_id(), 0, 1, 0, nb, 0, 2, 256)
        return g

    out = []
    for partitioned in (False, True):
        g = engine(partitioned)
        res = []
        for _ in range(3):
            g.reallocate_external_bodies(1)
            g.rebuild_mapping(False)
            g.calc_fem_state_and_force(DT)
            g.particle_to_grid(DT)
            g.update_grid(-1)
            n = g.generate_contact_pairs([Collider(0, body=0, p_WB=(0.5, 0.5, floor_z))])
            r = g.update_contact(DT, 0.5, 1e5, 1e-3, exact_line_search=True)
            res.append((n, r["iterations"], g.contact_stats()["line_search_evals"]))
            g.grid_to_particle(DT)
        g.gpu_sync()
        assert g.stats()["error_flags"] == 0
        out.append((res, g.download(ARR.VELOCITIES), g.download(ARR.POSITIONS)))
        if partitioned:
            g.chain_destroy()
    (res_a, v_a, x_a), (res_b, v_b, x_b) = out
    assert all(n > 20 and it >= 2 for n, it, _ in res_a), res_a
    assert res_a == res_b, (res_a, res_b)     # same contacts, Newton iterations and probes
    np.testing.assert_allclose(v_b, v_a, rtol=0, atol=2e-6)
    np.testing.assert_allclose(x_b, x_a, rtol=0, atol=1e-7)
