"""BASELINE.json's configurations at their own sizes and grids: config 1 (plumbing_64k, 64^3) against
the oracle at full size, config 4's grid (256^3) against the oracle at a size the oracle finishes in
seconds and through size-independent properties at the full 8M particles, and the time-step limit
of that grid pinned as an error."""
import numpy as np
import pytest

from tests.helpers import build_pair, close, natural_scales

pytestmark = pytest.mark.gpu


def _phase_by_phase(o, g, bc, dt, steps, tag):
    from drake_amd import ARR as A
    for _ in range(steps):
        sc = natural_scales(o, dt, bc)
        g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        o.rebuild_mapping(False)
        g.rebuild_mapping(False)
        assert np.array_equal(g.download(A.SORT_KEYS), o.sort_keys)
        o.calc_fem_state_and_force(dt)
        g.calc_fem_state_and_force(dt)
        close(g.download(A.DEFORMATION_GRADIENTS), o.F, scale=1.0, what=tag + "F")
        s_tau = max(float(np.abs(o.taus).max()), sc["vol"] * 4e5)
        close(g.download(A.TAUS), o.taus, scale=s_tau, what=tag + "taus")
        close(g.download(A.FORCES), o.forces, scale=max(float(np.abs(o.forces).max()), s_tau * (1 << o.domain_bits)),
              what=tag + "forces")
        o.particle_to_grid(dt)
        g.particle_to_grid(dt)
        close(g.download(A.GRID_MASSES), o.g_m, what=tag + "grid mass")
        close(g.download(A.GRID_MOMENTUM), o.g_mv, scale=sc["vel"] * float(o.g_m.max()), what=tag + "grid mv")
        assert np.array_equal(g.download(A.GRID_TOUCHED_FLAGS), o.g_flags)
        o.update_grid(bc)
        g.update_grid(bc)
        assert g.grid_touched_cnt() == o.g_cnt
        assert np.array_equal(g.download(A.GRID_TOUCHED_IDS), o.touched_blocks())
        wgt = (o.g_m / o.g_m.max())[:, None]
        close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], what=tag + "grid v")
        close(g.download(A.GRID_V_STAR) * wgt, o.g_vstar * wgt, scale=sc["vel"], what=tag + "grid v*")
        o.grid_to_particle(dt)
        g.grid_to_particle(dt)
        close(g.download(A.POSITIONS), o.pos, scale=1.0, what=tag + "pos")
        close(g.download(A.VELOCITIES), o.vel, scale=sc["vel"], what=tag + "vel")
        close(g.download(A.AFFINE), o.C, scale=4.0 * (1 << o.domain_bits) * sc["vel"], what=tag + "C")
    g.gpu_sync()


@pytest.mark.parametrize("bc", [-1, 0, 1, 2, 3])
def test_phase_by_phase_on_the_256_grid(bc):
    """BASELINE config 4's grid (domain_bits = 8, dx = 1/256) at config 4's substep (2e-4): every
    phase against the oracle, all five boundary scenes (their colliders are in world coordinates)."""
    z0 = {-1: 0.5, 0: 0.56, 1: 0.775, 2: 0.11, 3: 0.5}[bc]
    side = {-1: 0.16, 0: 0.16, 1: 0.3, 2: 0.16, 3: 0.42}[bc]
    res = {-1: 40, 0: 40, 1: 60, 2: 40, 3: 84}[bc]
    o, g = build_pair(domain_bits=8, layers=3, res=res, z0=z0, side=side, vel_amp=0.3)
    o.vel[:, 2] -= 0.5
    _phase_by_phase(o, g, bc, 2e-4, 2, "256^3 ")
    if bc in (1, 3):   # the fixed colliders were reached: some massive nodes are at rest
        assert np.count_nonzero((o.g_m > 0) & np.all(o.g_vstar == 0, axis=1)) > 0


def test_plumbing_64k_full_size_against_the_oracle():
    """BASELINE config 1 / SURVEY.md 8(d): 63,248 particles on 64^3, 10 substeps, engine vs oracle:
    one phase-by-phase substep at full size, then the free-running trajectory."""
    from drake_amd import ARR as A, scenes
    bits, layers, res = scenes.CONFIGS["plumbing_64k"]
    dt = 1e-3
    o, g = build_pair(sheets=scenes.cloth_stack(layers, res, bits), domain_bits=bits)
    assert o.n_particles == 63248 == g.n_particles
    _phase_by_phase(o, g, -1, dt, 1, "64k ")
    o, g = build_pair(sheets=scenes.cloth_stack(layers, res, bits), domain_bits=bits)
    for _ in range(10):
        o.substep(dt, -1)
        g.substep(dt, -1)
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0
    pid = g.download(A.PIDS)
    assert np.array_equal(pid, o.pids)    # nobody sorted the slot order
    close(g.download(A.POSITIONS), o.pos, scale=1.0, what="64k traj pos")
    # Ten free-running substeps of a stiff cloth amplify the per-step rounding differences.  Measured with the oracle
    # in float and in double (tests/test_precision_gpu.py, same scene): after 10 substeps the float oracle is 2.0e-4 of
    # max|v| and 3.6e-5 in F away from the double one, the engine likewise; the two float results may differ by the
    # sum, and the tolerances are 3x the measured distance, against the plain maximum of the field
    close(g.download(A.VELOCITIES), o.vel, rtol=6e-4, what="64k traj vel")
    close(g.download(A.DEFORMATION_GRADIENTS), o.F, scale=1.0, rtol=1.1e-4, what="64k traj F")
    o.rebuild_mapping(False)
    assert (g.download(A.SORT_KEYS) == o.sort_keys).mean() > 0.999
    assert np.array_equal(g.download(A.GRID_TOUCHED_FLAGS), o.g_flags)


def test_cloth_1m_full_size_against_the_oracle():
    """BASELINE config 2, the headline workload (999,952 particles, 128^3, dt = 1e-3): one substep phase by
    phase against the oracle at full size (sort keys, F, tau, forces, grid masses / momenta / touched set,
    v, v*, positions, velocities, C), then four batched substeps (mpm_run_substeps: gated re-sort launches,
    vertex forces inside k_p2g) against the oracle's."""
    from drake_amd import ARR as A, scenes
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    dt = 1e-3
    o, g = build_pair(sheets=scenes.cloth_stack(layers, res, bits), domain_bits=bits)
    assert o.n_particles == 999952 == g.n_particles
    _phase_by_phase(o, g, -1, dt, 1, "1m ")
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    g.run_substeps(4, dt, -1)
    for _ in range(4):
        o.substep(dt, -1)
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0
    close(g.download(A.POSITIONS), o.pos, scale=1.0, what="1m traj pos")
    # (as for config 1; measured after 4 substeps on this scene: float oracle vs double oracle 1.2e-3 of max|v| and
    # 5.1e-5 in F -- the cloth starts with velocities of 0.01 m/s, the rounding of F alone moves them by that much;
    # 3x the measured distance, against the plain maximum of the field)
    close(g.download(A.VELOCITIES), o.vel, rtol=3.6e-3, what="1m traj vel")
    close(g.download(A.DEFORMATION_GRADIENTS), o.F, scale=1.0, rtol=1.5e-4, what="1m traj F")


def test_cloth_8m_on_256_grid_properties():
    """BASELINE config 4 at its full single-GPU size (8,036,544 particles, 256^3, dt = 2e-4), which the
    oracle is not run on: mass bookkeeping, free fall, momentum = mass * g * t, no error flags."""
    from drake_amd import ARR as A, GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS["cloth_8m"]
    dt = 2e-4
    g = GpuMpm(bits)
    scenes.populate(g, scenes.cloth_stack(layers, res, bits, z0=0.6, vel_amp=0.0, jitter=0.0))
    assert g.n_particles == 8036544
    vol = g.download(A.VOLUMES).astype(np.float64)
    mass = float(vol.sum() * 2000.0)
    n = 24
    g.run_substeps(n - 1, dt, -1)
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(dt)
    g.particle_to_grid(dt)
    m = g.download(A.GRID_MASSES).astype(np.float64)
    mv = g.download(A.GRID_MOMENTUM).astype(np.float64)
    g.update_grid(-1)
    g.grid_to_particle(dt)
    g.gpu_sync()
    st = g.stats()
    assert st["error_flags"] == 0 and st["substeps"] == n
    assert abs(m.sum() - mass) <= 2e-5 * mass
    v = g.download(A.VELOCITIES)
    np.testing.assert_allclose(v[:, 2], -9.8 * dt * n, rtol=5e-5)
    assert np.max(np.abs(v[:, :2])) < 1e-4
    # grid momentum of the last scatter = particle momentum before it + that step's gravity impulse
    expect = mass * (-9.8 * dt * n)
    assert abs(mv[:, 2].sum() - expect) <= 1e-4 * abs(expect)
    assert abs(mv[:, 0].sum()) + abs(mv[:, 1].sum()) <= 1e-5 * abs(expect)
    # the fixed-point tile sums carry no drift: total momentum equals the float64 sum over particles
    pm = (vol * 2000.0 * (-9.8 * dt * (n - 1))).sum() + mass * (-9.8 * dt)
    assert abs(mv[:, 2].sum() - pm) <= 2e-5 * abs(pm)


def test_time_step_limit_of_the_256_grid_is_reported():
    """dx = 1/256 puts the elastic CFL limit (dx / sqrt(E / rho) = 2.8e-4 s) below dt = 1e-3: the state
    diverges.  The engine must say so instead of producing numbers; the same cloth at the
    configuration's dt = 2e-4 runs clean."""
    from drake_amd import GpuMpm, MpmError, scenes

    def run(dt, n):
        g = GpuMpm(8)
        scenes.populate(g, scenes.cloth_stack(6, 96, 8, z0=0.6, side=0.3, vel_amp=0.3))
        for _ in range(n // 20):
            g.run_substeps(20, dt, -1)
            g.gpu_sync()
        return g

    g = run(2e-4, 200)
    assert g.stats()["error_flags"] == 0
    with pytest.raises(MpmError) as ei:
        run(1e-3, 400)
    # MPM_ERR_DRIFT, or whichever consequence of the blow-up is detected first: particles leaving the grid,
    # non-finite node sums, particles scattered over more blocks than the tables hold
    assert ei.value.code in (-3, -6, -7, -4), ei.value


def test_config5_scale_many_bodies_properties():
    """BASELINE config 5 at its full size on one GPU (4,018,272 particles, 256^3, dt = 2e-4; the
    configuration's 4 GPUs split this domain): a floor and 16 moving capsule "links" with prescribed
    rigid velocities, pairs made on the device, UpdateContact with the bagging parameters.  The oracle is
    not run at this size; checked are size-independent properties: every body gets contacts, the solve
    converges, each body is pushed away from the cloth (impulse against its relative approach) by no more than the
    momentum the solve took out of the grid, nothing is flagged."""
    from drake_amd import ARR as A, Collider, GpuMpm, scenes
    bits, layers, res = scenes.CONFIGS["cloth_4m"]
    dt, k, d, mu = 2e-4, 1e6, 1e-5, 1.0
    g = GpuMpm(bits)
    z0 = 0.5
    sheets = scenes.cloth_stack(layers, res, bits, z0=z0, vel_amp=0.05)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.3
    scenes.populate(g, sheets)
    assert g.n_particles == 4018272
    bodies = [Collider(0, body=0, p_WB=(0.5, 0.5, z0 + 0.0005))]          # the world: a floor the lowest sheet has sunk into
    rot_y = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32)      # capsule axis along x
    for i in range(16):
        y = 0.28 + 0.44 * (i % 8) / 7.0
        top = i >= 8
        z = z0 + (0.034 if top else -0.004)                                # above the 16 sheets (dx/2 apart) / under them
        vz = -0.6 if top else 0.5
        bodies.append(Collider(3, body=1 + i, p_WB=(0.35 + 0.3 * (i % 2), y, z), R_WB=rot_y, dims=(0.006, 0.05, 0),
                               v=(0.1 * (-1) ** i, 0.0, vz), w=(0, 0, 0.5)))
    g.reallocate_external_bodies(17)
    for step in range(2):
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(dt)
        g.particle_to_grid(dt)
        g.update_grid(-1)
        mv0 = g.download(A.GRID_MOMENTUM).astype(np.float64)               # v* per node
        m = g.download(A.GRID_MASSES).astype(np.float64)
        n = g.generate_contact_pairs(bodies)
        pairs = g.download_contact_pairs()
        assert n > 5000
        counts = np.bincount(pairs[1], minlength=17)
        assert np.all(counts > 0), counts
        g.reallocate_external_bodies(17)                                    # per-substep accumulators for the check below
        r = g.update_contact(dt, mu, k, d)
        cs = g.contact_stats()
        assert 0 < r["iterations"] < 500 and r["residual"] <= 1e-4, (r, cs)
        tau, f = g.external_body_force_to_host()
        assert np.isfinite(tau).all() and np.isfinite(f).all()
        # the floor is pushed down, bodies coming from below are pushed down, bodies from above up
        assert f[0, 2] < 0
        assert np.all(f[1:9, 2] < 0) and np.all(f[9:, 2] > 0), f[:, 2]
        # the solve moved the grid: nodes under the stack upwards, and the per-body impulse (mass of the
        # contacting particles times their velocity change, cuda_mpm_kernels.cuh:1616-1658) is bounded by
        # the momentum change of all the mass on those nodes
        mv1 = g.download(A.GRID_MOMENTUM).astype(np.float64)
        dpz = np.abs((mv1[:, 2] - mv0[:, 2]) * m).sum()
        assert dpz > 0 and np.abs(f[:, 2].astype(np.float64)).sum() <= 1.05 * dpz, (dpz, f[:, 2])
        g.grid_to_particle(dt)
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0
    v = g.download(A.VELOCITIES)
    assert np.isfinite(v).all() and np.abs(v).max() < 20.0
