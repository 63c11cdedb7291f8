"""Contact solve (CopyContactPairs + UpdateContact) on the GPU vs the oracle."""
import numpy as np
import pytest

from tests.helpers import IMPULSE_RTOL

from tests.helpers import build_pair, close, natural_scales, oracle_f64_copy, solve_tolerance

# A converged float oracle may stop up to this factor above the solver's tolerance ("Tiny Alpha" steps); beyond
# that the double-precision build of the same source is the reference (its energy sums do not stall)
MAX_STALL = 3.0


def converged_reference(o, o64, ro, solve64):
    """(reference state, factor by which its residual exceeds kTol) for comparing a converged engine solve"""
    if ro["residual"] <= MAX_STALL * 1e-4:
        return o, max(1.0, ro["residual"] / 1e-4)
    r64 = solve64(o64)
    assert r64["residual"] <= 1e-4, (r64, ro)
    return o64, 1.0


def close_rel(a, b, frac, what, deterministic=True):
    """Relative bounds on top of the solver-tolerance bound (a regression of the Newton path must not
    hide inside the tail tolerance): rms(a - b) <= frac * rms(b), and single entries against max|b|.

    deterministic (engine in mpm_set_deterministic mode, the oracle's sums in fixed order: the comparison is ONE number,
    the same on every run -- VERDICT r5 item 2; profiles/r06_parity_margins_deterministic.txt, two runs
    identical to the digit): NO entry beyond 2 * frac of max|ref| (measured: 0.2 frac on the small scenes, 0.64 frac at
    the one worst contact of the 1M-particle config 3).
    otherwise (the smoke variant on the default engine, whose particle order inside a cell -- and with it the length of
    the noise-limited tail of the solve -- differs from run to run): round 5's bounds, all but a per-mille of the entries
    within 3 * frac and none beyond 8 * frac (measured over several runs of the 1M-particle config 3: 0.9 - 3.5 % rms,
    up to 12 % of max|v| at one contact)."""
    from tests import helpers
    from tests.helpers import MARGINS
    what = what + helpers.TAG
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err, ref = float(np.abs(a - b).max()), float(np.abs(b).max())
    rms_e, rms_b = float(np.sqrt(np.mean((a - b) ** 2))), float(np.sqrt(np.mean(b ** 2)))
    MARGINS.append((rms_e / (frac * rms_b + 1e-300), what + " (rms, relative)", frac, rms_e / (rms_b + 1e-300), err / (ref + 1e-300)))
    assert rms_e <= frac * rms_b, f"{what}: rms error {rms_e:.3e} > {frac:.0%} of rms(ref) {rms_b:.3e}"
    cap = 2.0 if deterministic else 8.0
    MARGINS.append((err / (cap * frac * ref + 1e-300), what + " (largest single entry)",
                    cap * frac, err / (ref + 1e-300), err / (ref + 1e-300)))
    if not deterministic:
        q999 = float(np.quantile(np.abs(a - b), 0.999))
        assert q999 <= 3 * frac * ref, f"{what}: 99.9th percentile {q999:.3e} > {3 * frac:.0%} of max|ref| {ref:.3e}"
    assert err <= cap * frac * ref, f"{what}: {err:.3e} > {cap * frac:.0%} of max|ref| {ref:.3e}"

pytestmark = pytest.mark.gpu
DT = 1e-3
Z_FLOOR = 0.5


def floor_contacts(pos_slot_order):
    """Contact pairs against the half-space z < Z_FLOOR, built like CalcMpmContactPairs
    (deformable_driver.h:120-194): one contact per penetrating particle, normal = -grad(phi)."""
    z = pos_slot_order[:, 2]
    idx = np.nonzero(z < Z_FLOOR)[0].astype(np.uint32)
    n = idx.size
    dist = (z[idx] - Z_FLOOR).astype(np.float32)
    normal = np.tile(np.array([0, 0, -1], np.float32), (n, 1))
    pos = pos_slot_order[idx].astype(np.float32)
    zeros = np.zeros((n, 3), np.float32)
    body = np.zeros(n, np.uint32)
    return idx, body, dist, normal, pos, zeros, zeros.copy()


def _diagnose(g, o):
    """State summary for a failed solve: where do GPU and oracle inputs differ, is anything non-finite."""
    from drake_amd import ARR as A
    out = {}
    for name, arr, ref in (("grid_m", A.GRID_MASSES, o.g_m), ("grid_mv", A.GRID_MOMENTUM, o.g_mv),
                           ("grid_vstar", A.GRID_V_STAR, o.g_vstar), ("vel0", A.CONTACT_VEL0, o.c_vel0),
                           ("vel", A.CONTACT_VEL, o.c_vel)):
        a = g.download(arr)
        out[name] = dict(nonfinite=int((~np.isfinite(a)).sum()), absmax=float(np.nanmax(np.abs(a))) if a.size else 0.0)
        if ref is not None and a.shape == np.asarray(ref).shape:
            out[name]["maxdiff"] = float(np.nanmax(np.abs(a - ref))) if a.size else 0.0
    out["stats"] = g.stats()
    return out


# (stiffness, damping, dt, friction): the first test's historical values, and SURVEY.md 8(d) config 3 =
# the bagging demo's parameters (examples/multibody/deformable/mpm_bagging.cc:9,15-17)
CONTACT_PARAMS = {"soft": (1e5, 1e-3, 1e-3), "config3": (1e6, 1e-5, 2e-4)}


# (deterministic: engine in mpm_set_deterministic mode against the oracle's fixed-order sums -- one number per comparison,
# the same on every run; the last entry is the smoke variant on the default engine with round 5's bounds)
@pytest.mark.parametrize("exact,params,mu,deterministic", [
    (False, "soft", 0.0, True), (True, "soft", 0.0, True), (False, "soft", 0.5, True), (True, "soft", 0.5, True),
    (False, "config3", 1.0, True), (True, "config3", 1.0, True), (False, "config3", 1.0, False)])
def test_update_contact_matches_oracle(exact, params, mu, deterministic):
    from drake_amd import ARR as A
    from oracle import oracle as orc
    from tests import helpers
    helpers.tag_default_engine(not deterministic)
    stiffness, damping, DT = CONTACT_PARAMS[params]
    # two sheets straddling the floor, moving down and sideways
    o, g = build_pair(layers=2, res=20, z0=Z_FLOOR - 0.004, vel_amp=0.3, deterministic=deterministic)
    o.vel[:, 2] -= 0.5
    o.vel[:, 0] += 0.3
    for step in range(3):
        g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        o.reallocate_external_bodies(1)
        g.reallocate_external_bodies(1)
        pos_g = g.sync_particle_state_to_cpu()
        assert np.array_equal(pos_g, o.pos)
        for s in (o, g):
            s.rebuild_mapping(False)
            s.calc_fem_state_and_force(DT)
            s.particle_to_grid(DT)
            s.update_grid(-1)
        cp = floor_contacts(pos_g)
        assert cp[0].size > 50
        o.copy_contact_pairs(orc.ContactPairs(*cp))
        g.copy_contact_pairs(*cp)
        # (iteration limit: the reference's 2000, cuda_mpm_solver.cu:234; 600 for config 3, where a stalled
        # oracle -- see below -- would otherwise spend minutes of CPU time on its last 1400 iterations)
        cap = 600 if params == "config3" else 0
        o64 = oracle_f64_copy(o)
        ro = o.update_contact(DT, mu, stiffness, damping, exact_line_search=exact, max_iters=cap)
        rg = g.update_contact(DT, mu, stiffness, damping, exact_line_search=exact, max_newton_iterations=cap)
        sc = natural_scales(o, DT)
        # Both solves must converge.  Iteration counts: with the soft parameters the two Newton paths stay
        # together to the end.  With config 3's (k = 1e6, mu = 1) the damped Jacobi iteration converges
        # slowly and ends in a noise-limited tail where the line search accepts a step on `E1 <= E0`
        # (cuda_mpm_solver.cu:518), two sums that agree to ~7 digits: the reference (and the oracle) add them
        # in float, the engine in double, and the length of that tail differs by
        # tens of percent -- sometimes a factor of two -- either way from run to run.  There the counts are
        # only required to stay below the iteration limit; the first
        # iteration, where noise plays no role, is compared exactly in the test below.
        slack = max(3, ro["iterations"] // 4)
        if params == "soft":
            assert abs(rg["iterations"] - ro["iterations"]) <= slack, (rg, ro, step, _diagnose(g, o))
        else:
            assert 0 < rg["iterations"] < cap and 0 < ro["iterations"] <= cap, (rg, ro, step, _diagnose(g, o))
        # (the oracle's float sums can stall above the tolerance -- "Tiny Alpha" steps, cuda_mpm_solver.cu:523-526 --
        # until it runs into its iteration limit; how often depends on the OpenMP summation order of the run.
        # Up to MAX_STALL times the tolerance the comparison widens by that factor; beyond it the reference is the
        # double-precision build of the oracle, which converges)
        assert rg["residual"] <= 1e-4
        close(g.download(A.CONTACT_VEL0), o.c_vel0, scale=sc["vel"], what="contact vel0")
        o_float = o
        o, stalled = converged_reference(o_float, o64, ro, lambda d: d.update_contact(
            DT, mu, stiffness, damping, exact_line_search=exact, max_iters=cap or 2000))
        # converged solves agree to the solver's stopping tolerance (rounding-level agreement: the
        # single-iteration test below)
        dofs = g.contact_stats()["dofs"]
        tol = solve_tolerance(dofs, iterations=max(rg["iterations"], ro["iterations"])) * stalled
        imp_rtol = IMPULSE_RTOL * tol / solve_tolerance(dofs)   # (impulses inherit the velocities' tolerance)
        close(g.download(A.CONTACT_VEL), o.c_vel, scale=1.0, rtol=tol, what="contact vel")
        wgt = (o.g_m / o.g_m.max())[:, None]
        close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=1.0, rtol=tol, what="grid v after contact")
        # grid_Dir accessor: the last relaxed Newton direction, non-zero only on nodes that see contacts
        gdir = g.download(A.GRID_DIR)
        assert gdir.shape == (g.n_cells, 3) and np.isfinite(gdir).all() and np.abs(gdir).max() > 0
        assert np.count_nonzero(np.abs(gdir).max(1)) < 0.05 * g.n_cells
        tau_g, f_g = g.external_body_force_to_host()
        fscale = float(np.abs(o.F_f).max())
        close(f_g, o.F_f, scale=fscale, rtol=imp_rtol, what="body impulse")
        close(tau_g, o.F_tau, scale=max(float(np.abs(o.F_tau).max()), fscale), rtol=imp_rtol, what="body angular impulse")
        # whatever the tail tolerance allows, a converged solve is within 4 % rms of the reference's contact velocities
        # (12 % of the largest one at any single contact) and within 2 % (6 %) on the impulses
        close_rel(g.download(A.CONTACT_VEL), o.c_vel, 0.04 * stalled, "contact vel", deterministic)
        close_rel(f_g, o.F_f, 0.02, "body impulse", deterministic)
        # the floor pushes up
        assert f_g[0, 2] < 0  # impulse ON the body is downward
        o.grid_to_particle(DT)
        g.grid_to_particle(DT)
        close(g.download(A.VELOCITIES), o.vel, scale=1.0, rtol=tol, what="vel after contact step")
        if o is not o_float:   # the float oracle carries the trajectory on
            o_float.grid_to_particle(DT)
            o = o_float
    g.gpu_sync()


def test_no_contacts_is_a_noop():
    from drake_amd import ARR as A
    o, g = build_pair(layers=2, res=12)
    g.reallocate_external_bodies(2)
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(DT)
    g.particle_to_grid(DT)
    g.update_grid(-1)
    before = g.download(A.GRID_MOMENTUM)
    e = np.zeros((0, 3), np.float32)
    g.copy_contact_pairs(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32), e, e, e, e)
    r = g.update_contact(DT, 0.5, 1e5, 1e-3)
    assert r["iterations"] == 0
    assert np.array_equal(before, g.download(A.GRID_MOMENTUM))
    tau, f = g.external_body_force_to_host()
    assert not tau.any() and not f.any()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("params,mu,scene", [("soft", 0.5, "sparse"), ("config3", 1.0, "sparse"), ("config3", 1.0, "dense")])
def test_single_newton_iteration_matches_oracle(exact, params, mu, scene):
    """max_newton_iterations = 1 ("one full Newton iteration's Dir / norm_dir"): the
    solver tolerance plays no role, so the direction (cuda_mpm_kernels.cuh:1217-1274), its norm, the
    line-search energies and the chosen step (cuda_mpm_solver.cu:383-528) are compared at rounding level."""
    from drake_amd import ARR as A
    from oracle import oracle as orc
    stiffness, damping, DT = CONTACT_PARAMS[params]
    if scene == "sparse":
        o, g = build_pair(layers=2, res=24, z0=Z_FLOOR - 0.004, vel_amp=0.3)
    else:
        # ~100 contacts per cell (two sheets of ~50 particles per cell each in every cell layer): the
        # contacts of a cell straddle the 64-contact tiles of k_ct_tile, and a tile holds few segments
        o, g = build_pair(layers=4, res=40, side=0.15, z0=Z_FLOOR - 0.02, vel_amp=0.3)
    o.vel[:, 2] -= 0.5
    o.vel[:, 0] += 0.3
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    for s in (o, g):
        s.reallocate_external_bodies(1)
        s.rebuild_mapping(False)
        s.calc_fem_state_and_force(DT)
        s.particle_to_grid(DT)
        s.update_grid(-1)
    cp = floor_contacts(g.sync_particle_state_to_cpu())
    assert cp[0].size > 100
    if scene == "dense":
        cells = np.floor(cp[4] * 64 - 0.5).astype(np.int64)
        _, per_cell = np.unique(cells[:, 0] * 4096 + cells[:, 1] * 64 + cells[:, 2], return_counts=True)
        assert per_cell.max() > 64 and cp[0].size > 5000
    o.copy_contact_pairs(orc.ContactPairs(*cp))
    g.copy_contact_pairs(*cp)
    ro = o.update_contact(DT, mu, stiffness, damping, exact_line_search=exact, max_iters=1)
    rg = g.update_contact(DT, mu, stiffness, damping, exact_line_search=exact, max_newton_iterations=1)
    cs = g.contact_stats()
    assert ro["iterations"] == rg["iterations"] == cs["iterations"] == 1
    assert cs["contacts"] == cp[0].size
    # Newton direction (relaxed by 0.3) on every node, its norm and the DoF count
    gdir = g.download(A.GRID_DIR)
    close(gdir, o.g_D, what="1-iteration Dir")
    assert cs["dofs"] == ro["dofs"] > 0
    assert np.count_nonzero(np.abs(o.g_D).max(1)) == int(ro["dofs"])
    close([cs["norm_dir_sq"]], [ro["norm_dir_sq"]], what="1-iteration |Dir|^2")
    close([rg["residual"]], [ro["residual"]], what="1-iteration residual")
    # line search: energies and the step.  The energies are sums of ~1e3 float terms that the oracle adds
    # in float in contact order (the reference: float atomics) and the engine in double: 1e-5 relative.
    close([cs["E0"]], [ro["E0"]], what="1-iteration E(0)")
    close([cs["energy"]], [ro["E1"]], scale=abs(ro["E0"]), what="1-iteration E(alpha)")
    if exact:
        close([cs["alpha"]], [ro["alpha"]], scale=1.0, rtol=1e-4, what="1-iteration alpha (exact search)")
    else:
        assert cs["alpha"] == ro["alpha"]
        assert cs["line_search_evals"] == ro["ls_last"]
    sc = natural_scales(o, DT)
    rt = 1e-4 if exact else 1e-5     # (the exact search's alpha itself is converged to 1e-8 * 0.3 in float)
    wgt = (o.g_m / o.g_m.max())[:, None]
    close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], rtol=rt, what="1-iteration grid v")
    close(g.download(A.CONTACT_VEL), o.c_vel, scale=sc["vel"], rtol=rt, what="1-iteration contact vel")
    tau_g, f_g = g.external_body_force_to_host()
    close(f_g, o.F_f, scale=float(np.abs(o.F_f).max()), rtol=10 * rt, what="1-iteration body impulse")


@pytest.mark.parametrize("exact,iters", [(False, 5), (True, 5)])
def test_five_newton_iterations_match_oracle(exact, iters):
    """max_newton_iterations = 5 on both sides, config-3 parameters, the dense scene: the
    solver's stopping tolerance still plays no role, but unlike the single iteration this goes through the lazy update
    (k_ct_tile reading v - alpha D, k_ct_node_dir writing it back), the batched / pipelined loop and, for the exact
    search, the device-resident root finder across pattern boundaries (cuda_mpm_solver.cu:274-570)."""
    from drake_amd import ARR as A
    from oracle import oracle as orc
    stiffness, damping, DT = CONTACT_PARAMS["config3"]
    o, g = build_pair(layers=4, res=40, side=0.15, z0=Z_FLOOR - 0.02, vel_amp=0.3)
    o.vel[:, 2] -= 0.5
    o.vel[:, 0] += 0.3
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    for s in (o, g):
        s.reallocate_external_bodies(1)
        s.rebuild_mapping(False)
        s.calc_fem_state_and_force(DT)
        s.particle_to_grid(DT)
        s.update_grid(-1)
    cp = floor_contacts(g.sync_particle_state_to_cpu())
    assert cp[0].size > 5000
    o.copy_contact_pairs(orc.ContactPairs(*cp))
    g.copy_contact_pairs(*cp)
    ro = o.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_iters=iters)
    rg = g.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_newton_iterations=iters)
    cs = g.contact_stats()
    assert ro["iterations"] == rg["iterations"] == cs["iterations"] == iters
    # Backtracking: 1e-4 (measured 2e-6 .. 7e-6 per iteration: the two paths do not drift apart).  Exact search: the
    # root finder works on dE/dalpha, a float sum that is noise below ~1e-6 of its terms; with x_tol = 3e-9 it ends
    # after its 200 evaluations (or earlier, when the noise happens to cross |f'| < 1e-8) wherever the noise leaves it
    # -- on BOTH sides, cuda_mpm_solver.cu:383-471 -- so alpha agrees to ~5e-5 and five iterations compound to ~3e-4
    rt = 1e-3 if exact else 1e-4
    close(g.download(A.GRID_DIR), o.g_D, rtol=rt, what=f"{iters}-iteration Dir")
    assert cs["dofs"] == ro["dofs"] > 0
    close([cs["norm_dir_sq"]], [ro["norm_dir_sq"]], rtol=rt, what=f"{iters}-iteration |Dir|^2")
    close([rg["residual"]], [ro["residual"]], rtol=rt, what=f"{iters}-iteration residual")
    close([cs["E0"]], [ro["E0"]], rtol=rt, what=f"{iters}-iteration E(0)")
    close([cs["energy"]], [ro["E1"]], scale=abs(ro["E0"]), rtol=rt, what=f"{iters}-iteration E(alpha)")
    if exact:
        close([cs["alpha"]], [ro["alpha"]], scale=1.0, rtol=2e-4, what=f"{iters}-iteration alpha (exact search)")
    else:
        assert cs["alpha"] == ro["alpha"]
    sc = natural_scales(o, DT)
    wgt = (o.g_m / o.g_m.max())[:, None]
    close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], rtol=rt, what=f"{iters}-iteration grid v")
    close(g.download(A.CONTACT_VEL), o.c_vel, scale=sc["vel"], rtol=rt, what=f"{iters}-iteration contact vel")
    tau_g, f_g = g.external_body_force_to_host()
    close(f_g, o.F_f, scale=float(np.abs(o.F_f).max()), rtol=10 * rt, what=f"{iters}-iteration body impulse")


def _history_scene(deterministic=True):
    from drake_amd import ARR as A
    from oracle import oracle as orc
    from tests.helpers import oracle_copy
    o, g = build_pair(layers=4, res=40, side=0.15, z0=Z_FLOOR - 0.02, vel_amp=0.3, deterministic=deterministic)
    o.vel[:, 2] -= 0.5
    o.vel[:, 0] += 0.3
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    for s in (o, g):
        s.reallocate_external_bodies(1)
        s.rebuild_mapping(False)
        s.calc_fem_state_and_force(CONTACT_PARAMS["config3"][2])
        s.particle_to_grid(CONTACT_PARAMS["config3"][2])
        s.update_grid(-1)
    cp = floor_contacts(g.sync_particle_state_to_cpu())
    o64 = oracle_copy(o, np.float64)
    for s in (o, o64):
        s.copy_contact_pairs(orc.ContactPairs(*cp))
    g.copy_contact_pairs(*cp)
    return o, o64, g


# Bounds of the 20-iteration history in units of its yardstick (the running maximum of |oracle32 - oracle64|).
# Deterministic mode: engine and oracle are both pure functions of the scene, every ratio below is ONE number, the same on
# every run (gpurun_out/parity_margins.txt of two consecutive runs agree to the digit), so the bound is a fixed multiple
# at EVERY iteration.  The smoke variant on the default engine (particle order inside a cell from atomics: the chaotic
# iteration amplifies that last-bit difference like any other) keeps round 5's three statements.
HISTORY_EARLY = 5.0     # iterations 0 - 5, before rounding has grown: both modes
HISTORY_EVERY = 16.0    # deterministic mode: every iteration (measured, the same on every run: sum |Dir|^2 10.59 and the residual
                        # 9.33 at the one iteration where the float and the double oracle happen to lie close together, E(0)
                        # and E(alpha) 2.87; the exact search 0.8 - 1.06 throughout)
HISTORY_FIELDS = 2.5    # deterministic mode: rms |engine - oracle32| of the fields after 20 iterations, in rms |oracle32 - oracle64|
                        # (measured 1.24 for the grid velocities, 0.26 for the directions)


@pytest.mark.parametrize("exact,deterministic", [(False, True), (True, True), (False, False)])
def test_twenty_newton_iterations_decision_by_decision(exact, deterministic):
    """the contact decisions pinned over the WHOLE iteration history, not at the last step.  The
    reference takes its decisions on the host, one set per Newton iteration (cuda_mpm_solver.cu:472-528 backtracking,
    :383-471 exact search, :567-570 stopping test); the engine takes them on the device and logs them
    (mpm_download_contact_log), the oracle logs the reference's.  Compared row by row over 20 iterations of config 3's
    stiff parameters:
      * backtracking: the accepted step alpha and the number of energy evaluations EXACTLY (alpha is a power of two chosen
        by `E1 <= E0`: one different decision anywhere would show), DoFs exactly;
      * residual, E(0), E(alpha), sum |Dir|^2 to the digits the float and the double build of the ORACLE share: the
        relaxed Jacobi iteration is not contractive over these iterations (a perturbation of the direction field grows
        about threefold per iteration), so the yardstick of iteration i is the largest distance |oracle32 - oracle64| of
        iterations 0 .. i (plus 2e-6 relative).  Deterministic mode (round 6): within HISTORY_EARLY yardsticks in
        iterations 0 - 5 and within HISTORY_EVERY at EVERY iteration -- both sides are pure functions of the scene
        (the oracle's sums have a fixed order, the engine runs in mpm_set_deterministic mode), so these are fixed
        numbers, not draws.  The smoke variant on the default engine keeps round 5's statements (early <= 5, median
        <= 2, no iteration beyond 40: there one of the two sides IS a draw);
      * exact search: alpha is a continuous function of the state, so it drifts apart with the fields (engine and float
        oracle agree to 1e-5 for eight iterations and to nothing after fifteen -- as do the two builds of the oracle): the
        same yardstick, plus 2e-4 (the root finder works on dE/dalpha, a float sum that is noise below ~1e-6 of its terms,
        and ends wherever the noise leaves it after up to 200 evaluations, on both sides); the evaluation count is
        reported, not compared.
    The FIELDS after 20 iterations (directions, grid velocities): rms |engine - oracle32| <= HISTORY_FIELDS rms
    |oracle32 - oracle64| in deterministic mode (8 on the default engine)."""
    from drake_amd import ARR as A
    stiffness, damping, DT = CONTACT_PARAMS["config3"]
    iters = 20
    o, o64, g = _history_scene(deterministic)
    ro = o.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_iters=iters)
    r64 = o64.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_iters=iters)
    rg = g.update_contact(DT, 1.0, stiffness, damping, exact_line_search=exact, max_newton_iterations=iters)
    assert ro["iterations"] == rg["iterations"] == r64["iterations"] == iters
    L32, L64, Lg = o.contact_log, o64.contact_log, g.contact_log().astype(np.float64)
    assert L32.shape == (iters, 7) and Lg.shape == (iters, 8)
    # oracle columns: alpha, E0, E1, nd, dofs, ls, residual; engine: residual, ls, E1, alpha, E0, nd, dofs
    col = dict(alpha=(0, 3), E0=(1, 4), E1=(2, 2), nd=(3, 5), dofs=(4, 6), ls=(5, 1), residual=(6, 0))
    print(f"{'it':>3} {'alpha o32':>10} {'alpha eng':>10} {'ls':>5} {'res o32':>11} {'res eng':>11} {'res o64':>11} {'E1 o32':>13} {'E1 eng':>13}")
    for i in range(iters):
        print(f"{i:3d} {L32[i, 0]:10.3e} {Lg[i, 3]:10.3e} {int(L32[i, 5]):2d}/{int(Lg[i, 1]):2d} {L32[i, 6]:11.4e} {Lg[i, 0]:11.4e} "
              f"{L64[i, 6]:11.4e} {L32[i, 2]:13.6e} {Lg[i, 2]:13.6e}")
    assert np.array_equal(Lg[:, col["dofs"][1]], L32[:, col["dofs"][0]])
    mode = ("deterministic" if deterministic else "default engine") + (", exact" if exact else ", backtracking")

    def bounded(ratio, what):
        """the statement of this mode about a history of ratios; returns error / allowed for the margins table"""
        early, med, peak = float(ratio[:6].max()), float(np.median(ratio)), float(ratio.max())
        print(f"20-iteration history ({mode}) {what}: early {early:.3f}, median {med:.3f}, peak {peak:.3f} yardsticks")
        if deterministic:
            assert early <= HISTORY_EARLY and peak <= HISTORY_EVERY, (what, early, peak, ratio)
            return max(early / HISTORY_EARLY, peak / HISTORY_EVERY)
        assert early <= 5.0 and med <= 2.0 and peak <= 40.0, (what, early, med, peak, ratio)
        return max(early / 5.0, med / 2.0, peak / 40.0)

    # the yardstick of iteration i: the largest |oracle32 - oracle64| of iterations 0 .. i (the distance grows with the
    # iterations, and at a single iteration the two may happen to coincide)
    runmax = lambda x: np.maximum.accumulate(np.abs(x))
    upto = iters
    if exact:
        # the exact search is pinned while the float and the double oracle themselves still agree on alpha to 2 % (at
        # least the first eight iterations); beyond that both sides' alphas are the noise of the direction fields
        tol_a = 5.0 * runmax(L32[:, 0] - L64[:, 0]) + 2e-4
        upto = int(np.argmax(tol_a > 0.1)) if np.any(tol_a > 0.1) else iters
        assert upto >= 8, tol_a
        ratio_a = (np.abs(Lg[:, 3] - L32[:, 0]) / (runmax(L32[:, 0] - L64[:, 0]) + 4e-5))[:upto]
        bounded(ratio_a, "alpha")
    else:
        assert np.array_equal(Lg[:, 3], L32[:, 0]), (Lg[:, 3], L32[:, 0])          # every accepted step
        assert np.array_equal(Lg[:, 1], L32[:, 5])                                  # every evaluation count
        assert np.array_equal(L32[:, 0], L64[:, 0].astype(np.float32))              # (the premise: the double build decides alike)
    from tests.helpers import MARGINS
    for name in ("residual", "E0", "E1", "nd"):
        a, b, c = Lg[:, col[name][1]], L32[:, col[name][0]], L64[:, col[name][0]]
        yard = runmax(b - c) + 0.2 * (2e-4 if exact else 1e-5) * np.abs(b) + 1e-30
        worst = bounded((np.abs(a - b) / yard)[:upto], name)
        MARGINS.append((worst, f"20-iteration history ({mode}): {name}", HISTORY_EVERY if deterministic else 40.0, worst,
                        float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30)))))
    # the last row is what the call itself reports
    cs = g.contact_stats()
    assert cs["alpha"] == np.float32(Lg[-1, 3]) and rg["residual"] == np.float32(Lg[-1, 0])
    # fields: root mean squares against the float noise of the iteration
    wgt = (o.g_m / o.g_m.max())[:, None]
    rms = lambda x: float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))
    scale_D = float(np.abs(o.g_D).max())
    noise_D, noise_v = rms(o64.g_D - o.g_D), rms((o64.g_mv - o.g_mv) * wgt)
    err_D, err_v = rms(g.download(A.GRID_DIR) - o.g_D), rms((g.download(A.GRID_MOMENTUM) - o.g_mv) * wgt)
    print(f"20 iterations ({mode}): |Dir| {scale_D:.3g}; rms float vs double oracle: Dir {noise_D:.2e}, "
          f"grid v {noise_v:.2e}; rms engine vs float oracle: Dir {err_D:.2e} ({err_D / noise_D:.2f} x), grid v {err_v:.2e} ({err_v / noise_v:.2f} x)")
    if not exact:
        assert noise_D > 1e-4 * scale_D          # (the premise: rounding alone has grown this far)
        k = HISTORY_FIELDS if deterministic else 8.0
        MARGINS.append((max(err_D / noise_D, err_v / noise_v) / k, f"20-iteration fields ({mode})", k, err_D / noise_D, err_v / noise_v))
        assert err_D <= k * noise_D and err_v <= k * noise_v, (err_D, noise_D, err_v, noise_v)


@pytest.mark.parametrize("deterministic", [True, False])
def test_config3_full_size_against_the_oracle(deterministic):
    """BASELINE config 3 at its full size: the 1M-particle stack (999,952 particles, 128^3) with its lowest
    sheets below a floor, the bagging demo's contact parameters (k = 1e6, d = 1e-5, mu = 1, dt = 2e-4).
    One Newton iteration against the oracle at rounding level, then the converged solve.
    deterministic: engine in mpm_set_deterministic mode against the oracle's fixed-order sums -- every margin is the
    same number on every run, single contacts within 2 x 4 % of max|v| (close_rel); the other variant is the smoke run
    on the default engine with round 5's bounds."""
    from drake_amd import ARR as A, scenes
    from oracle import oracle as orc
    from tests import helpers
    helpers.tag_default_engine(not deterministic)
    stiffness, damping, DT = CONTACT_PARAMS["config3"]
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    sheets = scenes.cloth_stack(layers, res, bits, z0=Z_FLOOR - 0.006, vel_amp=0.2)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
        vel[:, 0] += 0.3
    o, g = build_pair(sheets=sheets, domain_bits=bits, deterministic=deterministic)
    assert g.n_particles == 999952

    def prepare():
        g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        for s in (o, g):
            s.reallocate_external_bodies(1)
            s.rebuild_mapping(False)
            s.calc_fem_state_and_force(DT)
            s.particle_to_grid(DT)
            s.update_grid(-1)
        cp = floor_contacts(g.sync_particle_state_to_cpu())
        o.copy_contact_pairs(orc.ContactPairs(*cp))
        g.copy_contact_pairs(*cp)
        return cp

    cp = prepare()
    assert cp[0].size > 50000
    # one iteration: direction, its norm, energies, step
    ro = o.update_contact(DT, 1.0, stiffness, damping, max_iters=1)
    rg = g.update_contact(DT, 1.0, stiffness, damping, max_newton_iterations=1)
    cs = g.contact_stats()
    assert ro["iterations"] == rg["iterations"] == 1 and cs["dofs"] == ro["dofs"] > 10000
    close(g.download(A.GRID_DIR), o.g_D, what="1m 1-iteration Dir")
    # (global sums over ~2e4 nodes / 6e4 contacts: the oracle adds them in float like the reference's
    # atomics, the engine in double -- the per-node direction above is the 1e-5 comparison)
    close([cs["norm_dir_sq"]], [ro["norm_dir_sq"]], rtol=1e-4, what="1m 1-iteration |Dir|^2")
    close([cs["E0"]], [ro["E0"]], rtol=1e-4, what="1m 1-iteration E(0)")
    close([cs["energy"]], [ro["E1"]], scale=abs(ro["E0"]), rtol=1e-4, what="1m 1-iteration E(alpha)")
    assert cs["alpha"] == ro["alpha"]
    sc = natural_scales(o, DT)
    wgt = (o.g_m / o.g_m.max())[:, None]
    close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], what="1m 1-iteration grid v")
    # the converged solve from the same state
    cp = prepare()
    o64 = oracle_f64_copy(o)
    ro = o.update_contact(DT, 1.0, stiffness, damping, max_iters=600)
    rg = g.update_contact(DT, 1.0, stiffness, damping, max_newton_iterations=600)
    assert rg["residual"] <= 1e-4, (rg, ro)
    o, stalled = converged_reference(o, o64, ro, lambda d: d.update_contact(DT, 1.0, stiffness, damping, max_iters=600))
    tol = solve_tolerance(g.contact_stats()["dofs"], iterations=max(rg["iterations"], ro["iterations"])) * stalled
    print(f"1m converged solve: oracle {ro['iterations']} iterations, residual {ro['residual']:.3e} (stall factor {stalled:.2f}); "
          f"engine {rg['iterations']} iterations, residual {rg['residual']:.3e}")
    close(g.download(A.CONTACT_VEL), o.c_vel, scale=1.0, rtol=tol, what="1m contact vel")
    # (like `tol` above: an oracle that stalled above the solver's tolerance -- up to MAX_STALL times -- stopped that much
    # further from the solution; 2.1 - 3.5 % rms over fourteen runs of the suite)
    close_rel(g.download(A.CONTACT_VEL), o.c_vel, 0.04 * stalled, "1m contact vel", deterministic)
    tau_g, f_g = g.external_body_force_to_host()
    assert f_g[0, 2] < 0
    close(f_g, o.F_f, scale=float(np.abs(o.F_f).max()), rtol=IMPULSE_RTOL * tol / solve_tolerance(g.contact_stats()["dofs"]),
          what="1m body impulse")
    close_rel(f_g, o.F_f, 0.02, "1m body impulse", deterministic)
    g.gpu_sync()
    assert g.stats()["error_flags"] == 0


def test_backtracking_beyond_the_first_candidates_matches_oracle(monkeypatch):
    """The engine evaluates the step lengths 1, 1/2, 1/4, 1/8 in a first pass and the other 24 only when none
    of them is accepted (cuda_mpm_solver.cu:472-528 halves alpha one evaluation at a time).  With the Jacobi
    relaxation raised from 0.3 to 40 (test hooks on both sides) the Newton step overshoots ~40-fold, so the
    accepted step lies beyond the first pass: same step, same count of evaluations, same energies."""
    from drake_amd import ARR as A
    from oracle import oracle as orc
    stiffness, damping, DT = CONTACT_PARAMS["soft"]
    monkeypatch.setenv("MPM_CT_RELAX", "40")   # (read per handle, when the engine is created)
    o, g = build_pair(layers=2, res=24, z0=Z_FLOOR - 0.004, vel_amp=0.3)
    o.vel[:, 2] -= 0.5
    g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
    for s in (o, g):
        s.reallocate_external_bodies(1)
        s.rebuild_mapping(False)
        s.calc_fem_state_and_force(DT)
        s.particle_to_grid(DT)
        s.update_grid(-1)
    cp = floor_contacts(g.sync_particle_state_to_cpu())
    o.copy_contact_pairs(orc.ContactPairs(*cp))
    g.copy_contact_pairs(*cp)
    o.set_contact_relax(40.0)
    try:
        ro = o.update_contact(DT, 0.5, stiffness, damping, exact_line_search=False, max_iters=1)
        rg = g.update_contact(DT, 0.5, stiffness, damping, exact_line_search=False, max_newton_iterations=1)
    finally:
        o.set_contact_relax(0.3)
    cs = g.contact_stats()
    assert ro["iterations"] == rg["iterations"] == 1
    assert ro["alpha"] < 1.0 / 8.0, ro            # beyond the first pass
    assert cs["alpha"] == ro["alpha"] and cs["line_search_evals"] == ro["ls_last"], (cs, ro)
    close([cs["E0"]], [ro["E0"]], what="deep backtracking E(0)")
    close([cs["energy"]], [ro["E1"]], scale=abs(ro["E0"]), what="deep backtracking E(alpha)")
    close(g.download(A.GRID_DIR), o.g_D, what="deep backtracking Dir")
    sc = natural_scales(o, DT)
    wgt = (o.g_m / o.g_m.max())[:, None]
    close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], what="deep backtracking grid v")


def test_pairs_survive_a_resort_between_copy_and_update():
    """CopyContactPairs names particles by the caller's slot; the engine's own particle order changes
    with every internal re-sort.  A RebuildMapping that really re-sorts between CopyContactPairs and
    UpdateContact must not change the solve."""
    from drake_amd import ARR as A

    def prepared(seed=5):
        _, g = build_pair(layers=2, res=24, z0=Z_FLOOR - 0.004, vel_amp=0.3, seed=seed)
        v = g.download(A.VELOCITIES)
        v[:, 2] -= 0.5
        g.upload_particle_state(None, v)
        g.reallocate_external_bodies(1)
        for _ in range(2):
            g.substep(DT, -1)
        return g

    def grid(g):
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(DT)
        g.particle_to_grid(DT)
        g.update_grid(-1)

    a = prepared()
    cp = floor_contacts(a.sync_particle_state_to_cpu())
    assert cp[0].size > 100
    a.copy_contact_pairs(*cp)
    grid(a)
    ra = a.update_contact(DT, 0.5, 1e5, 1e-3)
    b = prepared()
    b.copy_contact_pairs(*cp)
    n0 = b.stats()["rebuilds"]
    b.upload_particle_state(b.sync_particle_state_to_cpu())   # same positions: only raises the re-sort flag
    grid(b)                                                   # this RebuildMapping re-sorts: every internal slot changes
    assert b.stats()["rebuilds"] == n0 + 1
    rb = b.update_contact(DT, 0.5, 1e5, 1e-3)
    assert abs(ra["iterations"] - rb["iterations"]) <= 2
    tol = solve_tolerance(a.contact_stats()["dofs"])
    close(b.download(A.CONTACT_VEL), a.download(A.CONTACT_VEL), scale=1.0, rtol=tol, what="contact vel across a re-sort")
    close(b.download(A.CONTACT_VEL0), a.download(A.CONTACT_VEL0), scale=1.0, rtol=1e-5, what="contact vel0 across a re-sort")
    fa, fb = a.external_body_force_to_host()[1], b.external_body_force_to_host()[1]
    close(fb, fa, scale=float(np.abs(fa).max()), rtol=IMPULSE_RTOL, what="body impulse across a re-sort")
