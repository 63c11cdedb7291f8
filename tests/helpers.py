"""Shared helpers for the parity tests (oracle vs HIP engine)."""
import numpy as np

from drake_amd import scenes
from oracle import oracle as orc

RTOL = 1e-5  # BASELINE.json north_star: "float state within 1e-5 relative"
MARGINS = []  # (error / allowed, what) of every comparison, reported at the end of the session
# appended to the name of every comparison while a test's smoke variant on the DEFAULT engine runs (particle order inside a
# cell from atomics: its margins differ from run to run; the deterministic variants' do not -- gpurun_out/parity_margins.txt
# of two runs agree on every line without this tag); reset after every test (tests/conftest.py)
TAG = ""


def tag_default_engine(on=True):
    global TAG
    TAG = " [default engine: varies from run to run]" if on else ""


def close(a, b, scale=None, rtol=RTOL, what=""):
    """max|a-b| <= rtol * scale, where scale is the natural magnitude of the
    quantity (max|b| unless the caller knows better, e.g. sums that cancel)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if b.size == 0:
        return
    if scale is None:
        scale = float(np.max(np.abs(b)))
    err = float(np.max(np.abs(a - b)))
    plain = float(np.max(np.abs(b)))
    # (error / allowed, what, tolerance, error relative to the test's scale, error relative to plain max|ref|)
    MARGINS.append((err / (rtol * scale + 1e-30), what + TAG, rtol, err / (scale + 1e-300), err / (plain + 1e-300)))
    assert np.all(np.isfinite(a)), what
    assert err <= rtol * scale + 1e-30, f"{what}: max err {err:.3e} > {rtol:.0e} * scale {scale:.3e}"


# material fields: (engine name, oracle name)
_MATERIAL_FIELDS = dict(youngs_modulus="youngs", poisson_ratio="poisson", density="density", gamma="gamma", K="K", V="V",
                        c_F="cF", sdf_friction="sdf_friction", gravity="gravity", epsv="epsv")

# the bagging demo's variants of the settings.h constants (settings.h:88-111: K = 4e5, V = 0.2, SDF_FRICTION = 1.0)
BAGGING_MATERIAL = dict(K=4e5, V=0.2, sdf_friction=1.0)


def build_pair(domain_bits=6, layers=3, res=20, z0=0.5, side=0.3, seed=7, vel_amp=0.2, material=None, sheets=None,
               deterministic=False):
    """Same cloth stack on the oracle and on the engine.  `material`: dict of mpm_material_t fields
    applied to both sides.  `deterministic`: the engine runs in mpm_set_deterministic mode from Finalize's own first
    sort on (canonical particle order inside every cell, fixed-point tile sums); the oracle's sums have a fixed order
    always (oracle.OracleMpm.ordered_scatter), so a comparison of the two then yields THE SAME NUMBER ON EVERY RUN and
    its bound is a fixed multiple of a fixed yardstick, not a quantile over draws."""
    from drake_amd import GpuMpm

    if sheets is None:
        sheets = scenes.cloth_stack(layers, res, domain_bits, z0=z0, side=side, seed=seed, vel_amp=vel_amp)
    o = orc.OracleMpm(domain_bits)
    gm = None
    if material:
        gm = GpuMpm.default_material()
        for k, v in material.items():
            setattr(gm, k, v)
            setattr(o.p, _MATERIAL_FIELDS[k], v)
    g = GpuMpm(domain_bits, gm)
    if deterministic:
        g.set_deterministic(True)
    for pos, vel, idx in sheets:
        o.add_qr_cloth(pos, vel, idx)
        g.add_qr_cloth(pos, vel, idx)
    o.finalize()
    g.finalize()
    return o, g


def oracle_copy(o, real=np.float64):
    """A second oracle (float build or the double build, oracle/Makefile target f64) started from `o`'s current state."""
    d = orc.OracleMpm(o.domain_bits, params=o.p, real=real)
    for name in ("n_verts", "n_faces", "n_particles", "g_cnt", "finalized", "n_bodies"):
        setattr(d, name, getattr(o, name))
    for name in ("indices", "pids", "index_mappings", "sort_keys", "sort_ids", "g_flags", "g_ids"):
        setattr(d, name, getattr(o, name).copy())
    for name in ("pos", "vel", "vol", "C", "forces", "taus", "F", "DmInv", "g_m", "g_mv", "g_vstar"):
        setattr(d, name, np.ascontiguousarray(getattr(o, name), dtype=real).copy())
    d._contact_grid = False
    if o.n_bodies:
        d.reallocate_external_bodies(o.n_bodies)
    if o.contacts is not None:
        d.copy_contact_pairs(o.contacts)
    return d


def oracle_f64_copy(o):
    """The converged-solve tests use the double build when the float oracle's energy sums stall the line search."""
    return oracle_copy(o, np.float64)


def float_noise_of_a_substep(o, dt, bc=-1):
    """What float rounding alone does to one substep from `o`'s state: max |v32 - v64| over the particles, the float
    and the double build of the oracle advancing copies of the same state (`o` itself is not touched)."""
    a, b = oracle_copy(o, np.float32), oracle_copy(o, np.float64)
    for s in (a, b):
        s.substep(dt, bc)
    return float(np.abs(a.vel.astype(np.float64) - b.vel).max())


# |engine - float oracle| allowed on a one-substep velocity comparison, in units of the MEASURED float noise of that
# substep (max over the particles of |float oracle - double oracle|), where that exceeds 1e-5 of max|v|.  By the triangle
# inequality engine and float oracle are within (1 + k) noises of each other when the engine is within k noises of the
# double build; tests/test_precision_gpu.py REQUIRES k <= 2 field by field on configs 1 and 2 (60k and 1M particles) and
# finds 0.5 - 1.4; on the parity scenes of a few thousand particles the maxima of two noise fields spread further.
# Measured with the correctly rounded default of round 5: the worst scene (64^3, four pin spheres, max|v| = 0.14 m/s) 3.03
# noises, every other one below 2.6 -- hence 4 (rounds 3 - 4, fast math the only arithmetic: 4.3 measured, 6 allowed).  On
# the scenes that move at O(1 m/s) -- the 256^3 scene, the bagging and the material-friction scenes -- the floor is BELOW
# 1e-5 of max|v| and north_star's plain tolerance is what is tested (natural_scales()["floor_decides"]).
NOISE_FLOOR = 4.0
NOISE_FLOOR_FAST_MATH = 6.0


def natural_scales(o, dt=1e-3, bc=-1, noise_floor=NOISE_FLOOR):
    """Magnitudes against which 1e-5 relative is measured.

    Positions: 1.  Velocities: max|v| -- north_star's plain "1e-5 relative" -- wherever float arithmetic can deliver
    that.  Where it cannot: one ulp of the deformation gradient or of a vertex position is dt*E/(rho*dx) (12.8 m/s per
    unit strain at 64^3, dt = 1e-3) times 1e-7 .. 1e-5 of velocity, WHATEVER the velocities are, so on a scene that moves
    at centimetres per second 1e-5 of max|v| is below the distance between a float and a double evaluation of the SAME
    reference code.  That distance is MEASURED on the state at hand (the float and the double build of the oracle advance
    copies of it by one substep) and `noise_floor` times it is the least that engine and float oracle may differ by: by
    the triangle inequality the two are within (1 + k) distances of each other when the engine is within k of the exact
    result; tests/test_precision_gpu.py requires k <= 2 on configs 1 and 2 and finds 0.5 - 1.4.  Examples: 8e-7 m/s on
    config 1 (0.5 ulp(F) * gain: the floor decides there, max|v| = 0.014 m/s), 2.3e-6 m/s on the 256^3 scene moving at
    0.65 m/s (the floor, 4.6e-6 m/s, is below 1e-5 max|v| = 6.5e-6 m/s: the plain tolerance decides).  C (a velocity
    gradient) is measured against 4/dx times the velocity scale; trajectories against max|v| itself (see the tests)."""
    dxinv = float(1 << o.domain_bits)
    noise = float_noise_of_a_substep(o, dt, bc)
    vmax = max(float(np.max(np.abs(o.vel))), 9.8 * dt, noise_floor * noise / RTOL)
    return dict(pos=1.0, vel=vmax, C=4.0 * dxinv * vmax, vol=float(np.max(o.vol)), float_noise_vel=noise,
                plain_vel=float(np.max(np.abs(o.vel))), floor_decides=noise_floor * noise / RTOL > float(np.max(np.abs(o.vel))))


def solve_tolerance(dofs, k_tol=1e-4, iterations=None):
    """Velocity tolerance (m/s, absolute) for comparing two CONVERGED contact solves.  UpdateContact
    stops when sqrt(sum_nodes |Dir|^2) / DoFs <= kTol = 1e-4 (cuda_mpm_solver.cu:236, 567-570), i.e. at
    an RMS remaining Newton step of kTol * sqrt(DoFs) per node.  The damped Jacobi iteration converges
    linearly: with contraction rho per iteration the solution is still |step| / (1 - rho) away when the
    step has shrunk to |step|, and two solvers that stop at slightly different points of that tail
    differ by as much.  rho is taken from the iteration count, rho = kTol^(1/iterations) (residual 1 ->
    kTol): 15 iterations (the soft parameters) give 1/(1-rho) = 2.2, 100 iterations (config 3: k = 1e6,
    mu = 1) give 11.  The factor is that, and not less than 3 (maximum norm against RMS).
    Rounding-level agreement is tested on a single iteration."""
    factor = 3.0
    if iterations:
        rho = k_tol ** (1.0 / max(float(iterations), 1.0))
        factor = max(factor, 1.0 / (1.0 - rho))
    return factor * k_tol * float(np.sqrt(max(float(dofs), 1.0)))


# Per-body impulses of two converged solves: sum_contacts m (v_after - v_before) with every velocity
# good to solve_tolerance (~1e-2 m/s) against velocity changes of ~0.5 m/s, partly cancelling errors
IMPULSE_RTOL = 4e-3
