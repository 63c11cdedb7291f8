"""The arithmetic of CalcFemStateAndForce is the caller's choice, and the strict one is the default.

k_fem's 19 divisions and 10 square roots per face are correctly rounded by default (mpm_math.h, FM = 0: what the
reference's expressions say); mpm_set_fast_math(h, 1) selects the hardware approximations plus one Newton step (FM = 1:
2.4 us per substep at 1M particles, what rounds 3 - 4 shipped; the reference itself is compiled with -use_fast_math,
tools/skylark/cuda.bzl:66-80).  Both run here, in one process, on the same states:

* the 256^3 parity scene (0.8 m/s): the default is within the PLAIN 1e-5 of max|v| of the float oracle after one
  substep -- no noise floor; the fast path within 2e-5: the trade, in numbers;
* config 1 as released (max|v| = 0.014 m/s): here 1e-5 of max|v| is 1.4e-7 m/s, a sixth of what the float and the
  double build of the ORACLE differ by (8e-7 m/s: one ulp of F is dt E / (rho dx) x 1.2e-7 of velocity whatever the
  velocities are) -- NEITHER arithmetic can meet it, and none could: both must be within 2 (default) / 3 (fast) of
  that noise;
* config 2: the two differ by at most twice the float noise of the same substep: the fast math is a trade inside the
  rounding noise, not an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _one_substep_errors(fast):
    """velocity error of one substep against the float oracle relative to the PLAIN max|v|, and in units of the float
    noise of that substep, on the 256^3 parity scene moving at 0.8 m/s and on config 1 as released"""
    from drake_amd import ARR as A, scenes
    from tests.helpers import build_pair, float_noise_of_a_substep
    res = {}
    c1 = scenes.CONFIGS["plumbing_64k"]
    for tag, dt, make in (("256^3 scene", 2e-4, lambda: build_pair(domain_bits=8, layers=3, res=40, z0=0.5, side=0.16, vel_amp=0.3)),
                          ("config 1", 1e-3, lambda: build_pair(sheets=scenes.cloth_stack(*c1[1:], c1[0]), domain_bits=c1[0]))):
        o, g = make()
        assert g.fast_math is False     # the default
        g.set_fast_math(fast)
        assert g.fast_math is bool(fast)
        if tag.startswith("256"):
            o.vel[:, 2] -= 0.5
            g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        noise = float_noise_of_a_substep(o, dt)
        o.substep(dt, -1)
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(dt); g.particle_to_grid(dt); g.update_grid(-1); g.grid_to_particle(dt)
        g.gpu_sync()
        err = float(np.abs(g.download(A.VELOCITIES).astype(np.float64) - o.vel).max())
        eF = float(np.abs(g.download(A.DEFORMATION_GRADIENTS).astype(np.float64) - o.F).max())
        res[tag] = dict(vel_rel_plain=err / float(np.abs(o.vel).max()), vel_in_noise=err / noise, F_abs=eF,
                        max_abs_vel=float(np.abs(o.vel).max()), float_noise=noise)
        g.destroy()
    return res


def test_the_default_meets_the_plain_1e5_and_the_fast_math_stays_inside_the_float_noise():
    from drake_amd import ARR as A, GpuMpm, scenes
    from oracle import oracle as orc
    from tests.helpers import float_noise_of_a_substep
    ieee, fast = _one_substep_errors(False), _one_substep_errors(True)
    for tag in ieee:
        print(f"{tag}: one substep, |v - float oracle| / max|v|: default (correctly rounded) {ieee[tag]['vel_rel_plain']:.2e}, "
              f"fast math {fast[tag]['vel_rel_plain']:.2e}; in units of the float noise ({ieee[tag]['float_noise']:.1e} m/s): "
              f"{ieee[tag]['vel_in_noise']:.2f} / {fast[tag]['vel_in_noise']:.2f}; max|v| {ieee[tag]['max_abs_vel']:.3g} m/s")
    assert ieee["256^3 scene"]["vel_rel_plain"] <= 1e-5, ieee
    assert fast["256^3 scene"]["vel_rel_plain"] <= 2e-5, fast
    assert ieee["config 1"]["vel_in_noise"] <= 2.0 and fast["config 1"]["vel_in_noise"] <= 3.0, (ieee, fast)
    # config 2: one substep from the initial state in both arithmetics
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    sheets = scenes.cloth_stack(layers, res, bits)
    out = {}
    for mode in (False, True):
        g = GpuMpm(bits)
        scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
        g.set_fast_math(mode)
        g.substep(1e-3, -1)
        g.gpu_sync()
        out[mode] = (g.download(A.VELOCITIES).astype(np.float64), g.download(A.DEFORMATION_GRADIENTS).astype(np.float64))
        g.destroy()
    o = orc.OracleMpm(bits)
    for pos, vel, idx in sheets:
        o.add_qr_cloth(pos, vel, idx)
    o.finalize()
    noise = float_noise_of_a_substep(o, 1e-3)
    d_v = float(np.abs(out[True][0] - out[False][0]).max())
    d_F = float(np.abs(out[True][1] - out[False][1]).max())
    print(f"config 2: |fast - default| velocities {d_v:.3e} m/s, F {d_F:.3e}; float noise of the substep {noise:.3e} m/s")
    assert d_v > 0.0                  # (the switch does switch)
    assert d_v <= 2.0 * noise, (d_v, noise)
    assert d_F <= 10 * 1.2e-7, d_F    # a few ulp of F (|F| ~ 1)


def test_the_switch_may_change_between_substeps():
    """mpm_set_fast_math between two batches of substeps (it settles first: substeps that mpm_run_substeps deferred run
    with the arithmetic they were enqueued with): the run is reproducible -- in deterministic mode to the bit -- and differs
    from a run that never switched."""
    import os
    from drake_amd import ARR as A, GpuMpm, scenes
    sheets = scenes.cloth_stack(3, 24, 6, z0=0.5, side=0.3, seed=11, vel_amp=0.3)

    def run(plan):
        os.environ["MPM_DETERMINISTIC"] = "1"     # (read at mpm_create: Finalize's own first sort is canonical too)
        try:
            g = GpuMpm(6)
        finally:
            del os.environ["MPM_DETERMINISTIC"]
        scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
        for mode, n in plan:
            g.set_fast_math(mode)
            g.run_substeps(n, 1e-3, -1)
        g.gpu_sync()
        r = g.download(A.VELOCITIES)
        g.destroy()
        return r
    a = run([(False, 12), (True, 12)])
    assert np.array_equal(a, run([(False, 12), (True, 12)]))
    assert not np.array_equal(a, run([(False, 24)]))
    assert not np.array_equal(a, run([(True, 24)]))
