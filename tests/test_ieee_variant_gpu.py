"""The strict tolerance stays demonstrable.  k_fem's reciprocals and square roots are the hardware
approximations plus one Newton step since round 3 (mpm_math.h, MPM_FEM_MATH = 1: 2.4 us per substep), which ended the
coincidence that engine and float oracle round alike and moved the one-substep velocity error on the fast parity scenes
above north_star's 1e-5 of max|v| (1.0 - 1.7e-5; tests/helpers.py measures velocities against the float noise of the state
instead).  The correctly rounded path is still in the source (-DMPM_FEM_MATH=0).  Here it is BUILT and RUN, next to the
product build, on the same states:

* the 256^3 parity scene (0.8 m/s): the IEEE build is within the PLAIN 1e-5 of max|v| of the float oracle after one
  substep -- no noise floor; the product build within 2e-5: the trade that was chosen, in numbers;
* config 1 as released (max|v| = 0.014 m/s): here 1e-5 of max|v| is 1.4e-7 m/s, a sixth of what the float and the
  double build of the ORACLE differ by (8e-7 m/s: one ulp of F is dt E / (rho dx) x 1.2e-7 of velocity whatever the
  velocities are) -- NEITHER build can meet it, and none could: both must be within 2 (IEEE) / 3 (fast) of that noise;
* config 2: the product build and the IEEE build differ by at most twice the float noise of the same substep: the fast
  math is a trade inside the rounding noise, not an error.

A process holds one engine library, so the IEEE build runs in a process of its own (tests/ieee_worker.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
VARIANT = os.path.join(ROOT, "drake_amd", "variants", "libmpm_hip_ieee.so")


def _build_variant():
    from drake_amd import _build
    src_time = max(os.path.getmtime(os.path.join(_build.CSRC, f)) for f in os.listdir(_build.CSRC))
    if not os.path.exists(VARIANT) or os.path.getmtime(VARIANT) < src_time:
        os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
        _build.build(out=VARIANT, extra=("-DMPM_FEM_MATH=0",))
    return VARIANT


def test_ieee_build_meets_the_plain_1e5_and_the_fast_math_stays_inside_the_float_noise(tmp_path):
    from drake_amd import ARR as A, GpuMpm, scenes
    from oracle import oracle as orc
    from tests.helpers import float_noise_of_a_substep
    sys.path.insert(0, ROOT)
    lib = _build_variant()
    out = str(tmp_path / "ieee.npz")
    env = dict(os.environ, MPM_HIP_LIBRARY=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ieee_worker.py"), out], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    ieee = json.loads(str(z["rel"]))
    from tests.ieee_worker import one_substep_errors
    fast = one_substep_errors()    # the product build, this process
    for tag in ieee:
        print(f"{tag}: one substep, |v - float oracle| / max|v|: IEEE build {ieee[tag]['vel_rel_plain']:.2e}, product "
              f"{fast[tag]['vel_rel_plain']:.2e}; in units of the float noise ({ieee[tag]['float_noise']:.1e} m/s): "
              f"{ieee[tag]['vel_in_noise']:.2f} / {fast[tag]['vel_in_noise']:.2f}; max|v| {ieee[tag]['max_abs_vel']:.3g} m/s")
    assert ieee["256^3 scene"]["vel_rel_plain"] <= 1e-5, ieee
    assert fast["256^3 scene"]["vel_rel_plain"] <= 2e-5, fast
    assert ieee["config 1"]["vel_in_noise"] <= 2.0 and fast["config 1"]["vel_in_noise"] <= 3.0, (ieee, fast)
    # config 2, the same substep on the product build (this process)
    bits, layers, res = scenes.CONFIGS["cloth_1m"]
    sheets = scenes.cloth_stack(layers, res, bits)
    g = GpuMpm(bits)
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    o = orc.OracleMpm(bits)
    for pos, vel, idx in sheets:
        o.add_qr_cloth(pos, vel, idx)
    o.finalize()
    noise = float_noise_of_a_substep(o, 1e-3)
    g.substep(1e-3, -1)
    g.gpu_sync()
    d_v = float(np.abs(g.download(A.VELOCITIES).astype(np.float64) - z["vel"]).max())
    d_F = float(np.abs(g.download(A.DEFORMATION_GRADIENTS).astype(np.float64) - z["F"]).max())
    print(f"config 2: |fast - ieee| velocities {d_v:.3e} m/s, F {d_F:.3e}; float noise of the substep {noise:.3e} m/s")
    assert d_v <= 2.0 * noise, (d_v, noise)
    assert d_F <= 10 * 1.2e-7, d_F    # a few ulp of F (|F| ~ 1)
