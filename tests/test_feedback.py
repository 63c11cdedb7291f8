"""Rigid feedback wire format, host side: SpatialForce::Shift and the shift
to the body origin that MultibodyPlant::AddAppliedExternalSpatialForces applies to
external_forces_host() (multibody/plant/multibody_plant.cc:2385-2407)."""
import numpy as np

from drake_amd import capi


def test_spatial_force_shift_reference_known_answer():
    """The reference's own vector: multibody/math/test/spatial_algebra_test.cc:612-653
    (ElementsInF6Test.ShiftOperation): tau = (0,0,3), f = (1,2,0), p_AB = (2,-2,1) -> tau' = (2,-1,-3)."""
    tau = np.array([[0, 0, 3]], np.float32)
    f = np.array([[1, 2, 0]], np.float32)
    p = np.array([[2, -2, 1]], np.float32)
    out = capi.spatial_force_shift(tau, f, p)
    assert np.array_equal(out, np.array([[2, -1, -3]], np.float32))
    # and back (ShiftInPlace(-p_AB), :669-673)
    assert np.array_equal(capi.spatial_force_shift(out, f, -p), tau)


def test_shift_to_body_origin_with_nonzero_p_BoBq():
    """tau_Bo = tau + (R_WB p_BoBq_B) x f  (Shift(-p_BoBq_W), multibody_plant.cc:2396-2404)."""
    rng = np.random.default_rng(5)
    n = 7
    # random rotations
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).astype(np.float32)
    p = rng.normal(size=(n, 3)).astype(np.float32)
    tau = rng.normal(size=(n, 3)).astype(np.float32)
    f = rng.normal(size=(n, 3)).astype(np.float32)
    got = capi.external_forces_at_body_origin(R, p, tau, f)
    pw = np.einsum("nij,nj->ni", R.reshape(n, 3, 3).astype(np.float64), p.astype(np.float64))
    want = tau.astype(np.float64) + np.cross(pw, f.astype(np.float64))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
    # p_BoBq_B = 0 (what FinalizeExternalContactForces leaves, deformable_driver.h:215): identity
    assert np.array_equal(capi.external_forces_at_body_origin(R, np.zeros_like(p), tau, f), tau)
    # a pure force at Bq produces the moment of that force about Bo
    f1 = np.array([[0, 0, -2.0]], np.float32)
    got = capi.external_forces_at_body_origin(np.eye(3, dtype=np.float32).reshape(1, 9), [[0.5, 0, 0]],
                                              np.zeros((1, 3), np.float32), f1)
    assert np.allclose(got, [[0, 1.0, 0]])
