"""Device-side contact-pair generation (mpm_generate_contact_pairs)
against a numpy restatement of DeformableDriver::CalcMpmContactPairs (deformable_driver.h:120-194)
with analytic signed distance fields."""
import numpy as np
import pytest

from tests.helpers import IMPULSE_RTOL

pytestmark = pytest.mark.gpu
DT = 1e-3
F = np.float32


def _rot(axis, angle):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return (np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K).astype(F)


def sdf(kind, xb, dims):
    """phi and body-frame gradient for points xb [n,3] (float32 arithmetic)."""
    n = xb.shape[0]
    g = np.zeros((n, 3), F)
    if kind == 0:
        g[:, 2] = 1
        return xb[:, 2].copy(), g
    if kind == 1:
        ln = np.sqrt((xb * xb).sum(1, dtype=F))
        return ln - F(dims[0]), xb / ln[:, None]
    if kind == 2:
        h = np.asarray(dims, F)
        q = np.abs(xb) - h
        sg = np.where(xb < 0, F(-1), F(1))
        m = q.max(1)
        inside = m <= 0
        a = np.where((q[:, 0] >= q[:, 1]) & (q[:, 0] >= q[:, 2]), 0, np.where(q[:, 1] >= q[:, 2], 1, 2))
        gi = np.zeros((n, 3), F)
        gi[np.arange(n), a] = sg[np.arange(n), a]
        o = np.maximum(q, 0)
        lo = np.sqrt((o * o).sum(1, dtype=F))
        with np.errstate(invalid="ignore", divide="ignore"):
            go = sg * o / lo[:, None]
        return np.where(inside, m, lo), np.where(inside[:, None], gi, go).astype(F)
    zc = np.clip(xb[:, 2], -F(dims[1]), F(dims[1]))
    r = xb.copy()
    r[:, 2] -= zc
    ln = np.sqrt((r * r).sum(1, dtype=F))
    return ln - F(dims[0]), r / ln[:, None]


def reference_pairs(pos, colliders):
    """Pairs in ascending (slot, collider) order."""
    per = []
    for j, c in enumerate(colliders):
        R = np.array(c.R_WB[:], F).reshape(3, 3)
        p = np.array(c.p_WB[:], F)
        d = pos - p
        xb = (d @ R).astype(F)              # R^T d
        phi, gb = sdf(c.kind, xb, c.dims[:])
        gw = (gb @ R.T).astype(F)
        v, w = np.array(c.v[:], F), np.array(c.w[:], F)
        rv = v + np.cross(np.broadcast_to(w, d.shape), d).astype(F)
        per.append((phi.astype(F), gw, rv, p, c.body))
    rows = []
    for j, (phi, gw, rv, p, body) in enumerate(per):
        s = np.nonzero(phi < 0)[0]
        rows.append(np.stack([s, np.full_like(s, j)], 1))
    order = np.concatenate(rows)
    order = order[np.lexsort((order[:, 1], order[:, 0]))]
    s, j = order[:, 0], order[:, 1]
    phi = np.array([per[b][0][a] for a, b in zip(s, j)], F)
    nrm = np.array([-per[b][1][a] for a, b in zip(s, j)], F).reshape(-1, 3)
    rv = np.array([per[b][2][a] for a, b in zip(s, j)], F).reshape(-1, 3)
    pwb = np.array([per[b][3] for b in j], F).reshape(-1, 3)
    body = np.array([per[b][4] for b in j], np.uint32)
    return s.astype(np.uint32), body, phi, nrm, pos[s], rv, pwb


def _colliders():
    from drake_amd import Collider
    return [
        Collider(0, body=0, p_WB=(0.5, 0.5, 0.493), R_WB=_rot((1, 0, 0), 0.05)),
        Collider(1, body=1, p_WB=(0.45, 0.5, 0.5), dims=(0.06, 0, 0), v=(0.1, 0, 0), w=(0, 0, 2.0)),
        Collider(2, body=2, p_WB=(0.62, 0.55, 0.5), R_WB=_rot((0, 0, 1), 0.4), dims=(0.05, 0.04, 0.03), w=(0.5, 0, 0)),
        Collider(3, body=3, p_WB=(0.5, 0.35, 0.5), R_WB=_rot((0, 1, 0), 1.2), dims=(0.03, 0.08, 0), v=(0, 0, 0.2)),
    ]


def test_generated_pairs_match_the_host_loop():
    from drake_amd import GpuMpm, scenes
    g = GpuMpm(7)
    scenes.populate(g, scenes.cloth_stack(4, 60, 7, z0=0.49, vel_amp=0.2))
    g.reallocate_external_bodies(4)
    g.run_substeps(3, DT, -1)
    g.rebuild_mapping(True)                       # pairs are indexed by the caller's (sorted) slot order
    cols = _colliders()
    n = g.generate_contact_pairs(cols)
    pos = g.sync_particle_state_to_cpu()
    got = g.download_contact_pairs()
    ref = reference_pairs(pos, cols)
    # pairs closer to a surface than rounding can tell apart may differ between the two evaluations
    tol = 2e-6
    key = lambda t: {(int(a), int(b)) for a, b, d in zip(t[0], t[1], t[2]) if abs(d) > tol}
    assert key(got) == key(ref)
    assert abs(n - ref[0].size) <= max(4, n // 500)
    assert n > 1000 and len(set(int(b) for b in got[1])) == 4
    # ascending (slot, collider) order, collider ids are 0..3 here
    ids = got[0].astype(np.int64) * 8 + got[1]
    assert np.all(np.diff(ids) > 0)
    lookup = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(ref[0], ref[1]))}
    sel = [(k, lookup[(int(a), int(b))]) for k, (a, b) in enumerate(zip(got[0], got[1])) if (int(a), int(b)) in lookup]
    gi, ri = np.array(sel).T
    assert gi.size >= n - 8
    np.testing.assert_allclose(got[2][gi], ref[2][ri], atol=2e-6)
    # the normal of a box contact can flip between two faces when the point is equally deep in both
    nd = np.abs(got[3][gi] - ref[3][ri]).max(1)
    assert np.mean(nd > 1e-4) < 2e-3
    np.testing.assert_array_equal(got[4][gi], ref[4][ri])
    np.testing.assert_allclose(got[5][gi], ref[5][ri], atol=1e-6)
    np.testing.assert_array_equal(got[6][gi], ref[6][ri])
    np.testing.assert_allclose(np.linalg.norm(got[3], axis=1), 1.0, atol=1e-5)


def test_solve_is_the_same_with_generated_and_with_uploaded_pairs():
    from drake_amd import ARR as A, Collider, GpuMpm, scenes

    def prepared():
        g = GpuMpm(7)
        sheets = scenes.cloth_stack(2, 40, 7, z0=0.5 - 0.004, vel_amp=0.2)
        for pos, vel, idx in sheets:
            vel[:, 2] -= 0.5
        scenes.populate(g, sheets)
        g.reallocate_external_bodies(1)
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(DT)
        g.particle_to_grid(DT)
        g.update_grid(-1)
        return g

    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, 0.5))]
    a = prepared()
    n = a.generate_contact_pairs(floor)
    pairs = a.download_contact_pairs()
    ra = a.update_contact(DT, 0.5, 1e5, 1e-3)
    b = prepared()
    b.copy_contact_pairs(*pairs)
    rb = b.update_contact(DT, 0.5, 1e5, 1e-3)
    # two engines: the re-sort's order inside a cell differs between them, so sums agree to rounding
    assert n > 100 and abs(ra["iterations"] - rb["iterations"]) <= 1
    np.testing.assert_allclose(a.download(A.CONTACT_VEL), b.download(A.CONTACT_VEL), rtol=2e-3, atol=2e-5)
    ta, fa = a.external_body_force_to_host()
    tb, fb = b.external_body_force_to_host()
    np.testing.assert_allclose(fa, fb, rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(ta, tb, rtol=2e-3, atol=1e-7)
    assert abs(fa[0, 2]) > 0


def test_moving_capsules_impulses_per_body_match_the_oracle():
    """SURVEY.md 8(d) config 5 in miniature: several bodies with prescribed rigid velocities and
    analytic capsule fields; pairs made on the device, the same pairs solved by the oracle; the
    per-body impulse output (F_Bq_W_tau, F_Bq_W_f) must agree body by body."""
    from drake_amd import ARR as A, Collider
    from oracle import oracle as orc
    from tests.helpers import build_pair, close, natural_scales
    o, g = build_pair(layers=3, res=24, z0=0.5, vel_amp=0.2)
    links = [
        Collider(3, body=0, p_WB=(0.42, 0.45, 0.497), R_WB=_rot((0, 1, 0), 1.5708), dims=(0.012, 0.05, 0), v=(0, 0, 0.3)),
        Collider(3, body=1, p_WB=(0.58, 0.45, 0.499), R_WB=_rot((1, 0, 0), 1.5708), dims=(0.012, 0.05, 0), v=(0.2, 0, 0.2),
                 w=(0, 0, 3.0)),
        Collider(3, body=2, p_WB=(0.50, 0.58, 0.515), R_WB=_rot((1, 1, 0), 1.2), dims=(0.015, 0.04, 0), v=(0, -0.1, -0.4)),
        Collider(1, body=3, p_WB=(0.45, 0.56, 0.49), dims=(0.02, 0, 0), v=(0, 0, 0.5)),
    ]
    for s in (o, g):
        s.reallocate_external_bodies(4)
        s.rebuild_mapping(False)
        s.calc_fem_state_and_force(DT)
        s.particle_to_grid(DT)
        s.update_grid(-1)
    n = g.generate_contact_pairs(links)
    pairs = g.download_contact_pairs()
    assert n > 60 and set(int(b) for b in pairs[1]) == {0, 1, 2, 3}
    o.copy_contact_pairs(orc.ContactPairs(*pairs))
    ro = o.update_contact(DT, 0.5, 1e5, 1e-3)
    rg = g.update_contact(DT, 0.5, 1e5, 1e-3)
    assert abs(rg["iterations"] - ro["iterations"]) <= max(3, ro["iterations"] // 4), (rg, ro)
    sc = natural_scales(o)
    from tests.helpers import solve_tolerance
    close(g.download(A.CONTACT_VEL), o.c_vel, scale=1.0, rtol=solve_tolerance(g.contact_stats()["dofs"]),
          what="contact vel (capsules)")
    tau_g, f_g = g.external_body_force_to_host()
    fscale = float(np.abs(o.F_f).max())
    close(f_g, o.F_f, scale=fscale, rtol=IMPULSE_RTOL, what="per-body impulse")
    close(tau_g, o.F_tau, scale=float(np.abs(o.F_tau).max()), rtol=IMPULSE_RTOL, what="per-body angular impulse")
    assert np.all(np.abs(o.F_f).max(1) > 0)      # every body took part
    # FinalizeExternalContactForces (deformable_driver.h:210-219): impulse / plant step, then what the
    # plant does with it (multibody_plant.cc:2396-2404) for a force reported at a point Bq != Bo
    from drake_amd import capi
    plant_dt = 4 * DT
    tau_F, f_F = g.finalize_external_contact_forces(plant_dt)
    assert np.array_equal(tau_F, tau_g / np.float32(plant_dt)) and np.array_equal(f_F, f_g / np.float32(plant_dt))
    close(f_F, o.F_f / plant_dt, scale=fscale / plant_dt, rtol=IMPULSE_RTOL, what="per-body force")
    R = np.stack([np.asarray(c.R_WB, np.float32) for c in links])
    p_BoBq_B = np.array([[0.01, -0.02, 0.03]] * 4, np.float32)
    tau_Bo = capi.external_forces_at_body_origin(R, p_BoBq_B, tau_F, f_F)
    pw = np.einsum("nij,nj->ni", R.reshape(4, 3, 3).astype(np.float64), p_BoBq_B.astype(np.float64))
    want = (o.F_tau.astype(np.float64) + np.cross(pw, o.F_f.astype(np.float64))) / plant_dt
    close(tau_Bo, want, scale=float(np.abs(want).max()), rtol=IMPULSE_RTOL, what="per-body torque about Bo")
