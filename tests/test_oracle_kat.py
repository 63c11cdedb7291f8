"""Pins the CPU oracle against the reference-derived known answers of
SURVEY.md Appendix A (tests/golden/survey_known_answers.json) and against
physical invariants.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

RT = 2e-6  # the known answers are printed with 9 significant digits


def _arr(x):
    return np.ascontiguousarray(np.array(x, dtype=np.float32))


def _close(a, b, rtol=RT, atol=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) if b.size else 1.0
    assert np.all(np.abs(a - b) <= rtol * np.abs(b) + rtol * scale * 1e-1 + atol), (a, b)


def test_cell_index_and_morton(golden):
    for e in golden["cell_index"]:
        assert orc.cell_index(*e["xyz"]) == e["key"]
        assert orc.inverse_cell_index(e["key"]) == tuple(e["xyz"])
    for e in golden["morton_code"]:
        assert orc.morton_code(*e["xyz"]) == e["code"]


def test_particle_key(golden):
    e = golden["particle_key"]
    p = orc.default_params(7)
    pos = _arr([e["x"]])
    keys = np.zeros(1, np.uint32)
    ids = np.zeros(1, np.uint32)
    orc.lib().orc_compute_keys(C.byref(p), C.c_size_t(1), orc._f(pos), orc._u(keys), orc._u(ids))
    assert int(keys[0]) == e["key"]
    assert orc.inverse_cell_index(int(keys[0])) == tuple(e["base"])


def test_cell_index_roundtrip_random():
    rng = np.random.default_rng(0)
    for x, y, z in rng.integers(0, 1024, size=(2000, 3)):
        assert orc.inverse_cell_index(orc.cell_index(x, y, z)) == (x, y, z)


def test_givens_qr(golden):
    e = golden["givens_qr33"]
    F = _arr(e["F"])
    Q = np.zeros(9, np.float32)
    R = np.zeros(9, np.float32)
    orc.lib().orc_kat_givens_qr33(orc._f(F), orc._f(Q), orc._f(R))
    _close(Q, e["Q"], atol=1e-9)
    _close(R, e["R"], atol=1e-9)
    np.testing.assert_allclose(Q.reshape(3, 3) @ R.reshape(3, 3), F.reshape(3, 3), atol=2e-7)


def test_dphi_dF(golden):
    e = golden["compute_dphi_dF"]
    p = orc.default_params(7)
    F = _arr(e["F"])
    P = np.zeros(9, np.float32)
    orc.lib().orc_kat_dphi_dF(C.byref(p), orc._f(F), orc._f(P))
    # entries are O(4e4) with cancellation: compare relative to the matrix scale
    assert np.max(np.abs(P - np.array(e["P"]))) <= 3e-6 * np.max(np.abs(e["P"]))
    P0 = np.zeros(9, np.float32)
    I = _arr([1, 0, 0, 0, 1, 0, 0, 0, 1])
    orc.lib().orc_kat_dphi_dF(C.byref(p), orc._f(I), orc._f(P0))
    assert np.all(P0 == 0)


def test_project_strain(golden):
    e = golden["project_strain"]
    p = orc.default_params(7)
    F = _arr(e["F"])
    orc.lib().orc_kat_project_strain(C.byref(p), orc._f(F))
    assert np.max(np.abs(F - np.array(e["out"]))) <= 3e-7


def test_svd2_and_pk1(golden):
    e = golden["svd2x2"]
    p = orc.default_params(7)
    A = _arr(e["A"])
    U = np.zeros(4, np.float32)
    s = np.zeros(4, np.float32)
    V = np.zeros(4, np.float32)
    orc.lib().orc_kat_svd2(orc._f(A), orc._f(U), orc._f(s), orc._f(V))
    _close(U, e["U"])
    _close([s[0], s[3]], e["sigma"])
    _close(V, e["V"])
    P = np.zeros(4, np.float32)
    orc.lib().orc_kat_pk1_2d(C.byref(p), orc._f(A), orc._f(P))
    assert np.max(np.abs(P - np.array(e["pk1_2d"]))) <= 3e-6 * np.max(np.abs(e["pk1_2d"]))


def test_frame_from_unit_vector(golden):
    e = golden["make_from_one_unit_vector"]
    u = _arr(e["u"])
    J = np.zeros(9, np.float32)
    orc.lib().orc_kat_frame(orc._f(u), C.c_int(e["axis"]), orc._f(J))
    assert np.max(np.abs(J - np.array(e["J"]))) <= 2e-7
    Jm = J.reshape(3, 3)
    np.testing.assert_allclose(Jm @ Jm.T, np.eye(3), atol=3e-7)


def test_contact_grad_hess(golden):
    e = golden["contact_grad_hess"]
    p = orc.default_params(7)
    H = np.zeros(9, np.float32)
    g = np.zeros(3, np.float32)
    v0 = _arr(e["v0"])
    v = _arr(e["v"])
    orc.lib().orc_kat_contact_grad_hess(C.byref(p), orc._cf(e["phi0"]), orc._cf(e["dt"]), orc._cf(e["k"]),
                                        orc._cf(e["d"]), orc._cf(e["mu"]), orc._f(v0), orc._f(v), orc._f(H), orc._f(g))
    _close(g, e["grad"])
    _close([H[0], H[1], H[3], H[4]], e["hess_tt"])
    _close([H[8]], [e["hess_nn"]])
    assert H[2] == H[5] == H[6] == H[7] == 0


def test_contact_cost_gradient_consistency():
    """The contact gradient is minus the derivative of the cost l(v) (cuda_mpm_kernels.cuh:1425-1435 vs :1000-1038)."""
    p = orc.default_params(7)
    L = orc.lib()
    phi0, dt, k, d, mu = 2e-3, 1e-3, 1e5, 1e-3, 0.5
    v0 = _arr([0.05, 0.02, -0.3])
    v = np.array([0.3, -0.2, -0.1], np.float64)
    H = np.zeros(9, np.float32)
    g = np.zeros(3, np.float32)
    vf = _arr(v)
    L.orc_kat_contact_grad_hess(C.byref(p), orc._cf(phi0), orc._cf(dt), orc._cf(k), orc._cf(d), orc._cf(mu),
                                orc._f(v0), orc._f(vf), orc._f(H), orc._f(g))
    h = 1e-2
    for a in range(3):
        vp, vm = v.copy(), v.copy()
        vp[a] += h
        vm[a] -= h
        lp = L.orc_kat_contact_cost(C.byref(p), orc._cf(phi0), orc._cf(dt), orc._cf(k), orc._cf(d), orc._cf(mu),
                                    orc._f(v0), orc._f(_arr(vp)))
        lm = L.orc_kat_contact_cost(C.byref(p), orc._cf(phi0), orc._cf(dt), orc._cf(k), orc._cf(d), orc._cf(mu),
                                    orc._f(v0), orc._f(_arr(vm)))
        fd = (lp - lm) / (2 * h)
        assert abs(-fd - g[a]) <= 2e-2 * max(1.0, abs(g[a])), (a, fd, g[a])


def _one_particle(golden):
    e = golden["one_particle_substep"]
    o = orc.OracleMpm(7)
    # state is injected directly (the known answer bypasses the cloth set-up)
    o.n_faces, o.n_verts, o.n_particles = 0, 1, 1
    o._pos, o._vel, o._idx = [np.array([e["x"]], np.float32)], [np.array([e["v"]], np.float32)], [np.zeros(0, np.int32)]
    o.finalize()
    o.vol[:] = e["vol"]
    o.C[:] = np.array(e["C"], np.float32)
    return e, o


def test_one_particle_p2g_grid_g2p(golden):
    e, o = _one_particle(golden)
    dt = e["dt"]
    o.rebuild_mapping(False)
    o.forces[:] = np.array(e["f"], np.float32)
    o.taus[:] = np.array(e["tau"], np.float32)
    o.particle_to_grid(dt)
    c = orc.cell_index(*e["node"])
    _close([o.g_m[c]], [e["node_m"]])
    _close(o.g_mv[c], e["node_mv"])
    # mass / momentum bookkeeping of the scatter
    m = e["vol"] * 2000.0
    assert abs(o.g_m.sum() - m) <= 1e-6 * m
    o.update_grid(-1)
    assert o.g_cnt == e["touched_blocks"]
    assert int(o.g_ids[0]) == e["first_touched_id"]
    o.grid_to_particle(dt)
    _close(o.pos[0], e["x_after"])
    _close(o.vel[0], e["v_after"])
    assert np.max(np.abs(o.C[0] - np.array(e["C_after"]))) <= 3e-6 * np.max(np.abs(e["C_after"]))


def test_stable_sort_low_bits():
    rng = np.random.default_rng(3)
    n = 5000
    keys = rng.integers(0, 1 << 21, n).astype(np.uint32)
    ids = np.arange(n, dtype=np.uint32)
    k_in, i_in = keys.copy(), ids.copy()
    k_out = np.zeros(n, np.uint32)
    i_out = np.zeros(n, np.uint32)
    orc.lib().orc_sort_pairs_low_bits(C.c_size_t(n), orc._u(k_in), orc._u(i_in), orc._u(k_out), orc._u(i_out), C.c_int(16))
    order = np.argsort(keys & 0xFFFF, kind="stable")
    assert np.array_equal(i_out, ids[order])
    assert np.array_equal(k_out, keys[order])
    assert np.array_equal(k_in, k_out) and np.array_equal(i_in, i_out)


def _small_cloth(o, res=24, z=0.5, side=0.25):
    xs = np.linspace(0.5 - side / 2, 0.5 + side / 2, res, dtype=np.float32)
    pos = np.stack(np.meshgrid(xs, xs, indexing="ij"), -1).reshape(-1, 2)
    pos = np.concatenate([pos, np.full((pos.shape[0], 1), z, np.float32)], 1).astype(np.float32)
    idx = []
    for i in range(res - 1):
        for j in range(res - 1):
            p = lambda a, b: a * res + b
            idx += [p(i, j), p(i + 1, j), p(i, j + 1), p(i + 1, j + 1), p(i, j + 1), p(i + 1, j)]
    o.add_qr_cloth(pos, np.zeros_like(pos), np.array(idx, np.int32))
    o.finalize()


def test_free_fall_and_sort_invariance():
    """A flat cloth in free fall has v_z = g t (SURVEY Appendix A sanity check); sorting must not change physics."""
    dt = 1e-3
    a = orc.OracleMpm(6)
    b = orc.OracleMpm(6)
    _small_cloth(a)
    _small_cloth(b)
    total_vol = a.vol.sum()
    for step in range(10):
        a.substep(dt, -1, sort=False)
        b.substep(dt, -1, sort=(step % 3 == 0))
    np.testing.assert_allclose(a.vel[:, 2], -9.8 * dt * 10, rtol=2e-5)
    sa, sb = a.state_in_original_order(), b.state_in_original_order()
    # C is a sum of terms of size 4/dx * |v| * w that cancel for a rigid translation, so its
    # natural scale is 4/dx * max|v|, not max|C| (which is rounding noise here).
    scale = {"pos": 1.0, "vel": np.max(np.abs(sa["vel"])), "vol": np.max(sa["vol"]),
             "C": 4.0 * 64 * np.max(np.abs(sa["vel"]))}
    for k in sa:
        assert np.max(np.abs(sa[k] - sb[k])) <= 1e-5 * scale[k], k
    assert abs(a.vol.sum() - total_vol) == 0
    # grid mass equals particle mass after the last scatter
    m = a.vol.sum() * 2000.0
    assert abs(a.g_m.sum() - m) <= 2e-5 * m
    # touched set is exactly the set of blocks reached by a particle stencil
    keys = a.sort_keys
    assert a.g_cnt > 0 and a.g_cnt == len(a.touched_blocks())


def test_dump_cpu_state_unpermutes():
    o = orc.OracleMpm(6)
    _small_cloth(o, res=10)
    p0, i0 = o.dump_cpu_state()
    o.rebuild_mapping(True)
    p1, i1 = o.dump_cpu_state()
    assert np.array_equal(p0, p1) and np.array_equal(i0, i1)
    assert not np.array_equal(o.pids, np.arange(o.n_particles))


def test_colored_scatter_matches_plain_scatter():
    """The multi-core scatter used for the CPU baseline is the same function up to summation order."""
    a = orc.OracleMpm(6)
    b = orc.OracleMpm(6)
    _small_cloth(a)
    _small_cloth(b)
    b.fast_scatter = True
    for _ in range(3):
        a.substep(1e-3, -1)
        b.substep(1e-3, -1)
    assert np.array_equal(a.g_flags, b.g_flags)
    assert np.max(np.abs(a.g_m - b.g_m)) <= 2e-6 * a.g_m.max()
    assert np.max(np.abs(a.pos - b.pos)) <= 1e-6
    assert np.max(np.abs(a.vel - b.vel)) <= 1e-5 * max(np.abs(a.vel).max(), 1e-2)


@pytest.mark.parametrize("bc", [-1, 0, 1, 2, 3])
def test_collider_table_presets_reproduce_the_scenes(bc):
    """orc_update_grid_table fed with the preset table of scene bc (the engine's
    mpm_grid_collider_preset: host code, no GPU) equals orc_update_grid(bc), the line-by-line
    restatement of update_grid_kernel<bc> (cuda_mpm_kernels.cuh:632-796), bit for bit."""
    from drake_amd import grid_collider_preset, scenes
    z0 = {-1: 0.5, 0: 0.56, 1: 0.75, 2: 0.11, 3: 0.5}[bc]
    side = {-1: 0.3, 0: 0.3, 1: 0.34, 2: 0.3, 3: 0.5}[bc]

    def scattered():
        o = orc.OracleMpm(6)
        for pos, vel, idx in scenes.cloth_stack(3, 20, 6, z0=z0, side=side, seed=7, vel_amp=0.5):
            vel[:, 2] -= 0.4
            o.add_qr_cloth(pos, vel, idx)
        o.finalize()
        o.rebuild_mapping(False)
        o.calc_fem_state_and_force(1e-3)
        o.particle_to_grid(1e-3)
        return o

    a = scattered()
    mv0 = a.g_mv.copy()
    a.update_grid(bc)
    want_mv, want_vs = a.g_mv.copy(), a.g_vstar.copy()
    table = []
    for c in grid_collider_preset(bc, a.p.sdf_friction):
        t = orc.GridCollider()
        C.memmove(C.byref(t), C.byref(c), C.sizeof(t))
        table.append(t)
    assert len(table) == {-1: 0, 0: 1, 1: 2, 2: 1, 3: 4}[bc]
    a.g_mv[:] = mv0
    a.g_vstar[:] = 0
    a.update_grid_table(table)
    assert np.array_equal(a.g_mv, want_mv) and np.array_equal(a.g_vstar, want_vs)
    if bc >= 0:   # the scene does something
        a.g_mv[:] = mv0
        a.update_grid(-1)
        assert not np.array_equal(a.g_mv, want_mv)
