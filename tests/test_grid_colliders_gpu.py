"""Runtime material / collider tables: the grid update with a caller-supplied
table of analytic colliders, and the demo variants of the settings.h constants, against the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import BAGGING_MATERIAL, build_pair, close, natural_scales

pytestmark = pytest.mark.gpu
DT = 1e-3


def _to_oracle(table):
    from oracle import oracle as orc
    out = []
    for c in table:
        t = orc.GridCollider()
        C.memmove(C.byref(t), C.byref(c), C.sizeof(t))
        out.append(t)
    return out


def _phase_compare(o, g, grid_update_o, grid_update_g, steps=3, tag=""):
    from drake_amd import ARR as A
    for _ in range(steps):
        sc = natural_scales(o)
        g.upload_particle_state(o.pos, o.vel, o.C, None, o.F)
        for s in (o, g):
            s.rebuild_mapping(False)
            s.calc_fem_state_and_force(DT)
            s.particle_to_grid(DT)
        s_tau = max(float(np.abs(o.taus).max()), sc["vol"] * 4e5)
        close(g.download(A.TAUS), o.taus, scale=s_tau, what=tag + "taus")
        close(g.download(A.GRID_MASSES), o.g_m, what=tag + "grid mass")
        grid_update_o()
        grid_update_g()
        assert np.array_equal(g.download(A.GRID_TOUCHED_IDS), o.touched_blocks())
        wgt = (o.g_m / o.g_m.max())[:, None]
        close(g.download(A.GRID_MOMENTUM) * wgt, o.g_mv * wgt, scale=sc["vel"], what=tag + "grid v")
        close(g.download(A.GRID_V_STAR) * wgt, o.g_vstar * wgt, scale=sc["vel"], what=tag + "grid v*")
        o.grid_to_particle(DT)
        g.grid_to_particle(DT)
        close(g.download(A.POSITIONS), o.pos, scale=1.0, what=tag + "pos")
        close(g.download(A.VELOCITIES), o.vel, scale=sc["vel"], what=tag + "vel")
        close(g.download(A.AFFINE), o.C, scale=4.0 * (1 << o.domain_bits) * sc["vel"], what=tag + "C")


def test_custom_collider_table_matches_the_oracle():
    """A table no reference scene has: a moving slip sphere, a tilted fixed half-space with a velocity,
    an always-active slip plane with friction 1, in that priority order."""
    from drake_amd import BC_TABLE, GridCollider
    o, g = build_pair(layers=3, res=22, z0=0.5, side=0.36, vel_amp=0.4)
    o.vel[:, 2] -= 0.6
    n = np.array([0.6, 0.0, 0.8], np.float32)
    table = [
        GridCollider(0, 1, p=(0.5, 0.5, 0.47), radius=0.06, v=(0.0, 0.1, 0.3), friction=0.3),
        GridCollider(1, 0, p=(0.36, 0.5, 0.5), n=n, v=(0.05, 0.0, 0.0)),
        GridCollider(1, 2, p=(0.0, 0.0, 0.497), n=(0, 0, 1), friction=1.0),
    ]
    g.set_grid_colliders(table)
    ot = _to_oracle(table)
    _phase_compare(o, g, lambda: o.update_grid_table(ot), lambda: g.update_grid(BC_TABLE), tag="table ")
    # the colliders acted: some node velocities are the prescribed collider velocity
    vs = o.g_vstar[o.g_m > 0]
    assert np.any(np.all(vs == np.array([0.05, 0, 0], np.float32), axis=1))


def test_negative_friction_selects_the_material_value_and_bad_tables_are_refused():
    from drake_amd import BC_TABLE, GridCollider, MpmError
    o, g = build_pair(layers=2, res=16, z0=0.11, material=dict(sdf_friction=0.65))
    g.set_grid_colliders([GridCollider(1, 2, p=(0, 0, 0.11), n=(0, 0, 1))])   # friction < 0: material's
    _phase_compare(o, g, lambda: o.update_grid(2), lambda: g.update_grid(BC_TABLE), steps=1, tag="mat-friction ")
    with pytest.raises(MpmError):
        g.set_grid_colliders([GridCollider(1, 2, n=(0, 0, 2))])      # not a unit normal
    with pytest.raises(MpmError):
        g.set_grid_colliders([GridCollider(0, 0, radius=0.0)])
    with pytest.raises(MpmError):
        g.set_grid_colliders([GridCollider(0, 0, radius=0.1)] * 17)
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(DT)
    g.particle_to_grid(DT)
    with pytest.raises(MpmError):
        g.update_grid(7)


@pytest.mark.parametrize("bc", [3, 2, -1])
def test_bagging_material_variants(bc):
    """settings.h:88-111: the demos rebuild the reference with K = 4e5, V = 0.2, SDF_FRICTION = 1.0
    (bagging: four pin spheres, bc 3; folding: plane, bc 2).  Here they are runtime material fields."""
    z0 = {-1: 0.5, 2: 0.11, 3: 0.5}[bc]
    side = {-1: 0.3, 2: 0.3, 3: 0.5}[bc]
    o, g = build_pair(z0=z0, side=side, vel_amp=0.4, material=BAGGING_MATERIAL)
    assert abs(o.p.K - 4e5) < 1 and abs(o.p.V - 0.2) < 1e-6 and o.p.sdf_friction == 1.0
    # compressed sheets so that the normal penalty (K) is active: scale the normal column of F
    o.F[:, 2] *= 0.97
    o.F[:, 5] *= 0.97
    o.F[:, 8] *= 0.97
    _phase_compare(o, g, lambda: o.update_grid(bc), lambda: g.update_grid(bc), tag="bagging ")
