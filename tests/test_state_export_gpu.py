"""State export: DumpCpuState un-permutation after slot sorts and re-sorts,
the OBJ writer (GpuMpmSolver::Dump, cuda_mpm_solver.cu:168-183) and the solver-statistics JSON of
UpdateContact (cuda_mpm_solver.cu:587-612)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 1e-3


def test_dump_cpu_state_follows_the_vertices_through_sorts(tmp_path):
    from drake_amd import GpuMpm, scenes
    g = GpuMpm(6)
    sheets = scenes.cloth_stack(2, 14, 6, z0=0.55, vel_amp=0.4)
    scenes.populate(g, sheets)
    verts0 = np.concatenate([s[0] for s in sheets])
    p0, idx0 = g.dump_cpu_state()
    assert np.array_equal(p0, verts0)                       # original vertex order, original indices
    tri = np.concatenate([s[2].reshape(-1, 3) + off for s, off in zip(sheets, np.cumsum([0] + [s[0].shape[0] for s in sheets[:-1]]))])
    assert np.array_equal(idx0.reshape(-1, 3), tri)
    for k in range(12):
        g.rebuild_mapping(k % 4 == 0)                        # slot sorts in between
        g.calc_fem_state_and_force(DT)
        g.particle_to_grid(DT)
        g.update_grid(-1)
        g.grid_to_particle(DT)
    p1, idx1 = g.dump_cpu_state()
    assert np.array_equal(idx1, idx0)
    # vertices are still listed in their original order: neighbours in the mesh stay neighbours in space
    edge = np.linalg.norm(p1[tri[:, 0]] - p1[tri[:, 1]], axis=1)
    assert edge.max() < 3 * np.linalg.norm(verts0[tri[:, 0]] - verts0[tri[:, 1]], axis=1).max()
    # and agree with the slot-order view mapped back through pids
    from drake_amd import ARR as A
    pos_slot, pids = g.sync_particle_state_to_cpu(), g.download(A.PIDS)
    back = np.empty_like(pos_slot)
    back[pids] = pos_slot
    assert np.array_equal(back[g.n_faces:], p1)
    # OBJ writer
    path = os.path.join(tmp_path, "cloth.obj")
    g.dump(path)
    lines = open(path).read().splitlines()
    v = [ln for ln in lines if ln.startswith("v ")]
    f = [ln for ln in lines if ln.startswith("f ")]
    assert len(v) == g.n_verts and len(f) == g.n_faces
    got = np.array([[float(t) for t in ln.split()[1:4]] for ln in v], np.float32)
    np.testing.assert_allclose(got, p1, rtol=0, atol=1e-5)
    fi = np.array([[int(t) for t in ln.split()[1:4]] for ln in f])
    assert fi.min() == 1 and fi.max() == g.n_verts and np.array_equal(fi - 1, tri)


def test_contact_statistics_json(tmp_path):
    from drake_amd import Collider, GpuMpm, scenes
    g = GpuMpm(6)
    sheets = scenes.cloth_stack(2, 16, 6, z0=0.5 - 0.004, vel_amp=0.2)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.5
    scenes.populate(g, sheets)
    g.reallocate_external_bodies(1)
    g.set_dump_dir(str(tmp_path))
    g.rebuild_mapping(False)
    g.calc_fem_state_and_force(DT)
    g.particle_to_grid(DT)
    g.update_grid(-1)
    assert g.generate_contact_pairs([Collider(0, body=0, p_WB=(0.5, 0.5, 0.5))]) > 20
    for exact in (False, True):
        r = g.update_contact(DT, 0.5, 1e5, 1e-3, exact_line_search=exact, frame=3, substep=7 + int(exact), dump=True)
        fn = os.path.join(tmp_path, f"jacobi_iter_2000_frame_3_substep_{7 + int(exact)}.json")
        rec = json.load(open(fn))
        assert isinstance(rec, list) and len(rec) >= 1
        if exact:       # one record per Newton iteration
            assert len(rec) == r["iterations"] and {"residual", "line_search_cnt", "energy"} <= set(rec[0])
            assert rec[-1]["residual"] <= 1.5e-4
        else:           # one summary record
            assert rec[0]["iterations"] == r["iterations"] and abs(rec[0]["residual"] - r["residual"]) < 1e-6
