"""The contact path without host round trips (the reference: deformable_driver.h:244-258 calling
cuda_mpm_solver.cu:185-621, which reads every position back, builds the pairs on the host, uploads them and then
synchronises >= 3 times per Newton iteration).

* pairs counted on the device and LEFT counted there (mpm_generate_contact_pairs without a count): the solve gives bit
  for bit what it gives with the count read back;
* more pairs than the buffers hold: nothing is solved on a truncated list -- the solve refuses itself on the device, the
  host grows the buffers, makes the pairs again and repeats; the result is that of buffers that were large enough;
* a settled scene, whose pair list repeats from substep to substep: the previous solve's sorted order, per-cell runs and
  node list are reused (verified on the device, entry by entry); results equal to the bit those of the full set-up; a
  list that changes after a repeat is caught on the device and the solve repeated with the full set-up."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT, MU, K, D = 2e-4, 1.0, 1e6, 1e-5      # config 3's parameters (mpm_bagging.cc:9,15-17)
Z_FLOOR = 0.5


def _engine(env=None, sheets=None, bodies=1):
    from drake_amd import GpuMpm, scenes
    # (deterministic from Finalize's own first sort on: two engines then agree to the bit, and a difference between two
    # of them is a difference between the paths under test)
    env = dict(env or {}, MPM_DETERMINISTIC="1")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g = GpuMpm(6)     # (environment switches are read per handle, at mpm_create)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    scenes.populate(g, [(p.copy(), v.copy(), i.copy()) for p, v, i in sheets])
    g.reallocate_external_bodies(bodies)
    return g


def _pressed_stack():
    """four sheets, the lowest two below the floor, moving down: contacts from the first substep on"""
    from drake_amd import scenes
    sheets = scenes.cloth_stack(4, 40, 6, z0=Z_FLOOR - 0.012, side=0.3, seed=21, vel_amp=0.05)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.3
    return sheets


def _coupled(g, floor, n, want_count):
    out = []
    for _ in range(n):
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(DT)
        g.particle_to_grid(DT)
        g.update_grid(-1)
        cnt = g.generate_contact_pairs(floor, want_count=want_count)
        r = g.update_contact(DT, MU, K, D)
        if want_count:
            assert cnt == r["contacts"]
        g.grid_to_particle(DT)
        out.append(r)
    g.gpu_sync()
    return out


def _state(g):
    from drake_amd import ARR as A
    tau, f = g.external_body_force_to_host()
    return dict(pos=g.download(A.POSITIONS), vel=g.download(A.VELOCITIES), F=g.download(A.DEFORMATION_GRADIENTS), tau=tau, f=f)


def _same(a, b):
    """particle state AND per-body impulses to the bit (round 6: k_ct_impulse adds the impulses as 64-bit fixed-point
    integers -- exact sums, whatever order the contacts arrive in; the reference's float atomics,
    cuda_mpm_kernels.cuh:1616-1658, and round 5's had no fixed order)"""
    for k in ("pos", "vel", "F", "tau", "f"):
        assert np.array_equal(a[k], b[k]), k


def _rows(rs, *keys):
    """per-substep results as an array (NaN residuals -- no DoF, as in the reference -- compare equal)"""
    return np.array([[float(r[k]) for k in keys] for r in rs])


def _same_rows(ra, rb, *keys):
    A, B = _rows(ra, *keys), _rows(rb, *keys)
    assert A.shape == B.shape, (A.shape, B.shape)
    bad = np.nonzero(~np.all((A == B) | (np.isnan(A) & np.isnan(B)), axis=1))[0]
    assert bad.size == 0, (keys, [(int(i), A[i].tolist(), B[i].tolist()) for i in bad[:6]])


def test_pairs_left_counted_on_the_device_give_the_same_solve():
    from drake_amd import Collider
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = _pressed_stack()
    a, b = _engine({"MPM_CT_NO_REUSE": "1"}, sheets), _engine({"MPM_CT_NO_REUSE": "1"}, sheets)
    ra, rb = _coupled(a, floor, 6, True), _coupled(b, floor, 6, False)
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    assert ra[0]["contacts"] > 500
    _same(_state(a), _state(b))
    # the pairs themselves, read back afterwards, are the ones the counted call produced
    a.rebuild_mapping(False); b.rebuild_mapping(False)
    for g in (a, b):
        g.calc_fem_state_and_force(DT); g.particle_to_grid(DT); g.update_grid(-1)
    na = a.generate_contact_pairs(floor)
    b.generate_contact_pairs(floor, want_count=False)
    pa, pb = a.download_contact_pairs(), b.download_contact_pairs()
    assert b.contact_pair_count() == na
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)


def test_more_pairs_than_the_buffers_hold_refuses_the_solve_and_repeats_it():
    from drake_amd import Collider
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = _pressed_stack()
    big = _engine({"MPM_CT_NO_REUSE": "1", "MPM_CT_INITIAL_CAPACITY": "100000"}, sheets)
    small = _engine({"MPM_CT_NO_REUSE": "1", "MPM_CT_INITIAL_CAPACITY": "100"}, sheets)
    rb, rs = _coupled(big, floor, 5, False), _coupled(small, floor, 5, False)
    assert rb[0]["contacts"] > 500
    cb, cs = big.contact_counters(), small.contact_counters()
    assert cb["repeated_overflow"] == 0 and cs["repeated_overflow"] >= 1, (cb, cs)
    _same_rows(rb, rs, "iterations", "contacts", "residual")
    _same(_state(big), _state(small))
    # ... and with the count read back at once (the overflow is then found by the read-back, before any solve)
    small2 = _engine({"MPM_CT_NO_REUSE": "1", "MPM_CT_INITIAL_CAPACITY": "100"}, sheets)
    rs2 = _coupled(small2, floor, 5, True)
    assert small2.contact_counters()["repeated_overflow"] >= 1
    _same_rows(rb, rs2, "iterations", "contacts")
    _same(_state(big), _state(small2))


def _resting_sheet():
    """one flat sheet a little below the floor, at rest: every particle is in contact and stays there for many substeps
    (the list of pairs repeats); a second sheet falls onto it later and changes the list"""
    from drake_amd import scenes
    low = scenes.cloth_stack(1, 36, 6, z0=Z_FLOOR - 0.0015, side=0.3, seed=5, vel_amp=0.0)
    high = scenes.cloth_stack(1, 36, 6, z0=Z_FLOOR + 0.004, side=0.3, seed=6, vel_amp=0.0)
    for pos, vel, idx in high:
        vel[:, 2] -= 1.0
    return low + high


def test_a_repeating_pair_list_reuses_the_setup_and_a_change_is_caught_on_the_device():
    from drake_amd import Collider
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = _resting_sheet()
    full, fast = _engine({"MPM_CT_NO_REUSE": "1"}, sheets), _engine(None, sheets)
    steps = 40
    rf, rr = _coupled(full, floor, steps, False), _coupled(fast, floor, steps, False)
    cf, cr = full.contact_counters(), fast.contact_counters()
    print("contact counters: full set-up", cf, "with reuse", cr, "contacts", [r["contacts"] for r in rr][::4])
    assert cf["reused"] == 0
    assert cr["reused"] >= 5, cr              # the list did repeat, and the repeats ran on the reused set-up
    assert cr["refused_stale"] >= 1, cr       # ... and a guess was wrong at least once (the second sheet arriving)
    assert len({r["contacts"] for r in rr}) > 1
    _same_rows(rf, rr, "iterations", "contacts", "residual")
    assert any(r["setup_reused"] for r in rr) and not any(r["setup_reused"] for r in rf)
    _same(_state(full), _state(fast))


def test_uploaded_pairs_go_through_the_same_paths():
    """CopyContactPairs (host-made pairs, the reference's route): the count is the host's, the set-up reuse and the mailbox
    are the same; a repeated identical hand-over is recognised as unchanged."""
    from drake_amd import ARR as A
    sheets = _resting_sheet()[:1]
    g, h = _engine(None, sheets), _engine({"MPM_CT_NO_REUSE": "1"}, sheets)
    res = {}
    for e in (g, h):
        rs = []
        for s in range(6):
            e.rebuild_mapping(False); e.calc_fem_state_and_force(DT); e.particle_to_grid(DT); e.update_grid(-1)
            pos = e.sync_particle_state_to_cpu()
            idx = np.nonzero(pos[:, 2] < Z_FLOOR)[0].astype(np.uint32)
            n = idx.size
            e.copy_contact_pairs(idx, np.zeros(n, np.uint32), (pos[idx, 2] - Z_FLOOR).astype(np.float32),
                                 np.tile(np.array([0, 0, -1], np.float32), (n, 1)), pos[idx].astype(np.float32),
                                 np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32))
            rs.append(e.update_contact(DT, MU, K, D))
            e.grid_to_particle(DT)
        e.gpu_sync()
        res[e] = rs
    assert g.contact_counters()["reused"] >= 2, g.contact_counters()
    _same_rows(res[g], res[h], "iterations", "contacts", "residual")
    _same(_state(g), _state(h))


def test_coupled_substeps_in_one_call_equal_the_seven_calls():
    """mpm_run_coupled_substeps (the loop body of deformable_driver.h:240-258, n times): bit for bit the state, the
    impulses and the per-substep iteration counts of the reference's seven calls per substep -- through a repeating pair
    list, a changing one, and no contacts at all."""
    from drake_amd import Collider
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = _resting_sheet()
    a, b = _engine(None, sheets), _engine(None, sheets)
    ra = _coupled(a, floor, 30, False)
    rb = b.run_coupled_substeps(12, DT, floor, MU, K, D) + b.run_coupled_substeps(18, DT, floor, MU, K, D)
    b.gpu_sync()
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    # (a substep without pairs has no set-up to reuse when the call went without generating pairs at all: see
    # test_contact_free_stretches_of_a_coupled_run_go_without_pair_generation)
    assert [r["setup_reused"] for r in ra if r["contacts"]] == [r["setup_reused"] for r in rb if r["contacts"]]
    assert any(r["contacts"] == 0 for r in rb) and any(r["setup_reused"] for r in rb)
    _same(_state(a), _state(b))
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0


def test_contact_free_stretches_of_a_coupled_run_go_without_pair_generation():
    """A sheet released two cells above the floor, falling at 1 m/s: 60 substeps without a pair, then the impact.  When a
    coupled substep had no pairs, a watch behind its GridToParticle asks whether the next one has any (exactly: a pair is a
    particle with phi < 0 where the substep starts), and the substeps after it are enqueued contact-free in chunks, each
    gated on "no watch has seen a particle in a collider since"; the ones that skipped themselves are run as coupled substeps.
    Bit for bit the seven calls per substep; most of the fall went without pair generation; the substeps of the chunk in
    which the sheet arrived were repeated."""
    from drake_amd import Collider, scenes
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = scenes.cloth_stack(1, 36, 6, z0=Z_FLOOR + 2.0 / 64, side=0.3, seed=9, vel_amp=0.02)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 1.0
    n = 220
    a, b, c = _engine(None, sheets), _engine(None, sheets), _engine({"MPM_CT_NO_WATCH": "1"}, sheets)
    ra = _coupled(a, floor, n, False)
    rb = b.run_coupled_substeps(90, DT, floor, MU, K, D) + b.run_coupled_substeps(n - 90, DT, floor, MU, K, D)
    rc = c.run_coupled_substeps(n, DT, floor, MU, K, D)
    for g in (a, b, c):
        g.gpu_sync()
        assert g.stats()["error_flags"] == 0
    first = next(i for i, r in enumerate(ra) if r["contacts"] > 0)
    assert 100 < first < 200 and max(r["contacts"] for r in ra) > 1000, first
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    _same_rows(ra, rc, "iterations", "contacts", "residual")
    _same(_state(a), _state(b))
    _same(_state(a), _state(c))
    cb, cc = b.contact_counters(), c.contact_counters()
    assert cc["contact_free"] == 0 and cc["solves"] == n
    # (each call starts with a coupled substep; chunks of 2, 4, 8, 16, 32, 32 ... substeps; the chunk that holds the
    # impact is cut short)
    assert cb["contact_free"] >= first - 8 and 1 <= cb["contact_free_repeated"] <= 32, (cb, first)
    assert cb["solves"] <= n - first + 40, (cb, first)


def test_contact_free_stretches_through_resorts():
    """The same with re-sorts on the way: a sheet that slides at 6 m/s two cells above the floor (a cell every 13 substeps)
    and never touches it.  The contact-free chunks go without re-sort checks while the quiet time of the last look at the
    control block lasts; a substep that meets a pending re-sort skips itself like one that meets a hit, and the tail of its
    chunk is repeated with the checks.  Bit for bit the seven calls."""
    from drake_amd import Collider, scenes
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = scenes.cloth_stack(1, 36, 6, z0=Z_FLOOR + 2.0 / 64, side=0.3, seed=4, vel_amp=0.05, center=(0.35, 0.5))
    for pos, vel, idx in sheets:
        vel[:, 0] += 6.0
    n = 120
    a, b = _engine(None, sheets), _engine(None, sheets)
    ra = _coupled(a, floor, n, False)
    rb = b.run_coupled_substeps(n, DT, floor, MU, K, D)
    b.gpu_sync()
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["rebuilds"] >= 3 and sa["rebuilds"] == sb["rebuilds"], (sa["rebuilds"], sb["rebuilds"])   # (Finalize's + two on the way)
    assert all(r["contacts"] == 0 for r in ra)
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    _same(_state(a), _state(b))
    cb = b.contact_counters()
    assert cb["contact_free"] >= n - 2 and cb["solves"] <= 2, cb


def test_coupled_substeps_without_colliders_are_contact_free_substeps():
    from drake_amd import scenes
    sheets = scenes.cloth_stack(2, 30, 6, z0=Z_FLOOR, side=0.3, seed=2, vel_amp=0.2)
    a, b = _engine(None, sheets), _engine(None, sheets)
    a.run_substeps(12, DT, -1)
    rb = b.run_coupled_substeps(12, DT, [], MU, K, D)
    a.gpu_sync()
    b.gpu_sync()
    assert all(r["contacts"] == 0 and r["iterations"] == 0 for r in rb) and b.contact_counters()["solves"] == 0
    _same(_state(a), _state(b))


def _coupled_with(g, colliders, n, exact):
    out = []
    for _ in range(n):
        g.rebuild_mapping(False)
        g.calc_fem_state_and_force(DT)
        g.particle_to_grid(DT)
        g.update_grid(-1)
        g.generate_contact_pairs(colliders, want_count=False)
        out.append(g.update_contact(DT, MU, K, D, exact_line_search=exact))
        g.grid_to_particle(DT)
    g.gpu_sync()
    return out


@pytest.mark.parametrize("exact", [False, True])
def test_coupled_substeps_with_several_bodies(exact):
    """The same equality with what a scene of Drake's has: three colliders on two bodies -- the floor, a box standing on
    it (body 0) and a sphere (body 1) that moves and spins, so that the rigid velocity at every contact point differs --, a
    particle inside two of them at once (two pairs), both line searches.  The impulses come back per body."""
    from drake_amd import Collider, scenes
    cols = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR)),
            Collider(2, body=0, p_WB=(0.42, 0.5, Z_FLOOR + 0.004), dims=(0.03, 0.05, 0.006)),
            Collider(1, body=1, p_WB=(0.58, 0.5, Z_FLOOR + 0.03), dims=(0.035, 0, 0), v=(0.2, 0.0, -0.4), w=(0.0, 3.0, 1.0))]
    sheets = scenes.cloth_stack(2, 36, 6, z0=Z_FLOOR - 0.001, side=0.3, seed=12, vel_amp=0.05)
    for pos, vel, idx in sheets:
        vel[:, 2] -= 0.3
    a, b = _engine(None, sheets, bodies=2), _engine(None, sheets, bodies=2)
    ra = _coupled_with(a, cols, 16, exact)
    rb = b.run_coupled_substeps(6, DT, cols, MU, K, D, exact_line_search=exact) + \
        b.run_coupled_substeps(10, DT, cols, MU, K, D, exact_line_search=exact)
    b.gpu_sync()
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    assert max(r["contacts"] for r in rb) > 1500 and min(r["iterations"] for r in rb) >= 1
    sa, sb = _state(a), _state(b)
    _same(sa, sb)
    # both bodies were pushed, the floor-and-box body downwards
    assert sb["f"].shape[0] == 2 and sb["f"][0, 2] < 0 and np.abs(sb["f"][1]).max() > 0


@pytest.mark.parametrize("gate_always", [False, True])
def test_coupled_substeps_through_resorts(gate_always):
    """A cloth sliding over the floor at 6 m/s (a cell every 13 substeps: several re-sorts).  mpm_run_coupled_substeps sends
    the four re-sort check launches only when the quiet time left says a re-sort may be due; a substep that goes without
    them and finds one pending skips itself as a whole -- transfer kernels, contact solve, GridToParticle -- and is run
    again (CT_DONE_GATED).  MPM_CT_GATE_ALWAYS makes every re-sort be found that way.  Either way: bit for bit the seven
    calls' result."""
    from drake_amd import Collider, scenes
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = scenes.cloth_stack(2, 36, 6, z0=Z_FLOOR - 0.002, side=0.3, seed=8, vel_amp=0.05, center=(0.35, 0.5))
    for pos, vel, idx in sheets:
        vel[:, 0] += 6.0
    a = _engine(None, sheets)
    # (MPM_CT_NO_WATCH: the stretches of this scene in which the cloth has left the floor would otherwise go contact-free,
    # with their re-sort checks in place -- here every re-sort is to be found by a coupled substep that skipped itself)
    b = _engine({"MPM_CT_GATE_ALWAYS": "1", "MPM_CT_NO_WATCH": "1"} if gate_always else None, sheets)
    ra = _coupled(a, floor, 60, False)
    rb = b.run_coupled_substeps(25, DT, floor, MU, K, D) + b.run_coupled_substeps(35, DT, floor, MU, K, D)
    b.gpu_sync()
    sa, sb = a.stats(), b.stats()
    assert sa["error_flags"] == 0 and sb["error_flags"] == 0
    assert sa["rebuilds"] >= 2 and sa["rebuilds"] == sb["rebuilds"], (sa["rebuilds"], sb["rebuilds"])
    if not gate_always:
        assert b.contact_counters()["contact_free"] > 0   # (the cloth leaves the floor on its way: those stretches too)
    if gate_always:
        # (Finalize's own sort is one of the re-sorts; the first substep of each of the two calls carries its check launches)
        assert b.contact_counters()["refused_stale"] >= 1, (b.contact_counters(), sb["rebuilds"])
    _same_rows(ra, rb, "iterations", "contacts", "residual")
    _same(_state(a), _state(b))


def test_a_pair_count_the_solve_may_not_index_with_is_refused_with_an_error_code(monkeypatch):
    """VERDICT r5 item 3 (the memory access fault of round 5's scratch, DESIGN.md section 3.3: a launch that formed an index
    into a per-pair array from its workgroup number instead of from the clamped count).  The invariant: every kernel that
    reads the device-side pair count clamps it to the capacity that sized the per-pair buffers, and a count it may not use
    -- larger than the capacity, or written by another pair generation than the one the solve was enqueued for -- makes the
    solve refuse itself with an error code before anything is indexed.  With MPM_POISON=1 every buffer that is not
    zero-initialised starts as 0xFF bytes, so an index formed from something never written would leave the arrays."""
    from drake_amd import Collider, MpmError
    monkeypatch.setenv("MPM_POISON", "1")
    floor = [Collider(0, body=0, p_WB=(0.5, 0.5, Z_FLOOR))]
    sheets = _pressed_stack()
    g, ref = _engine(sheets=sheets), _engine(sheets=sheets)

    def to_the_solve(e):
        e.rebuild_mapping(False)
        e.calc_fem_state_and_force(DT)
        e.particle_to_grid(DT)
        e.update_grid(-1)
        e.generate_contact_pairs(floor, want_count=False)

    # one good coupled substep on both (buffers allocated, a previous solve's count and set-up in place)
    for e in (g, ref):
        to_the_solve(e)
        r = e.update_contact(DT, MU, K, D)
        assert r["iterations"] > 0
        e.grid_to_particle(DT)
    n_good = g.contact_stats()["contacts"]
    assert n_good > 500
    # (1) a count far beyond the capacity of the per-pair buffers
    to_the_solve(g)
    g.debug_contact_count(count=50_000_000)
    with pytest.raises(MpmError) as err:
        g.update_contact(DT, MU, K, D)
    assert err.value.code == -4, err.value          # MPM_ERR_CAPACITY
    # (2) the largest int there is (index arithmetic in 32 bits would wrap around with it)
    g.generate_contact_pairs(floor, want_count=False)
    g.debug_contact_count(count=0x7FFFFFFF)
    with pytest.raises(MpmError) as err:
        g.update_contact(DT, MU, K, D)
    assert err.value.code == -4, err.value
    # (3) a count inside the capacity that the generation of THIS solve did not write (a stale one)
    g.generate_contact_pairs(floor, want_count=False)
    g.debug_contact_count(count=n_good, stamp_delta=-1)
    with pytest.raises(MpmError) as err:
        g.update_contact(DT, MU, K, D)
    assert err.value.code == -10, err.value         # MPM_ERR_INTERNAL
    # nothing was solved, nothing was written: with the pairs made again the substep is the untampered engine's, bit for bit
    g.generate_contact_pairs(floor, want_count=False)
    r_g = g.update_contact(DT, MU, K, D)
    g.grid_to_particle(DT)
    to_the_solve(ref)
    r_ref = ref.update_contact(DT, MU, K, D)
    ref.grid_to_particle(DT)
    assert (r_g["iterations"], r_g["residual"]) == (r_ref["iterations"], r_ref["residual"])
    g.gpu_sync(); ref.gpu_sync()
    assert g.stats()["error_flags"] == 0
    _same(_state(g), _state(ref))
