"""ctypes front end of the CPU oracle (oracle/mpm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under drake_amd/ may import this module.

`OracleMpm` mirrors the reference's GpuMpmState + GpuMpmSolver pair
(multibody/gpu_mpm/cuda_mpm_model.cuh:37-260, cuda_mpm_solver.cuh:19-35) on
numpy arrays laid out like the reference's device buffers (array of small
vectors, dense grid indexed by cell key).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPM_ORACLE_LIB: load another build of the same source (e.g. `make -C oracle asan`)
_LIB_PATH = os.environ.get("MPM_ORACLE_LIB") or os.path.join(_HERE, "libmpm_oracle.so")
_lib = None
_lib64 = None
_LIB64_PATH = os.path.join(_HERE, "libmpm_oracle_f64.so")


class Params(C.Structure):
    _fields_ = [
        ("domain_bits", C.c_int),
        ("wall", C.c_int),
        ("gravity_axis", C.c_int),
        ("youngs", C.c_float),
        ("poisson", C.c_float),
        ("density", C.c_float),
        ("gamma", C.c_float),
        ("K", C.c_float),
        ("V", C.c_float),
        ("cF", C.c_float),
        ("sdf_friction", C.c_float),
        ("gravity", C.c_float),
        ("epsv", C.c_float),
    ]


class GridCollider(C.Structure):
    """Runtime grid collider, same layout as mpm_grid_collider_t (include/mpm_hip.h)."""
    _fields_ = [("shape", C.c_int32), ("mode", C.c_int32), ("p", C.c_float * 3), ("n", C.c_float * 3),
                ("radius", C.c_float), ("v", C.c_float * 3), ("friction", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "mpm_oracle.c")
    if os.environ.get("MPM_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmpm_oracle.so"])
    return _LIB_PATH


def _declare(L, real):
    L.orc_cell_index.restype = C.c_uint32
    L.orc_cell_index.argtypes = [C.c_uint32] * 3
    L.orc_morton_code.restype = C.c_uint32
    L.orc_morton_code.argtypes = [C.c_uint32] * 3
    L.orc_gather_touched.restype = C.c_uint32
    L.orc_update_contact.restype = C.c_int
    L.orc_kat_contact_cost.restype = real
    L.orc_max_threads.restype = C.c_int
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _declare(C.CDLL(_LIB_PATH), C.c_float)
    return _lib


def build_f64(force: bool = False) -> str:
    """The same source with -DORC_REAL=double (oracle/Makefile target f64): the yardstick of
    tests/test_precision_gpu.py for what float rounding alone does; never the parity oracle."""
    src = os.path.join(_HERE, "mpm_oracle.c")
    if force or not os.path.exists(_LIB64_PATH) or os.path.getmtime(_LIB64_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmpm_oracle_f64.so"])
    return _LIB64_PATH


def lib64():
    global _lib64
    if _lib64 is None:
        build_f64()
        _lib64 = _declare(C.CDLL(_LIB64_PATH), C.c_double)
    return _lib64


def default_params(domain_bits: int = 7) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    p.domain_bits = domain_bits
    return p


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _cf(x):
    return C.c_float(float(x))


def set_threads(n: int) -> None:
    lib().orc_set_threads(C.c_int(n))


def max_threads() -> int:
    return int(lib().orc_max_threads())


def cell_index(x, y, z) -> int:
    return int(lib().orc_cell_index(int(x), int(y), int(z)))


def morton_code(x, y, z) -> int:
    return int(lib().orc_morton_code(int(x), int(y), int(z)))


def inverse_cell_index(key) -> tuple:
    out = (C.c_uint32 * 3)()
    lib().orc_inverse_cell_index(C.c_uint32(int(key)), out)
    return (int(out[0]), int(out[1]), int(out[2]))


class ContactPairs:
    """MpmParticleContactPairs (cpu_mpm_model.h:73-114) as numpy SoA."""

    def __init__(self, particle, body, dist, normal, pos, rigid_v, rigid_p_WB, real=np.float32):
        self.particle = np.ascontiguousarray(particle, dtype=np.uint32)
        self.body = np.ascontiguousarray(body, dtype=np.uint32)
        self.dist = np.ascontiguousarray(dist, dtype=real)
        self.normal = np.ascontiguousarray(normal, dtype=real).reshape(-1, 3)
        self.pos = np.ascontiguousarray(pos, dtype=real).reshape(-1, 3)
        self.rigid_v = np.ascontiguousarray(rigid_v, dtype=real).reshape(-1, 3)
        self.rigid_p_WB = np.ascontiguousarray(rigid_p_WB, dtype=real).reshape(-1, 3)

    def __len__(self):
        return int(self.body.shape[0])


class OracleMpm:
    """GpuMpmState + GpuMpmSolver restated on the CPU."""

    def __init__(self, domain_bits: int = 7, params: Params | None = None, real=np.float32):
        """real = np.float64 runs the double build of the same source (see build_f64)."""
        self.real = np.dtype(real).type
        assert self.real in (np.float32, np.float64)
        self.L = lib() if self.real is np.float32 else lib64()
        self._creal = C.c_float if self.real is np.float32 else C.c_double
        self.p = params if params is not None else default_params(domain_bits)
        self.p.domain_bits = domain_bits
        self.domain_bits = domain_bits
        self.n_cells = 1 << (3 * domain_bits)
        self.n_blocks = self.n_cells >> 6
        self._pos, self._vel, self._idx = [], [], []
        self.n_verts = self.n_faces = self.n_particles = 0
        self.finalized = False
        self.contacts = None
        self.n_bodies = 0

    def _f(self, a):
        assert a.dtype == self.real and a.flags["C_CONTIGUOUS"], (a.dtype, self.real)
        return a.ctypes.data_as(C.POINTER(self._creal))

    def _cf(self, x):
        return self._creal(float(x))

    # -- GpuMpmState::AddQRCloth (cuda_mpm_model.cu:16-33)
    def add_qr_cloth(self, pos, vel, indices):
        pos = np.asarray(pos, dtype=self.real).reshape(-1, 3)
        vel = np.asarray(vel, dtype=self.real).reshape(-1, 3)
        indices = np.asarray(indices, dtype=np.int32).reshape(-1)
        assert indices.size % 3 == 0
        self._pos.append(pos)
        self._vel.append(vel)
        self._idx.append(indices + self.n_verts)
        self.n_verts += pos.shape[0]
        self.n_faces += indices.size // 3
        self.n_particles = self.n_verts + self.n_faces

    # -- GpuMpmState::Finalize (cuda_mpm_model.cu:36-122)
    def finalize(self):
        nf, nv, n = self.n_faces, self.n_verts, self.n_particles
        self.pos = np.zeros((n, 3), self.real)
        self.vel = np.zeros((n, 3), self.real)
        self.pos[nf:] = np.concatenate(self._pos) if self._pos else 0
        self.vel[nf:] = np.concatenate(self._vel) if self._vel else 0
        self.indices = (np.concatenate(self._idx) + nf).astype(np.int32)
        self.vol = np.zeros(n, self.real)
        self.C = np.zeros((n, 9), self.real)
        self.pids = np.arange(n, dtype=np.int32)
        self.index_mappings = np.arange(n, dtype=np.int32)
        self.sort_keys = np.zeros(n, np.uint32)
        self.sort_ids = np.zeros(n, np.uint32)
        self.forces = np.zeros((n, 3), self.real)
        self.taus = np.zeros((n, 9), self.real)
        self.F = np.zeros((nf, 9), self.real)
        self.DmInv = np.zeros((nf, 4), self.real)
        nc = self.n_cells
        self.g_m = np.zeros(nc, self.real)
        self.g_mv = np.zeros((nc, 3), self.real)
        self.g_vstar = np.zeros((nc, 3), self.real)
        self.g_flags = np.zeros(self.n_blocks, np.uint32)
        self.g_ids = np.zeros(self.n_blocks, np.uint32)
        self.g_cnt = 0
        self._contact_grid = False
        self.L.orc_initialize_fem_state(C.byref(self.p), C.c_size_t(nf), _i(self.indices), self._f(self.pos),
                                        self._f(self.vel), self._f(self.vol), self._f(self.F), self._f(self.DmInv))
        self.finalized = True

    def _ensure_contact_grid(self):
        if not self._contact_grid:
            nc = self.n_cells
            self.g_H = np.zeros((nc, 9), self.real)
            self.g_G = np.zeros((nc, 3), self.real)
            self.g_D = np.zeros((nc, 3), self.real)
            self.g_alpha = np.zeros(nc, self.real)
            self.g_E0 = np.zeros(nc, self.real)
            self.g_E1 = np.zeros(nc, self.real)
            self._contact_grid = True

    # -- GpuMpmSolver::RebuildMapping (cuda_mpm_solver.cu:17-70)
    def rebuild_mapping(self, sort: bool):
        n = self.n_particles
        self.L.orc_compute_keys(C.byref(self.p), C.c_size_t(n), self._f(self.pos), _u(self.sort_keys), _u(self.sort_ids))
        if sort:
            nk = np.zeros(n, np.uint32)
            ni = np.zeros(n, np.uint32)
            nbits = min(3 * self.domain_bits, 16)
            self.L.orc_sort_pairs_low_bits(C.c_size_t(n), _u(self.sort_keys), _u(self.sort_ids), _u(nk), _u(ni),
                                           C.c_int(nbits))
            npos = np.empty_like(self.pos)
            nvel = np.empty_like(self.vel)
            nvol = np.empty_like(self.vol)
            nC = np.empty_like(self.C)
            npid = np.empty_like(self.pids)
            self.L.orc_compute_sorted_state(C.c_size_t(n), self._f(self.pos), self._f(self.vel), self._f(self.vol), self._f(self.C),
                                            _i(self.pids), _u(ni), self._f(npos), self._f(nvel), self._f(nvol), self._f(nC), _i(npid),
                                            _i(self.index_mappings))
            self.pos, self.vel, self.vol, self.C, self.pids = npos, nvel, nvol, nC, npid
            self.sort_keys, self.sort_ids = nk, ni

    # -- GpuMpmSolver::CalcFemStateAndForce (cuda_mpm_solver.cu:72-84)
    def calc_fem_state_and_force(self, dt: float):
        self.forces[:] = 0
        self.taus[:] = 0
        self.L.orc_calc_fem_state_and_force(C.byref(self.p), C.c_size_t(self.n_faces), _i(self.indices),
                                            _i(self.index_mappings), self._f(self.vol), self._f(self.C), self._f(self.DmInv),
                                            self._f(self.pos), self._f(self.vel), self._f(self.F), self._f(self.forces), self._f(self.taus),
                                            self._cf(dt))

    # -- GpuMpmSolver::ParticleToGrid (cuda_mpm_solver.cu:86-105)
    fast_scatter = False  # True: the multi-core variant (cpu_baseline only)
    # True (the default): every node sum of ParticleToGrid is added in particle order by one thread -- the same bits on
    # every run and for every thread count.  False: OpenMP atomics in order of arrival (the reference's float atomics have
    # no order either); the parity tests keep one smoke variant each on that draw.  CalcFemStateAndForce's vertex forces
    # and the contact solve have a fixed order always (mpm_oracle.c header, "Determinism").
    ordered_scatter = True

    def particle_to_grid(self, dt: float):
        if self.g_cnt > 0:
            self.L.orc_clean_grid(C.c_uint32(self.g_cnt * 64), _u(self.g_ids), _u(self.g_flags), self._f(self.g_m),
                                  self._f(self.g_mv))
        fn = self.L.orc_particle_to_grid_colored if self.fast_scatter else (
            self.L.orc_particle_to_grid_ordered if self.ordered_scatter else self.L.orc_particle_to_grid)
        fn(C.byref(self.p), C.c_size_t(self.n_particles), self._f(self.pos), self._f(self.vel),
                                    self._f(self.vol), self._f(self.C), self._f(self.forces), self._f(self.taus), _u(self.g_flags),
                                    self._f(self.g_m), self._f(self.g_mv), self._cf(dt))

    # -- GpuMpmSolver::UpdateGrid (cuda_mpm_solver.cu:107-151)
    def update_grid(self, mpm_bc: int = -1):
        self.g_cnt = int(self.L.orc_gather_touched(C.c_uint32(self.n_blocks), _u(self.g_flags), _u(self.g_ids)))
        self.L.orc_update_grid(C.byref(self.p), C.c_int(mpm_bc), C.c_uint32(self.g_cnt * 64), _u(self.g_ids),
                               self._f(self.g_m), self._f(self.g_mv), self._f(self.g_vstar))

    def update_grid_table(self, colliders):
        """UpdateGrid with a runtime collider table (list of GridCollider)."""
        self.g_cnt = int(self.L.orc_gather_touched(C.c_uint32(self.n_blocks), _u(self.g_flags), _u(self.g_ids)))
        arr = (GridCollider * max(len(colliders), 1))(*colliders)
        self.L.orc_update_grid_table(C.byref(self.p), C.c_int(len(colliders)), arr, C.c_uint32(self.g_cnt * 64),
                                     _u(self.g_ids), self._f(self.g_m), self._f(self.g_mv), self._f(self.g_vstar))

    # -- GpuMpmSolver::GridToParticle (cuda_mpm_solver.cu:153-161)
    def grid_to_particle(self, dt: float):
        self.L.orc_grid_to_particle(C.byref(self.p), C.c_size_t(self.n_particles), self._f(self.pos), self._f(self.vel),
                                    self._f(self.C), self._f(self.g_m), self._f(self.g_mv), self._cf(dt), C.c_int(0))

    def substep(self, dt: float, mpm_bc: int = -1, sort: bool = False):
        """The five solver calls of cuda_mpm_test.cc:66-72."""
        self.rebuild_mapping(sort)
        self.calc_fem_state_and_force(dt)
        self.particle_to_grid(dt)
        self.update_grid(mpm_bc)
        self.grid_to_particle(dt)

    # -- GpuMpmState::ReallocateExternelBodies (cuda_mpm_model.cu:319-338)
    def reallocate_external_bodies(self, n: int):
        self.n_bodies = n
        self.F_tau = np.zeros((n, 3), self.real)
        self.F_f = np.zeros((n, 3), self.real)

    # -- GpuMpmSolver::CopyContactPairs (cuda_mpm_solver.cu:193-212)
    def copy_contact_pairs(self, pairs: ContactPairs):
        if pairs.dist.dtype != self.real:
            pairs = ContactPairs(pairs.particle, pairs.body, pairs.dist, pairs.normal, pairs.pos, pairs.rigid_v,
                                 pairs.rigid_p_WB, real=self.real)
        self.contacts = pairs
        nk = len(pairs)
        self.c_vel = np.zeros((nk, 3), self.real)
        self.c_vel0 = np.zeros((nk, 3), self.real)
        if nk:
            self.L.orc_initialize_contact_velocities(C.c_size_t(nk), self._f(self.c_vel), _u(pairs.particle), self._f(self.vel))

    # -- GpuMpmSolver::UpdateContact (cuda_mpm_solver.cu:214-621)
    def set_contact_relax(self, r: float):
        """Test hook: the Jacobi relaxation coefficient (0.3 in the reference, cuda_mpm_solver.cu:239)."""
        self.L.orc_set_contact_relax.argtypes = [self._creal]
        self.L.orc_set_contact_relax(float(r))

    def update_contact(self, dt, friction_mu, stiffness, damping, exact_line_search=False, max_iters=2000):
        pc = self.contacts
        nk = 0 if pc is None else len(pc)
        if nk == 0:
            return dict(iterations=0, residual=0.0, line_search_avg=0.0, energy=0.0)
        self._ensure_contact_grid()
        if self.n_bodies == 0:
            self.reallocate_external_bodies(int(pc.body.max()) + 1)
        res = self._creal(0)
        lsa = self._creal(0)
        en = self._creal(0)
        it = self.L.orc_update_contact(
            C.byref(self.p), C.c_size_t(nk), self._f(pc.pos), self._f(self.c_vel), self._f(self.c_vel0), self._f(self.vel), self._f(self.vol),
            _u(pc.particle), _u(pc.body), self._f(pc.dist), self._f(pc.normal), self._f(pc.rigid_v), self._f(pc.rigid_p_WB),
            C.c_uint32(self.g_cnt), _u(self.g_ids), self._f(self.g_m), self._f(self.g_mv), self._f(self.g_vstar), self._f(self.g_H),
            self._f(self.g_G), self._f(self.g_D), self._f(self.g_alpha), self._f(self.g_E0), self._f(self.g_E1), self._f(self.F_tau), self._f(self.F_f),
            self._cf(dt), self._cf(friction_mu), self._cf(stiffness), self._cf(damping), C.c_int(1 if exact_line_search else 0),
            C.c_int(max_iters), C.byref(res), C.byref(lsa), C.byref(en))
        diag = np.zeros(6, self.real)
        self.L.orc_last_contact_diag(self._f(diag))
        # one row per Newton iteration: alpha, E(0), E(alpha), sum |Dir|^2, DoFs, line-search evaluations, residual
        log = np.zeros((min(int(it), 4096), 7), self.real)
        self.L.orc_contact_iteration_log.restype = C.c_int
        self.L.orc_contact_iteration_log(self._f(log), C.c_int(log.shape[0]))
        self.contact_log = log.astype(np.float64)
        return dict(iterations=int(it), residual=float(res.value), line_search_avg=float(lsa.value),
                    energy=float(en.value), alpha=float(diag[0]), E0=float(diag[1]), E1=float(diag[2]),
                    norm_dir_sq=float(diag[3]), dofs=float(diag[4]), ls_last=int(diag[5]))

    # -- GpuMpmState::DumpCpuState (cuda_mpm_model.cu:244-265)
    def dump_cpu_state(self):
        orig = np.empty_like(self.pos)
        orig[self.pids] = self.pos
        return orig[self.n_faces:].copy(), (self.indices - self.n_faces).astype(np.int32)

    # helpers for tests ------------------------------------------------
    def state_in_original_order(self):
        """x, v, C, vol un-permuted to the Finalize() order [faces | verts]."""
        out = {}
        for name in ("pos", "vel", "C", "vol"):
            a = getattr(self, name)
            o = np.empty_like(a)
            o[self.pids] = a
            out[name] = o
        return out

    def touched_blocks(self):
        return np.sort(self.g_ids[: self.g_cnt].copy())
