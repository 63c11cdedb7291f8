"""ctypes binding of include/mpm_hip.h.

`GpuMpm` mirrors the reference's GpuMpmState<float> + GpuMpmSolver<float> pair
(multibody/gpu_mpm/cuda_mpm_model.cuh:37-260, cuda_mpm_solver.cuh:19-35) with
snake_case method names; argument meaning and call order are the reference's.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

_LIB = None


class MpmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mpm_hip error {code}: {msg}")
        self.code = code


class Collider(C.Structure):
    """mpm_collider_t: kind 0 half-space (z_B <= 0 inside), 1 sphere, 2 box (half extents), 3 capsule (z_B)."""
    _fields_ = [("kind", C.c_int32), ("body", C.c_uint32), ("p_WB", C.c_float * 3), ("R_WB", C.c_float * 9),
                ("dims", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3)]

    def __init__(self, kind, body=0, p_WB=(0, 0, 0), R_WB=None, dims=(0, 0, 0), v=(0, 0, 0), w=(0, 0, 0)):
        super().__init__()
        self.kind, self.body = int(kind), int(body)
        R = np.eye(3, dtype=np.float32) if R_WB is None else np.asarray(R_WB, np.float32).reshape(3, 3)
        self.p_WB[:] = [float(x) for x in p_WB]
        self.R_WB[:] = [float(x) for x in R.reshape(-1)]
        self.dims[:] = [float(x) for x in dims]
        self.v[:] = [float(x) for x in v]
        self.w[:] = [float(x) for x in w]


class GridCollider(C.Structure):
    """mpm_grid_collider_t: shape 0 sphere / 1 half-space; mode 0 fixed / 1 slip while approaching / 2 slip."""
    _fields_ = [("shape", C.c_int32), ("mode", C.c_int32), ("p", C.c_float * 3), ("n", C.c_float * 3),
                ("radius", C.c_float), ("v", C.c_float * 3), ("friction", C.c_float)]

    def __init__(self, shape=0, mode=0, p=(0, 0, 0), n=(0, 0, 1), radius=0.0, v=(0, 0, 0), friction=-1.0):
        super().__init__()
        self.shape, self.mode, self.radius, self.friction = int(shape), int(mode), float(radius), float(friction)
        self.p[:] = [float(x) for x in p]
        self.n[:] = [float(x) for x in n]
        self.v[:] = [float(x) for x in v]


BC_TABLE = 4   # mpm_bc value that selects the table of set_grid_colliders


def grid_collider_preset(mpm_bc: int, sdf_friction: float = 0.3):
    """The collider table that reproduces the reference's scene mpm_bc (cuda_mpm_kernels.cuh:673-774)."""
    lib = load_library()
    arr = (GridCollider * 16)()
    n = C.c_size_t()
    rc = lib.mpm_grid_collider_preset(mpm_bc, C.c_float(sdf_friction), arr, 16, C.byref(n))
    if rc:
        raise MpmError(rc, (lib.mpm_last_error() or b"").decode())
    return [arr[k] for k in range(n.value)]


class Material(C.Structure):
    _fields_ = [
        ("youngs_modulus", C.c_float), ("poisson_ratio", C.c_float), ("density", C.c_float), ("gamma", C.c_float),
        ("K", C.c_float), ("V", C.c_float), ("c_F", C.c_float), ("sdf_friction", C.c_float), ("gravity", C.c_float),
        ("epsv", C.c_float), ("gravity_axis", C.c_int32), ("wall_cells", C.c_int32),
    ]


class ContactStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("line_search_evals", C.c_int32), ("contacts", C.c_uint32),
                ("nodes", C.c_uint32), ("residual", C.c_float), ("alpha", C.c_float), ("energy", C.c_float),
                ("E0", C.c_float), ("norm_dir_sq", C.c_float), ("dofs", C.c_float)]


class CoupledParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("mpm_bc", C.c_int32), ("friction_mu", C.c_float), ("stiffness", C.c_float),
                ("damping", C.c_float), ("exact_line_search", C.c_int32), ("max_newton_iterations", C.c_int32)]


class CoupledResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("contacts", C.c_uint32), ("nodes", C.c_uint32), ("residual", C.c_float),
                ("setup_reused", C.c_int32)]


class DistConfig(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("own_lo_block", C.c_int32), ("own_hi_block", C.c_int32),
                ("left_lo_block", C.c_int32), ("right_hi_block", C.c_int32), ("zone_blocks", C.c_int32),
                ("ghost_cells", C.c_int32), ("ghost_margin_cells", C.c_int32)]


class DistGeometry(C.Structure):
    _fields_ = [("face_band_cells", C.c_float), ("vertex_band_cells", C.c_float), ("drift_budget_cells", C.c_float),
                ("longest_edge_cells", C.c_float), ("slot_resizes", C.c_uint32), ("migrations", C.c_uint32),
                ("retunes", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("substeps", C.c_uint64), ("rebuilds", C.c_uint64), ("home_blocks", C.c_uint32), ("active_blocks", C.c_uint32),
        ("touched_blocks", C.c_uint32), ("error_flags", C.c_uint32), ("active_faces", C.c_uint32),
        ("active_vertices", C.c_uint32), ("face_slots", C.c_uint32), ("vertex_slots", C.c_uint32),
        ("particle_bytes", C.c_uint64), ("scene_index_bytes", C.c_uint64), ("resort_checks", C.c_uint64),
        ("quiet_time_s", C.c_float), ("since_resort_s", C.c_float),
    ]


class ARR:
    POSITIONS, VELOCITIES, VOLUMES, AFFINE, PIDS, INDEX_MAPPINGS, SORT_KEYS, FORCES, TAUS = range(9)
    DEFORMATION_GRADIENTS, DM_INVERSES, INDICES, GRID_MASSES, GRID_MOMENTUM, GRID_V_STAR = range(9, 15)
    GRID_TOUCHED_FLAGS, GRID_TOUCHED_IDS, CONTACT_VEL, CONTACT_VEL0, GRID_DIR = range(15, 20)


PHASES = ("rebuild", "fem", "vforce", "p2g", "grid", "g2p")

# every symbol include/mpm_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "mpm_last_error", "mpm_default_material", "mpm_create", "mpm_add_qr_cloth", "mpm_finalize", "mpm_destroy",
    "mpm_counts", "mpm_grid_touched_cnt", "mpm_dump_cpu_state", "mpm_reallocate_external_bodies",
    "mpm_external_body_force_to_host", "mpm_rebuild_mapping", "mpm_calc_fem_state_and_force", "mpm_particle_to_grid",
    "mpm_update_grid", "mpm_grid_to_particle", "mpm_sync", "mpm_sync_particle_state_to_cpu", "mpm_dump_obj",
    "mpm_copy_contact_pairs", "mpm_generate_contact_pairs", "mpm_download_contact_pairs", "mpm_update_contact", "mpm_set_dump_dir", "mpm_substep", "mpm_run_substeps",
    "mpm_profile_substeps", "mpm_set_stream", "mpm_set_deterministic", "mpm_get_stats", "mpm_debug_counters", "mpm_grid_gather",
    "mpm_halo_buffer_bytes", "mpm_halo_pack", "mpm_halo_add", "mpm_update_grid_from_sums", "mpm_substep_begin",
    "mpm_substep_end", "mpm_substep_begin_halo", "mpm_substep_mid_halo", "mpm_substep_end_halo", "mpm_chain_unique_id",
    "mpm_chain_init", "mpm_chain_substeps", "mpm_chain_destroy", "mpm_download_array", "mpm_upload_particle_state",
    "mpm_newton_bisect_f64", "mpm_newton_bisect_f32", "mpm_finalize_external_contact_forces",
    "mpm_spatial_force_shift", "mpm_external_forces_at_body_origin", "mpm_set_grid_colliders",
    "mpm_grid_collider_preset", "mpm_get_contact_stats", "mpm_dist_init", "mpm_dist_migration_buffer_bytes",
    "mpm_dist_migrate_pack", "mpm_dist_migrate_apply", "mpm_dist_roles", "mpm_chain_enable_migration",
    "mpm_dist_set_transport", "mpm_device_synchronize", "mpm_debug_owed_substeps",
    "mpm_memcpy_d2h", "mpm_memcpy_h2d", "mpm_profile_contact_iteration", "mpm_contact_frame", "mpm_halo_zone_blocks",
    "mpm_dist_get_geometry", "mpm_dist_set_headroom", "mpm_dist_migration_quiet_time", "mpm_dist_retune",
    "mpm_dist_plan_migration", "mpm_debug_throw", "mpm_debug_fail_alloc", "mpm_set_fast_math", "mpm_get_fast_math",
    "mpm_get_contact_pair_count", "mpm_download_contact_log", "mpm_last_contact_counts",
    "mpm_debug_contact_counters", "mpm_run_coupled_substeps", "mpm_chain_direct_prepare", "mpm_chain_direct_connect",
    "mpm_debug_contact_count", "mpm_chain_direct_base", "mpm_chain_direct_connect_local", "mpm_team_prepare", "mpm_team_connect",
    "mpm_world_coupled_substeps",
]

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)

ROOTFIND_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double))


def library_path() -> str:
    return _build.LIB


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheels carry their own libamdhip64.so (soname
    libamdhip64.so.7, like /opt/rocm's, but torch asks for it as "libamdhip64.so"): loaded after this
    library, torch would bring in a second runtime next to the one libmpm_hip.so is bound to, and the two
    tear each other's state down at exit ("double free or corruption").  When torch is installed its copy
    is loaded first, so that both resolve to it whichever is imported first; torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    path = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    # the same for RCCL, which the engine binds lazily (csrc/mpm_chain.h): point it at torch's copy
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("MPM_RCCL_LIBRARY", rccl)


def load_library(build: bool = True):
    """Loads drake_amd/libmpm_hip.so, building it with hipcc first if needed.
    Raises if the library cannot be produced: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.build() if build else _build.LIB
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -m drake_amd._build` (no CPU fallback exists)")
    _preload_hip_runtime()
    lib = C.CDLL(path)
    lib.mpm_last_error.restype = C.c_char_p
    vp, f, i, sz = C.c_void_p, C.c_float, C.c_int, C.c_size_t
    P = C.POINTER
    sigs = {
        "mpm_default_material": [P(Material)],
        "mpm_create": [i, P(Material), i, P(vp)],
        "mpm_add_qr_cloth": [vp, vp, vp, sz, vp, sz],
        "mpm_finalize": [vp], "mpm_destroy": [vp],
        "mpm_counts": [vp, P(sz), P(sz), P(sz)],
        "mpm_grid_touched_cnt": [vp, P(C.c_uint32)],
        "mpm_dump_cpu_state": [vp, vp, vp],
        "mpm_reallocate_external_bodies": [vp, sz],
        "mpm_external_body_force_to_host": [vp, vp, vp],
        "mpm_rebuild_mapping": [vp, i],
        "mpm_calc_fem_state_and_force": [vp, f],
        "mpm_particle_to_grid": [vp, f],
        "mpm_update_grid": [vp, i],
        "mpm_grid_to_particle": [vp, f],
        "mpm_sync": [vp],
        "mpm_debug_owed_substeps": [vp, P(C.c_uint32)],
        "mpm_memcpy_d2h": [vp, vp, vp, sz],
        "mpm_profile_contact_iteration": [vp, i, P(C.c_float)],
        "mpm_halo_zone_blocks": [vp, i, i, P(C.c_uint32)],
        "mpm_memcpy_h2d": [vp, vp, vp, sz],
        "mpm_sync_particle_state_to_cpu": [vp, vp],
        "mpm_dump_obj": [vp, C.c_char_p],
        "mpm_copy_contact_pairs": [vp, sz, vp, vp, vp, vp, vp, vp, vp],
        "mpm_update_contact": [vp, i, i, f, f, f, f, i, i, i, P(i), P(f)],
        "mpm_set_dump_dir": [vp, C.c_char_p],
        "mpm_substep": [vp, f, i],
        "mpm_run_substeps": [vp, i, f, i],
        "mpm_profile_substeps": [vp, i, f, i, P(f), P(f)],
        "mpm_set_stream": [vp, vp],
        "mpm_get_stats": [vp, P(Stats)],
        "mpm_grid_gather": [vp],
        "mpm_halo_pack": [vp, i, i, i, vp, sz],
        "mpm_halo_add": [vp, vp, sz],
        "mpm_update_grid_from_sums": [vp, i],
        "mpm_substep_begin": [vp, f],
        "mpm_substep_end": [vp, f, i],
        "mpm_generate_contact_pairs": [vp, sz, vp, P(sz)],
        "mpm_download_contact_pairs": [vp, vp, vp, vp, vp, vp, vp, vp],
        "mpm_set_deterministic": [vp, i],
        "mpm_set_fast_math": [vp, i],
        "mpm_get_contact_pair_count": [vp, P(sz)],
        "mpm_last_contact_counts": [vp, P(C.c_uint32), P(C.c_uint32), P(i)],
        "mpm_download_contact_log": [vp, vp, sz, P(sz)],
        "mpm_debug_contact_counters": [vp, P(C.c_uint64)],
        "mpm_debug_contact_count": [vp, i, i],
        "mpm_chain_direct_base": [vp, P(vp)],
        "mpm_chain_direct_connect_local": [vp, vp, vp],
        "mpm_team_prepare": [vp, sz, vp, P(vp)],
        "mpm_team_connect": [vp, vp, P(vp)],
        "mpm_world_coupled_substeps": [P(vp), i, i, vp, sz, vp, P(vp)],
        "mpm_run_coupled_substeps": [vp, i, vp, sz, vp, vp],
        "mpm_get_fast_math": [vp, P(i)],
        "mpm_substep_begin_halo": [vp, f, i, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_void_p), sz],
        "mpm_substep_end_halo": [vp, f, i, i, P(C.c_void_p), sz],
        "mpm_substep_mid_halo": [vp, f, i],
        "mpm_chain_unique_id": [vp],
        "mpm_chain_init": [vp, vp, i, i, i, i, i, i, sz, i],
        "mpm_chain_substeps": [vp, i, f, i],
        "mpm_chain_destroy": [vp],
        "mpm_chain_direct_prepare": [vp, vp],
        "mpm_chain_direct_connect": [vp, vp, vp],
        "mpm_debug_counters": [vp, P(C.c_uint64), i],
        "mpm_download_array": [vp, i, vp, sz, P(sz)],
        "mpm_upload_particle_state": [vp, vp, vp, vp, vp, vp],
        "mpm_finalize_external_contact_forces": [vp, f, vp, vp],
        "mpm_spatial_force_shift": [sz, vp, vp, vp, vp],
        "mpm_set_grid_colliders": [vp, sz, vp],
        "mpm_get_contact_stats": [vp, P(ContactStats)],
        "mpm_dist_init": [vp, P(DistConfig)],
        "mpm_dist_migrate_pack": [vp, vp, vp, sz],
        "mpm_dist_migrate_apply": [vp, vp, vp, sz],
        "mpm_dist_roles": [vp, vp],
        "mpm_dist_get_geometry": [vp, P(DistGeometry)],
        "mpm_dist_set_headroom": [vp, f],
        "mpm_dist_migration_quiet_time": [vp, P(f)],
        "mpm_dist_retune": [vp, f, f, P(i)],
        "mpm_dist_plan_migration": [vp, vp, sz, sz, sz, sz, sz, sz, sz, f, P(sz)],
        "mpm_debug_throw": [i],
        "mpm_debug_fail_alloc": [vp, i],
        "mpm_chain_enable_migration": [vp, i, sz],
        "mpm_dist_set_transport": [vp, EXCHANGE_FN, ALLREDUCE_FN, vp, sz],
        "mpm_grid_collider_preset": [i, f, vp, sz, P(sz)],
        "mpm_external_forces_at_body_origin": [sz, vp, vp, vp, vp, vp],
    }
    for name, args in sigs.items():
        if os.environ.get("MPM_HIP_LIBRARY") and not hasattr(lib, name):
            continue   # (an older A/B variant of an experiment, scratch/ab_run.py)
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    d = C.c_double
    lib.mpm_newton_bisect_f64.argtypes = [ROOTFIND_FN, vp, d, d, d, d, d, i, i, P(d), P(i)]
    lib.mpm_newton_bisect_f64.restype = i
    lib.mpm_newton_bisect_f32.argtypes = [ROOTFIND_FN, vp, f, f, f, f, f, i, i, P(f), P(i)]
    lib.mpm_newton_bisect_f32.restype = i
    lib.mpm_halo_buffer_bytes.argtypes = [sz]
    if hasattr(lib, "mpm_device_synchronize"):
        lib.mpm_device_synchronize.argtypes = []
        lib.mpm_device_synchronize.restype = C.c_int
    lib.mpm_dist_migration_buffer_bytes.argtypes = [sz]
    lib.mpm_dist_migration_buffer_bytes.restype = sz
    lib.mpm_halo_buffer_bytes.restype = sz
    _LIB = lib
    return lib


def contact_frame(u):
    """Rows (tangent, tangent, u) of the contact frame the solver builds for the unit normal u (host build of the
    kernels' own inline function)."""
    u = _f32(u, (3,))
    J = np.zeros(9, np.float32)
    lib = load_library()
    lib.mpm_contact_frame.argtypes = [C.c_void_p, C.c_void_p]
    rc = lib.mpm_contact_frame(_ptr(u), _ptr(J))
    if rc:
        raise MpmError(rc, (lib.mpm_last_error() or b"").decode())
    return J.reshape(3, 3)


def spatial_force_shift(tau, f, offset):
    """SpatialForce::Shift for n forces (spatial_force.h:91-93): tau - offset x f."""
    tau, f, offset = (_f32(a, (-1, 3)) for a in (tau, f, offset))
    out = np.zeros_like(tau)
    lib = load_library()
    rc = lib.mpm_spatial_force_shift(tau.shape[0], _ptr(tau), _ptr(f), _ptr(offset), _ptr(out))
    if rc:
        raise MpmError(rc, (lib.mpm_last_error() or b"").decode())
    return out


def external_forces_at_body_origin(R_WB, p_BoBq_B, tau, f):
    """AddAppliedExternalSpatialForces (multibody_plant.cc:2385-2407): torque shifted to the body origin."""
    R = _f32(R_WB, (-1, 9))
    p, tau, f = (_f32(a, (-1, 3)) for a in (p_BoBq_B, tau, f))
    out = np.zeros_like(tau)
    lib = load_library()
    rc = lib.mpm_external_forces_at_body_origin(tau.shape[0], _ptr(R), _ptr(p), _ptr(tau), _ptr(f), _ptr(out))
    if rc:
        raise MpmError(rc, (lib.mpm_last_error() or b"").decode())
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


_LIVE = None   # engines alive: destroyed before the interpreter (and with it the HIP runtime) goes down


def _destroy_live():
    for g in list(_LIVE or ()):
        try:
            g.destroy()
        except Exception:  # noqa: BLE001
            pass


class GpuMpm:
    """One GpuMpmState + the GpuMpmSolver calls that act on it."""

    def __init__(self, domain_bits: int = 7, material: Material | None = None, device: int = 0):
        global _LIVE
        if _LIVE is None:
            import atexit
            import weakref
            _LIVE = weakref.WeakSet()
            atexit.register(_destroy_live)
        self.lib = load_library()
        self.h = C.c_void_p()
        _LIVE.add(self)
        self.domain_bits = domain_bits
        self.n_cells = 1 << (3 * domain_bits)
        self.n_blocks = self.n_cells >> 6
        self._ck(self.lib.mpm_create(domain_bits, C.byref(material) if material is not None else None, device,
                                     C.byref(self.h)))

    def _ck(self, rc: int):
        if rc != 0:
            raise MpmError(rc, (self.lib.mpm_last_error() or b"").decode())

    @staticmethod
    def default_material() -> Material:
        m = Material()
        load_library().mpm_default_material(C.byref(m))
        return m

    # ---- GpuMpmState --------------------------------------------------------
    def add_qr_cloth(self, pos, vel, indices):
        pos = _f32(pos, (-1, 3))
        vel = _f32(vel, (-1, 3))
        idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        self._ck(self.lib.mpm_add_qr_cloth(self.h, _ptr(pos), _ptr(vel), pos.shape[0], _ptr(idx), idx.size // 3))

    def finalize(self):
        self._ck(self.lib.mpm_finalize(self.h))
        nv, nf, n = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._ck(self.lib.mpm_counts(self.h, C.byref(nv), C.byref(nf), C.byref(n)))
        self.n_verts, self.n_faces, self.n_particles = nv.value, nf.value, n.value

    def destroy(self):
        if self.h:
            self.lib.mpm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def grid_touched_cnt(self) -> int:
        c = C.c_uint32()
        self._ck(self.lib.mpm_grid_touched_cnt(self.h, C.byref(c)))
        return int(c.value)

    def dump_cpu_state(self):
        pos = np.empty((self.n_verts, 3), np.float32)
        idx = np.empty(self.n_faces * 3, np.int32)
        self._ck(self.lib.mpm_dump_cpu_state(self.h, _ptr(pos), _ptr(idx)))
        return pos, idx

    def reallocate_external_bodies(self, n: int):
        self._ck(self.lib.mpm_reallocate_external_bodies(self.h, n))
        self._n_bodies = n

    def external_body_force_to_host(self):
        n = getattr(self, "_n_bodies", 0)
        tau = np.zeros((n, 3), np.float32)
        frc = np.zeros((n, 3), np.float32)
        self._ck(self.lib.mpm_external_body_force_to_host(self.h, _ptr(tau), _ptr(frc)))
        return tau, frc

    def finalize_external_contact_forces(self, dt: float):
        """FinalizeExternalContactForces (deformable_driver.h:210-219): (tau, f) forces of the plant step dt."""
        n = getattr(self, "_n_bodies", 0)
        tau = np.zeros((n, 3), np.float32)
        frc = np.zeros((n, 3), np.float32)
        self._ck(self.lib.mpm_finalize_external_contact_forces(self.h, dt, _ptr(tau), _ptr(frc)))
        return tau, frc

    # ---- GpuMpmSolver -------------------------------------------------------
    def rebuild_mapping(self, sort: bool = False):
        self._ck(self.lib.mpm_rebuild_mapping(self.h, 1 if sort else 0))

    def calc_fem_state_and_force(self, dt: float):
        self._ck(self.lib.mpm_calc_fem_state_and_force(self.h, dt))

    def particle_to_grid(self, dt: float):
        self._ck(self.lib.mpm_particle_to_grid(self.h, dt))

    def update_grid(self, mpm_bc: int = -1):
        self._ck(self.lib.mpm_update_grid(self.h, mpm_bc))

    def set_grid_colliders(self, colliders):
        """Table of analytic grid colliders used by mpm_bc = BC_TABLE (list of GridCollider)."""
        arr = (GridCollider * max(len(colliders), 1))(*colliders)
        self._ck(self.lib.mpm_set_grid_colliders(self.h, len(colliders), arr))

    def grid_to_particle(self, dt: float):
        self._ck(self.lib.mpm_grid_to_particle(self.h, dt))

    def gpu_sync(self):
        """GpuSync(state): waits for this engine's stream, runs what batched calls still owe and REPORTS the simulation's
        sticky errors (DRIFT, DOMAIN, HALO, RANGE, CAPACITY) -- the call to poll for divergence."""
        self._ck(self.lib.mpm_sync(self.h))

    @staticmethod
    def device_synchronize():
        """GpuMpmSolver::GpuSync() as the reference calls it: no state argument (cuda_mpm_solver.cu:164-166).  Like
        cudaDeviceSynchronize it reports runtime failures only: a simulation's sticky error flags are polled per engine,
        with gpu_sync() or stats()["error_flags"]."""
        lib = load_library()
        rc = lib.mpm_device_synchronize()
        if rc:
            raise MpmError(rc, (lib.mpm_last_error() or b"").decode())

    def memcpy_d2h(self, dst_host_ptr: int, src_device_ptr: int, nbytes: int):
        self._ck(self.lib.mpm_memcpy_d2h(self.h, dst_host_ptr, src_device_ptr, nbytes))

    def memcpy_h2d(self, dst_device_ptr: int, src_host_ptr: int, nbytes: int):
        self._ck(self.lib.mpm_memcpy_h2d(self.h, dst_device_ptr, src_host_ptr, nbytes))

    def profile_contact_iteration(self, reps: int = 20):
        """ms per launch of (k_ct_tile, k_ct_node_dir, k_ct_ls, k_ct_decide) on the last solve's state."""
        ms = (C.c_float * 4)()
        self._ck(self.lib.mpm_profile_contact_iteration(self.h, reps, ms))
        return dict(zip(("k_ct_tile", "k_ct_node_dir", "k_ct_ls", "k_ct_decide"), (float(x) for x in ms)))

    def owed_substeps(self) -> int:
        n = C.c_uint32()
        self._ck(self.lib.mpm_debug_owed_substeps(self.h, C.byref(n)))
        return n.value

    def sync_particle_state_to_cpu(self):
        pos = np.empty((self.n_particles, 3), np.float32)
        self._ck(self.lib.mpm_sync_particle_state_to_cpu(self.h, _ptr(pos)))
        return pos

    def set_dump_dir(self, path: str):
        """Directory of the solver-statistics JSON written by update_contact(dump=True)."""
        self._ck(self.lib.mpm_set_dump_dir(self.h, path.encode()))

    def dump(self, filename: str):
        self._ck(self.lib.mpm_dump_obj(self.h, filename.encode()))

    def copy_contact_pairs(self, particle, body, dist, normal, pos, rigid_v, rigid_p_WB):
        particle = np.ascontiguousarray(particle, dtype=np.uint32)
        body = np.ascontiguousarray(body, dtype=np.uint32)
        n = int(body.shape[0])
        arrs = [_f32(dist), _f32(normal, (-1, 3)), _f32(pos, (-1, 3)), _f32(rigid_v, (-1, 3)), _f32(rigid_p_WB, (-1, 3))]
        self._ck(self.lib.mpm_copy_contact_pairs(self.h, n, _ptr(particle), _ptr(body), *[_ptr(a) for a in arrs]))
        self._n_contacts = n

    def generate_contact_pairs(self, colliders, want_count: bool = True):
        """Device-side CalcMpmContactPairs + CopyContactPairs for analytic colliders; returns the pair count -- or, with
        want_count = False, None without waiting for anything: the count stays on the device, update_contact reports it
        (contact_stats()["contacts"]), contact_pair_count() reads it back."""
        if isinstance(colliders, C.Array):
            arr, nc = colliders, len(colliders)
        else:
            arr, nc = (Collider * max(len(colliders), 1))(*colliders), len(colliders)
        if not want_count:
            self._ck(self.lib.mpm_generate_contact_pairs(self.h, nc, arr, None))
            self._n_contacts = None
            return None
        n = C.c_size_t()
        self._ck(self.lib.mpm_generate_contact_pairs(self.h, nc, arr, C.byref(n)))
        self._n_contacts = int(n.value)
        return self._n_contacts

    def contact_pair_count(self) -> int:
        n = C.c_size_t()
        self._ck(self.lib.mpm_get_contact_pair_count(self.h, C.byref(n)))
        self._n_contacts = int(n.value)
        return self._n_contacts

    def run_coupled_substeps(self, n, dt, colliders, friction_mu, stiffness, damping, mpm_bc=-1, exact_line_search=False,
                             max_newton_iterations=0):
        """n times the body of DeformableDriver::CalcAbstractStates' substep loop (deformable_driver.h:240-258) for analytic
        colliders, in one call (mpm_run_coupled_substeps); returns the per-substep results."""
        if isinstance(colliders, C.Array):
            arr, nc = colliders, len(colliders)
        else:
            arr, nc = (Collider * max(len(colliders), 1))(*colliders), len(colliders)
        prm = CoupledParams(dt, mpm_bc, friction_mu, stiffness, damping, 1 if exact_line_search else 0, max_newton_iterations)
        res = (CoupledResult * max(n, 1))()
        self._ck(self.lib.mpm_run_coupled_substeps(self.h, n, C.byref(prm), nc, arr, res))
        out = [dict(iterations=r.iterations, contacts=r.contacts, nodes=r.nodes, residual=r.residual,
                    setup_reused=bool(r.setup_reused)) for r in res[:n]]
        if out:
            self._n_contacts = out[-1]["contacts"]
        return out

    @staticmethod
    def world_coupled_substeps(engines, n, dt, colliders, friction_mu, stiffness, damping, mpm_bc=-1, exact_line_search=False,
                               max_newton_iterations=0):
        """n coupled substeps of an in-process world (the ranks of ONE partition in this process, all on one stream):
        mpm_world_coupled_substeps; -> per rank the list of per-substep result dicts"""
        lib = engines[0].lib
        if isinstance(colliders, C.Array):
            arr, nc = colliders, len(colliders)
        else:
            arr, nc = (Collider * max(len(colliders), 1))(*colliders), len(colliders)
        prm = CoupledParams(dt, mpm_bc, friction_mu, stiffness, damping, 1 if exact_line_search else 0, max_newton_iterations)
        hs = (C.c_void_p * len(engines))(*[e.h for e in engines])
        res = [(CoupledResult * max(n, 1))() for _ in engines]
        ptrs = (C.c_void_p * len(engines))(*[C.cast(r, C.c_void_p) for r in res])
        engines[0]._ck(lib.mpm_world_coupled_substeps(hs, len(engines), n, C.byref(prm), nc, arr, ptrs))
        return [[dict(iterations=r.iterations, contacts=r.contacts, nodes=r.nodes, residual=r.residual) for r in rr[:n]] for rr in res]

    def contact_counters(self) -> dict:
        out = (C.c_uint64 * 6)()
        self._ck(self.lib.mpm_debug_contact_counters(self.h, out))
        return dict(solves=int(out[0]), reused=int(out[1]), refused_stale=int(out[2]), repeated_overflow=int(out[3]),
                    contact_free=int(out[4]), contact_free_repeated=int(out[5]))

    def debug_contact_count(self, count: int = -1, stamp_delta: int = 0):
        """tests: tamper with the pair count that generate_contact_pairs(want_count=False) left on the device"""
        self._ck(self.lib.mpm_debug_contact_count(self.h, count, stamp_delta))

    def contact_log(self):
        """rows (residual, line-search evaluations, E(alpha), alpha, E(0), sum |Dir|^2, DoFs, 0) of the last solve's Newton
        iterations (mpm_download_contact_log)"""
        n = C.c_size_t()
        self._ck(self.lib.mpm_download_contact_log(self.h, None, 0, C.byref(n)))
        out = np.zeros((int(n.value), 8), np.float32)
        if n.value:
            self._ck(self.lib.mpm_download_contact_log(self.h, _ptr(out), out.shape[0], C.byref(n)))
        return out

    def download_contact_pairs(self):
        """(particle, body, dist, normal, pos, rigid_v, rigid_p_WB) of the pairs currently in the engine."""
        n = getattr(self, "_n_contacts", 0)
        if n is None:
            n = self.contact_pair_count()
        particle, body = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        dist = np.zeros(n, np.float32)
        v3 = [np.zeros((n, 3), np.float32) for _ in range(4)]
        self._ck(self.lib.mpm_download_contact_pairs(self.h, _ptr(particle), _ptr(body), _ptr(dist),
                                                     *[_ptr(a) for a in v3]))
        return (particle, body, dist, *v3)

    def update_contact(self, dt, friction_mu, stiffness, damping, exact_line_search=False, frame=0, substep=0,
                       dump=False, max_newton_iterations=0):
        it = C.c_int()
        res = C.c_float()
        self._ck(self.lib.mpm_update_contact(self.h, frame, substep, dt, friction_mu, stiffness, damping,
                                             1 if dump else 0, 1 if exact_line_search else 0, max_newton_iterations,
                                             C.byref(it), C.byref(res)))
        nc, nn, reused = C.c_uint32(), C.c_uint32(), C.c_int()
        self.lib.mpm_last_contact_counts(self.h, C.byref(nc), C.byref(nn), C.byref(reused))
        self._n_contacts = int(nc.value)
        return dict(iterations=int(it.value), residual=float(res.value), contacts=int(nc.value), nodes=int(nn.value),
                    setup_reused=bool(reused.value))

    def contact_stats(self) -> dict:
        s = ContactStats()
        self._ck(self.lib.mpm_get_contact_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in ContactStats._fields_}

    # ---- conveniences -------------------------------------------------------
    def substep(self, dt: float, mpm_bc: int = -1):
        self._ck(self.lib.mpm_substep(self.h, dt, mpm_bc))

    def run_substeps(self, n: int, dt: float, mpm_bc: int = -1):
        self._ck(self.lib.mpm_run_substeps(self.h, n, dt, mpm_bc))

    def profile_substeps(self, n: int, dt: float, mpm_bc: int = -1):
        ph = (C.c_float * len(PHASES))()
        tot = C.c_float()
        self._ck(self.lib.mpm_profile_substeps(self.h, n, dt, mpm_bc, ph, C.byref(tot)))
        return {k: float(ph[i]) for i, k in enumerate(PHASES)}, float(tot.value)

    # ---- multi-GPU halo (see drake_amd/dist.py) --------------------------------
    def grid_gather(self):
        self._ck(self.lib.mpm_grid_gather(self.h))

    def halo_buffer_bytes(self, capacity_blocks: int) -> int:
        return int(self.lib.mpm_halo_buffer_bytes(capacity_blocks))

    def halo_zone_blocks(self, bx_lo: int, bx_hi: int) -> int:
        """Active blocks with x block coordinate in [bx_lo, bx_hi]: what a halo pack of that zone holds right now."""
        n = C.c_uint32()
        self._ck(self.lib.mpm_halo_zone_blocks(self.h, bx_lo, bx_hi, C.byref(n)))
        return n.value

    def halo_pack(self, bx_lo: int, bx_hi: int, shift_bx: int, dev_ptr: int, capacity_blocks: int):
        self._ck(self.lib.mpm_halo_pack(self.h, bx_lo, bx_hi, shift_bx, C.c_void_p(dev_ptr), capacity_blocks))

    def halo_add(self, dev_ptr: int, capacity_blocks: int):
        self._ck(self.lib.mpm_halo_add(self.h, C.c_void_p(dev_ptr), capacity_blocks))

    def substep_begin(self, dt: float):
        self._ck(self.lib.mpm_substep_begin(self.h, dt))

    def substep_end(self, dt: float, mpm_bc: int = -1):
        self._ck(self.lib.mpm_substep_end(self.h, dt, mpm_bc))

    @staticmethod
    def halo_zone_args(zones, send_ptrs):
        """ctypes argument pack for substep_begin_halo: zones = [(bx_lo, bx_hi, shift_bx)], built once."""
        n = len(zones)
        arr = lambda vals: (C.c_int * max(n, 1))(*vals)
        return (n, arr([z[0] for z in zones]), arr([z[1] for z in zones]), arr([z[2] for z in zones]),
                (C.c_void_p * max(n, 1))(*send_ptrs))

    @staticmethod
    def halo_buffer_args(ptrs):
        return (len(ptrs), (C.c_void_p * max(len(ptrs), 1))(*ptrs))

    def substep_begin_halo(self, dt: float, zone_args, capacity_blocks: int):
        n, lo, hi, sh, bufs = zone_args
        self._ck(self.lib.mpm_substep_begin_halo(self.h, dt, n, lo, hi, sh, bufs, capacity_blocks))

    # ---- one domain cut into x slabs (mpm_dist_*) -------------------------------------
    def dist_init(self, rank: int, world: int, cuts, zone_blocks: int = 2, ghost_cells: int = 2,
                  ghost_margin_cells: int = 2, headroom: float | None = None):
        """cuts: world + 1 ascending x block indices; rank r owns blocks [cuts[r], cuts[r + 1]).
        ghost_cells = ghost_margin_cells = 0: band widths from the mesh (see mpm_dist_init).  headroom: slot space
        of the rank as a multiple of what it holds (None: the library's default 1.5; 0: no shrink)."""
        assert len(cuts) == world + 1
        if headroom is not None:
            self._ck(self.lib.mpm_dist_set_headroom(self.h, float(headroom)))
        cfg = DistConfig(rank, world, cuts[rank], cuts[rank + 1], cuts[rank - 1] if rank > 0 else 0,
                         cuts[rank + 2] if rank + 2 <= world else cuts[world], zone_blocks, ghost_cells,
                         ghost_margin_cells)
        self._ck(self.lib.mpm_dist_init(self.h, C.byref(cfg)))

    def dist_geometry(self) -> dict:
        g = DistGeometry()
        self._ck(self.lib.mpm_dist_get_geometry(self.h, C.byref(g)))
        return {k: getattr(g, k) for k, _ in DistGeometry._fields_}

    def dist_migration_quiet_time(self) -> float:
        """Seconds for which, by this rank's ballistic estimate at its last mpm_dist_migrate_pack, no particle it holds
        drifts further along x than the bands allow (may be inf)."""
        t = C.c_float(0.0)
        self._ck(self.lib.mpm_dist_migration_quiet_time(self.h, C.byref(t)))
        return float(t.value)

    def dist_retune(self, quiet_time_all: float, dt: float) -> bool:
        """Re-sizes bands that came from the mesh for the speed the ranks' common estimate implies (mpm_dist_retune)."""
        changed = C.c_int(0)
        self._ck(self.lib.mpm_dist_retune(self.h, float(quiet_time_all), float(dt), C.byref(changed)))
        return bool(changed.value)

    def dist_migration_buffer_bytes(self, capacity_particles: int) -> int:
        return int(self.lib.mpm_dist_migration_buffer_bytes(capacity_particles))

    def dist_migrate_pack(self, send_left_ptr: int, send_right_ptr: int, capacity_particles: int):
        self._ck(self.lib.mpm_dist_migrate_pack(self.h, C.c_void_p(send_left_ptr), C.c_void_p(send_right_ptr),
                                                capacity_particles))

    def dist_migrate_apply(self, recv_left_ptr, recv_right_ptr, capacity_particles: int):
        self._ck(self.lib.mpm_dist_migrate_apply(self.h, C.c_void_p(recv_left_ptr) if recv_left_ptr else None,
                                                 C.c_void_p(recv_right_ptr) if recv_right_ptr else None,
                                                 capacity_particles))

    def debug_fail_alloc(self, nth: int):
        """tests: the engine's nth device allocation from now fails like an exhausted device (0 = off)"""
        self._ck(self.lib.mpm_debug_fail_alloc(self.h, int(nth)))

    def dist_set_transport(self, exchange, allreduce, zone_capacity_blocks: int = 1024):
        """Transport callbacks of the distributed contact solve (see mpm_dist_set_transport):
        exchange(send_left, send_right, recv_left, recv_right, nbytes) with device addresses (ints),
        allreduce(numpy float64 array) summing in place over the ranks."""
        def _ex(_user, sl, sr, rl, rr, nbytes):
            try:
                exchange(sl, sr, rl, rr, int(nbytes))
                return 0
            except Exception as exc:  # noqa: BLE001
                print(f"[drake_amd] exchange callback failed: {exc!r}")
                return 1

        def _ar(_user, values, n):
            try:
                a = np.ctypeslib.as_array(values, shape=(int(n),))
                allreduce(a)
                return 0
            except Exception as exc:  # noqa: BLE001
                print(f"[drake_amd] all-reduce callback failed: {exc!r}")
                return 1

        self._transport_cbs = (EXCHANGE_FN(_ex), ALLREDUCE_FN(_ar))   # keep them alive
        self._ck(self.lib.mpm_dist_set_transport(self.h, self._transport_cbs[0], self._transport_cbs[1], None,
                                                 zone_capacity_blocks))

    def dist_roles(self) -> np.ndarray:
        """Per slot: 0 the particle is not on this rank, 1 owned, 2 ghost copy."""
        out = np.zeros(self.n_particles, np.uint8)
        self._ck(self.lib.mpm_dist_roles(self.h, _ptr(out)))
        return out

    # ---- native chain: RCCL point-to-point on the engine's stream (mpm_chain_*) -----
    @staticmethod
    def chain_unique_id() -> bytes:
        try:   # (PyTorch's RCCL and its dependencies are bound once per process: torch first, see csrc/mpm_chain.h)
            import torch  # noqa: F401
        except ImportError:
            pass
        buf = C.create_string_buffer(128)
        lib = load_library()
        rc = lib.mpm_chain_unique_id(buf)
        if rc != 0:
            raise MpmError(rc, (lib.mpm_last_error() or b"").decode())
        return buf.raw

    def chain_init(self, unique_id: bytes | None, rank: int, world: int, cut_lo_block: int, cut_hi_block: int,
                   pitch_blocks: int, zone_blocks: int = 2, capacity_blocks: int = 512, periodic: bool = False):
        """unique_id = None: the geometry alone, no RCCL communicator (for the direct transport)"""
        assert unique_id is None or len(unique_id) == 128
        self._ck(self.lib.mpm_chain_init(self.h, C.c_char_p(unique_id) if unique_id is not None else None, rank, world,
                                         cut_lo_block, cut_hi_block, pitch_blocks, zone_blocks, capacity_blocks,
                                         1 if periodic else 0))

    def chain_direct_prepare(self) -> bytes:
        """allocates this rank's receive buffers of the direct halo exchange; returns their 64-byte IPC handle"""
        buf = C.create_string_buffer(64)
        self._ck(self.lib.mpm_chain_direct_prepare(self.h, buf))
        return buf.raw

    def chain_direct_connect(self, left: bytes | None, right: bytes | None):
        self._ck(self.lib.mpm_chain_direct_connect(self.h, C.c_char_p(left) if left else None, C.c_char_p(right) if right else None))

    def chain_direct_base(self) -> int:
        """address of this rank's direct-halo region (for neighbours that live in the same process)"""
        out = C.c_void_p()
        self._ck(self.lib.mpm_chain_direct_base(self.h, C.byref(out)))
        return int(out.value)

    def chain_direct_connect_local(self, left_base: int | None, right_base: int | None):
        self._ck(self.lib.mpm_chain_direct_connect_local(self.h, C.c_void_p(left_base) if left_base else None,
                                                         C.c_void_p(right_base) if right_base else None))

    def team_prepare(self, zone_capacity_blocks: int = 512):
        """TEAM transport of the distributed contact solve: allocates this rank's region; -> (64-byte IPC handle, address)"""
        buf = C.create_string_buffer(64)
        base = C.c_void_p()
        self._ck(self.lib.mpm_team_prepare(self.h, zone_capacity_blocks, buf, C.byref(base)))
        return buf.raw, int(base.value)

    def team_connect(self, handles=None, local_bases=None):
        """handles: the ranks' IPC handles in rank order (own entry ignored); local_bases: addresses of the regions of ranks
        that live in this process (None elsewhere)"""
        hb = b"".join((h if h else b"\0" * 64) for h in handles) if handles else None
        lb = None
        if local_bases:
            lb = (C.c_void_p * len(local_bases))(*[C.c_void_p(b) if b else None for b in local_bases])
        self._ck(self.lib.mpm_team_connect(self.h, C.c_char_p(hb) if hb else None, lb))

    def chain_enable_migration(self, every: int, capacity_particles: int):
        self._ck(self.lib.mpm_chain_enable_migration(self.h, every, capacity_particles))

    def chain_substeps(self, n: int, dt: float, mpm_bc: int = -1):
        self._ck(self.lib.mpm_chain_substeps(self.h, n, dt, mpm_bc))

    def chain_destroy(self):
        self._ck(self.lib.mpm_chain_destroy(self.h))

    def substep_mid_halo(self, dt: float, mpm_bc: int = -1):
        self._ck(self.lib.mpm_substep_mid_halo(self.h, dt, mpm_bc))

    def substep_end_halo(self, dt: float, mpm_bc: int, buffer_args, capacity_blocks: int):
        n, bufs = buffer_args
        self._ck(self.lib.mpm_substep_end_halo(self.h, dt, mpm_bc, n, bufs, capacity_blocks))

    def update_grid_from_sums(self, mpm_bc: int = -1):
        self._ck(self.lib.mpm_update_grid_from_sums(self.h, mpm_bc))

    def set_deterministic(self, on: bool = True):
        self._ck(self.lib.mpm_set_deterministic(self.h, 1 if on else 0))

    def set_fast_math(self, on: bool = True):
        """CalcFemStateAndForce's divisions / square roots: correctly rounded (default) or hardware approximation + one
        Newton step (mpm_set_fast_math)."""
        self._ck(self.lib.mpm_set_fast_math(self.h, 1 if on else 0))

    @property
    def fast_math(self) -> bool:
        v = C.c_int()
        self._ck(self.lib.mpm_get_fast_math(self.h, C.byref(v)))
        return bool(v.value)

    def set_stream(self, stream_handle: int | None):
        self._ck(self.lib.mpm_set_stream(self.h, C.c_void_p(stream_handle) if stream_handle else None))

    def debug_counters(self, reset: bool = True):
        out = (C.c_uint64 * 16)()
        self._ck(self.lib.mpm_debug_counters(self.h, out, 1 if reset else 0))
        return [int(x) for x in out]

    def stats(self) -> dict:
        s = Stats()
        self._ck(self.lib.mpm_get_stats(self.h, C.byref(s)))
        return {k: (float(getattr(s, k)) if t is C.c_float else int(getattr(s, k))) for k, t in Stats._fields_}

    _SHAPES = {
        ARR.POSITIONS: ("np", 3, np.float32), ARR.VELOCITIES: ("np", 3, np.float32), ARR.VOLUMES: ("np", 1, np.float32),
        ARR.AFFINE: ("np", 9, np.float32), ARR.PIDS: ("np", 1, np.int32), ARR.INDEX_MAPPINGS: ("np", 1, np.int32),
        ARR.SORT_KEYS: ("np", 1, np.uint32), ARR.FORCES: ("np", 3, np.float32), ARR.TAUS: ("np", 9, np.float32),
        ARR.DEFORMATION_GRADIENTS: ("nf", 9, np.float32), ARR.DM_INVERSES: ("nf", 4, np.float32),
        ARR.INDICES: ("nf", 3, np.int32), ARR.GRID_MASSES: ("cells", 1, np.float32),
        ARR.GRID_MOMENTUM: ("cells", 3, np.float32), ARR.GRID_V_STAR: ("cells", 3, np.float32),
        ARR.GRID_TOUCHED_FLAGS: ("blocks", 1, np.uint32), ARR.GRID_TOUCHED_IDS: ("blocks", 1, np.uint32),
        ARR.CONTACT_VEL: ("nk", 3, np.float32), ARR.CONTACT_VEL0: ("nk", 3, np.float32),
        ARR.GRID_DIR: ("cells", 3, np.float32),
    }

    def download(self, which: int) -> np.ndarray:
        kind, nc, dt = self._SHAPES[which]
        n = {"np": self.n_particles, "nf": self.n_faces, "cells": self.n_cells, "blocks": self.n_blocks,
             "nk": getattr(self, "_n_contacts", 0)}[kind]
        out = np.zeros((n, nc) if nc > 1 else (n,), dt)
        written = C.c_size_t(out.nbytes)
        self._ck(self.lib.mpm_download_array(self.h, which, _ptr(out), out.nbytes, C.byref(written)))
        if which == ARR.GRID_TOUCHED_IDS:
            out = out[: written.value // 4]
        return out

    def upload_particle_state(self, pos=None, vel=None, affine=None, volumes=None, deformation_gradients=None):
        a = [_f32(pos), _f32(vel), _f32(affine), _f32(volumes), _f32(deformation_gradients)]
        self._ck(self.lib.mpm_upload_particle_state(self.h, *[_ptr(x) for x in a]))
