/*
 * mpm_hip.h -- C ABI of the MI355X-native cloth-MPM substep engine.
 *
 * This is the drop-in boundary for the hot path of g1n0st/drake's
 * multibody/gpu_mpm layer: every entry point below replaces one method of the
 * reference's GpuMpmState<float> / GpuMpmSolver<float> pair (the only scalar
 * type the reference instantiates, cuda_mpm_model.cu:347, cuda_mpm_solver.cu:623).
 * Citations are file:line under the reference tree.
 *
 * Conventions
 *   - plain pointers and sizes only; host buffers are caller-owned
 *   - every call returns 0 on success or a negative mpm_status; the message
 *     for the calling thread's last failure is mpm_last_error().  No call throws and no call
 *     terminates the process: every entry point ends in a catch-all that turns a C++ exception
 *     raised below it into MPM_ERR_NOMEM / MPM_ERR_INTERNAL (the reference's contract --
 *     settings.h:11-25: errors never kill the caller outside DEBUG builds)
 *   - one handle <-> one HIP device; calls on one handle must be serialised
 *     by the caller (the reference is single-threaded per state as well)
 *   - "slot order" is the reference's current particle order: [faces | verts]
 *     after mpm_finalize (cuda_mpm_model.cu:40-45), permuted by every
 *     mpm_rebuild_mapping(h, 1) exactly like RebuildMapping(state, true)
 *     (stable sort on the low min(3*domain_bits,16) key bits,
 *     cuda_mpm_solver.cu:47-68).  The engine's internal memory order is
 *     different (block/cell sorted 16-byte records) and never visible through this API.
 *   - vectors cross the boundary as packed float triples / row-major 3x3,
 *     exactly like the reference's Vec3<float>/Mat3<float> device buffers.
 */
#ifndef MPM_HIP_H_
#define MPM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MPM_API __attribute__((visibility("default")))
#else
#define MPM_API
#endif

typedef struct mpm_engine *mpm_handle_t;

typedef enum {
    MPM_OK = 0,
    MPM_ERR_INVALID = -1,     /* bad argument / call order                     */
    MPM_ERR_HIP = -2,         /* a HIP runtime call failed                     */
    MPM_ERR_DRIFT = -3,       /* a face particle was re-centred out of its block's tile (diverging state) */
    MPM_ERR_CAPACITY = -4,    /* internal table overflow                       */
    MPM_ERR_NO_DEVICE = -5,   /* no usable GPU: there is no CPU fallback       */
    MPM_ERR_DOMAIN = -6,      /* a particle left the grid (the reference: undefined behaviour) */
    MPM_ERR_RANGE = -7,       /* a ParticleToGrid node sum was not finite / out of the accumulators' range */
    MPM_ERR_HALO = -8,        /* partitioned domain: a particle left the zone shared with the neighbour rank */
    MPM_ERR_NOMEM = -9,       /* host or device memory exhausted (std::bad_alloc / hipErrorOutOfMemory)        */
    MPM_ERR_INTERNAL = -10    /* a C++ exception reached the C boundary and was caught there: a bug in the engine,
                                 reported instead of terminating the calling process                         */
} mpm_status;

/* Runtime form of the compile-time constants in settings.h:36-127. */
typedef struct {
    float youngs_modulus; /* settings.h:73  (4e5)                              */
    float poisson_ratio;  /* settings.h:77  (0.3)                              */
    float density;        /* settings.h:81  (2000)                             */
    float gamma;          /* settings.h:85  (0) shear/friction of the cloth    */
    float K;              /* settings.h:92  (1e5) normal penalty stiffness     */
    float V;              /* settings.h:99  (0.8) RPIC damping blend           */
    float c_F;            /* settings.h:103 (0)                                */
    float sdf_friction;   /* settings.h:110 (0.3)                              */
    float gravity;        /* settings.h:121 (-9.8)                             */
    float epsv;           /* settings.h:125 (1e-3) friction regularisation     */
    int32_t gravity_axis; /* settings.h:118 (2)                                */
    int32_t wall_cells;   /* settings.h:56  (3) G_BOUNDARY_CONDITION           */
} mpm_material_t;

/* Arrays that mpm_download_array can return (parity tests, visualisation,
 * checkpoints).  Particle arrays come back in slot order, packed like the
 * reference's device buffers; grid arrays are dense and indexed by the
 * reference's cell key (cuda_mpm_kernels.cuh:334-341). */
typedef enum {
    MPM_ARR_POSITIONS = 0,       /* float[3*np]   current_positions()          */
    MPM_ARR_VELOCITIES = 1,      /* float[3*np]   current_velocities()         */
    MPM_ARR_VOLUMES = 2,         /* float[np]     current_volumes()            */
    MPM_ARR_AFFINE = 3,          /* float[9*np]   current_affine_matrices()    */
    MPM_ARR_PIDS = 4,            /* int32[np]     current_pids()               */
    MPM_ARR_INDEX_MAPPINGS = 5,  /* int32[np]     index_mappings()             */
    MPM_ARR_SORT_KEYS = 6,       /* uint32[np]    current_sort_keys() evaluated on the
                                    current positions (what RebuildMapping computes)  */
    MPM_ARR_FORCES = 7,          /* float[3*np]   forces()                     */
    MPM_ARR_TAUS = 8,            /* float[9*np]   taus()                       */
    MPM_ARR_DEFORMATION_GRADIENTS = 9, /* float[9*nf] original face order      */
    MPM_ARR_DM_INVERSES = 10,    /* float[4*nf]   original face order          */
    MPM_ARR_INDICES = 11,        /* int32[3*nf]   indices() (offset by +nf)    */
    MPM_ARR_GRID_MASSES = 12,    /* float[cells]  grid_masses()                */
    MPM_ARR_GRID_MOMENTUM = 13,  /* float[3*cells] grid_momentum(): momentum after
                                    ParticleToGrid, velocity after UpdateGrid   */
    MPM_ARR_GRID_V_STAR = 14,    /* float[3*cells] grid_v_star()               */
    MPM_ARR_GRID_TOUCHED_FLAGS = 15, /* uint32[blocks] grid_touched_flags()    */
    MPM_ARR_GRID_TOUCHED_IDS = 16,   /* uint32[cnt] ascending block ids        */
    MPM_ARR_CONTACT_VEL = 17,    /* float[3*nk]   contact_vel()                */
    MPM_ARR_CONTACT_VEL0 = 18,   /* float[3*nk]   contact_vel0()               */
    MPM_ARR_GRID_DIR = 19        /* float[3*cells] grid_Dir()                  */
} mpm_array_id;

/* Timed phases reported by mpm_profile_substeps. */
enum {
    MPM_PHASE_REBUILD = 0,
    MPM_PHASE_FEM = 1,     /* per-face kernel of CalcFemStateAndForce */
    MPM_PHASE_VFORCE = 2,  /* its per-vertex force gather */
    MPM_PHASE_P2G = 3,
    MPM_PHASE_GRID = 4,
    MPM_PHASE_G2P = 5,
    MPM_PHASE_COUNT = 6
};

typedef struct {
    uint64_t substeps;        /* substeps executed so far                      */
    uint64_t rebuilds;        /* internal block/cell re-sorts performed        */
    uint32_t home_blocks;     /* 4^3 blocks that currently own particles       */
    uint32_t active_blocks;   /* blocks whose nodes are updated each substep   */
    uint32_t touched_blocks;  /* reference-exact touched count of last P2G     */
    uint32_t error_flags;     /* sticky device error bits (0 = none)           */
    uint32_t active_faces;    /* particles this engine works on: all of them, or (partitioned   */
    uint32_t active_vertices; /* domain) the ones it owns plus its ghost copies                 */
    uint32_t face_slots;      /* particle slots allocated: the scene's counts, or -- after mpm_dist_init, which   */
    uint32_t vertex_slots;    /* shrinks the rank to its share -- 1.5 x what the rank held then, plus head room     */
    uint64_t particle_bytes;  /* device bytes of all arrays indexed by particle slot                              */
    uint64_t scene_index_bytes; /* device bytes indexed by ORIGINAL particle id (id -> slot map, slot-order
                                   bookkeeping, topology tables of Finalize): whole-scene sized on every rank;
                                   a partitioned engine keeps only the maps, 13 bytes per particle of the scene */
    uint64_t resort_checks;   /* times the kernels of the conditional re-sort were launched, Finalize's first sort
                                 included (mpm_run_substeps and complete phase-by-phase substeps leave them out while
                                 the quiet time below lasts, then send them with every 4th substep)                   */
    float quiet_time_s;       /* the last re-sort's estimate of how long no particle can leave its block's tile when
                                 all of them move ballistically (velocity + gravity); 0 = not estimated                */
    float since_resort_s;     /* simulated time since that re-sort                                                     */
} mpm_stats_t;

MPM_API const char *mpm_last_error(void);
MPM_API int mpm_default_material(mpm_material_t *out);

/* ---- GpuMpmState<float> -------------------------------------------------- */

/* `GpuMpmState() = default` + the grid constants of settings.h:48-59 made
 * runtime: the grid is (2^domain_bits)^3 cells of size 2^-domain_bits.
 * `device` is the HIP device ordinal.  `material` may be NULL (defaults). */
MPM_API int mpm_create(int domain_bits, const mpm_material_t *material, int device, mpm_handle_t *out);

/* GpuMpmState::AddQRCloth (cuda_mpm_model.cu:16-33).  pos/vel: float[3*n_verts];
 * indices: int32[3*n_faces] local to this cloth.  May be called repeatedly. */
MPM_API int mpm_add_qr_cloth(mpm_handle_t h, const float *pos, const float *vel, size_t n_verts,
                             const int32_t *indices, size_t n_faces);

/* GpuMpmState::Finalize (cuda_mpm_model.cu:36-122): allocates every device
 * buffer, uploads the particles and runs the FEM initialisation kernel. */
MPM_API int mpm_finalize(mpm_handle_t h);

/* GpuMpmState::Destroy (cuda_mpm_model.cu:124-242); also frees the handle. */
MPM_API int mpm_destroy(mpm_handle_t h);

/* n_verts() / n_faces() / n_particles() (cuda_mpm_model.cuh:43-45). */
MPM_API int mpm_counts(mpm_handle_t h, size_t *n_verts, size_t *n_faces, size_t *n_particles);

/* grid_touched_cnt_host() (cuda_mpm_model.cuh:95-100); synchronises. */
MPM_API int mpm_grid_touched_cnt(mpm_handle_t h, uint32_t *out);

/* GpuMpmState::DumpCpuState (cuda_mpm_model.cu:244-265): vertex positions in
 * original vertex order (float[3*n_verts]) and triangle indices local to the
 * vertex array (int32[3*n_faces]).  Either pointer may be NULL. */
MPM_API int mpm_dump_cpu_state(mpm_handle_t h, float *pos_out, int32_t *indices_out);

/* ReallocateContacts is implicit in mpm_copy_contact_pairs. */

/* GpuMpmState::ReallocateExternelBodies (cuda_mpm_model.cu:319-338): sizes and
 * zeroes the per-body impulse accumulators. */
MPM_API int mpm_reallocate_external_bodies(mpm_handle_t h, size_t n_bodies);

/* GpuMpmState::ExternelBodyForceToHost (cuda_mpm_model.cu:340-345):
 * tau_out / f_out: float[3*n_bodies] accumulated impulses (F_Bq_W_tau, F_Bq_W_f). */
MPM_API int mpm_external_body_force_to_host(mpm_handle_t h, float *tau_out, float *f_out);

/* ---- rigid feedback wire format (consumers of the accumulators above) ----- */

/* DeformableDriver::FinalizeExternalContactForces (multibody/plant/deformable_driver.h:210-219):
 * ExternelBodyForceToHost, then impulse / dt -> force for every body.  `dt` is the PLANT step
 * (the accumulators collect the impulses of all its substeps).  tau_out / f_out: float[3*n_bodies],
 * torque about the body origin p_WB handed in with the contact pairs, and force, in the world frame. */
MPM_API int mpm_finalize_external_contact_forces(mpm_handle_t h, float dt, float *tau_out, float *f_out);

/* SpatialForce::Shift (multibody/math/spatial_force.h:91-93, 151-155) for n forces:
 * tau_out[i] = tau[i] - offset[i] x f[i]; f is unchanged.  Host arithmetic, needs no GPU. */
MPM_API int mpm_spatial_force_shift(size_t n, const float *tau, const float *f, const float *offset, float *tau_out);

/* What MultibodyPlant::AddAppliedExternalSpatialForces does with external_forces_host()
 * (multibody/plant/multibody_plant.cc:2385-2407): p_BoBq_W = R_WB * p_BoBq_B,
 * F_BBo_W = SpatialForce(tau, f).Shift(-p_BoBq_W), i.e. tau_Bo = tau + p_BoBq_W x f.
 * R_WB: float[9*n] row major; the rest float[3*n].  Host arithmetic, needs no GPU. */
MPM_API int mpm_external_forces_at_body_origin(size_t n_bodies, const float *R_WB, const float *p_BoBq_B,
                                               const float *tau, const float *f, float *tau_Bo_out);

/* ---- GpuMpmSolver<float> ------------------------------------------------- */

/* GpuMpmSolver::RebuildMapping (cuda_mpm_solver.cu:17-70).  Must be called
 * once per substep before mpm_particle_to_grid, as every reference caller does
 * (deformable_driver.h:244, cuda_mpm_test.cc:66).  sort != 0 re-orders the
 * slot order like the reference's stable 16-bit radix sort. */
MPM_API int mpm_rebuild_mapping(mpm_handle_t h, int sort);

/* GpuMpmSolver::CalcFemStateAndForce (cuda_mpm_solver.cu:72-84). */
MPM_API int mpm_calc_fem_state_and_force(mpm_handle_t h, float dt);

/* GpuMpmSolver::ParticleToGrid (cuda_mpm_solver.cu:86-105). */
MPM_API int mpm_particle_to_grid(mpm_handle_t h, float dt);

/* GpuMpmSolver::UpdateGrid (cuda_mpm_solver.cu:107-151).  mpm_bc in {-1,0,1,2,3} are the
 * reference's scenes (update_grid_kernel<T, BC>, cuda_mpm_kernels.cuh:673-774); MPM_BC_TABLE
 * uses the colliders set with mpm_set_grid_colliders.  The same values are accepted wherever an
 * entry point takes mpm_bc. */
#define MPM_BC_TABLE 4
MPM_API int mpm_update_grid(mpm_handle_t h, int mpm_bc);

/* Runtime form of the analytic colliders that the reference hard-codes per scene in
 * update_grid_kernel (cuda_mpm_kernels.cuh:660-789).  For a node at x = (idx + 0.5) dx (:661-665)
 * the FIRST collider of the list with phi(x) < 0 decides (the reference's else/break chains):
 *   MPM_GC_FIXED             v <- v + (v_c - v)                                   (:778-781)
 *   MPM_GC_SLIP_APPROACHING  only if n.(v_c - v) > 0 (scene 0, :684-690), then
 *   MPM_GC_SLIP              v += mu (v_c - v) + (1 - mu) n (n.(v_c - v))          (:783-786)
 * with n = grad(phi) and mu = friction (SDF_FRICTION, settings.h:110; a negative value selects the
 * engine's material.sdf_friction).  Shapes: sphere (centre p, radius), half-space (inside where
 * n.(x - p) < 0, n a unit vector).  At most 16 colliders. */
enum { MPM_GC_SPHERE = 0, MPM_GC_HALF_SPACE = 1 };
enum { MPM_GC_FIXED = 0, MPM_GC_SLIP_APPROACHING = 1, MPM_GC_SLIP = 2 };
typedef struct mpm_grid_collider {
    int32_t shape;
    int32_t mode;
    float p[3];
    float n[3];
    float radius;
    float v[3];        /* collider velocity (0 in every reference scene) */
    float friction;
} mpm_grid_collider_t;
MPM_API int mpm_set_grid_colliders(mpm_handle_t h, size_t n, const mpm_grid_collider_t *colliders);
/* The table that reproduces scene mpm_bc in {-1,0,1,2,3} (what mpm_update_grid(h, mpm_bc) runs). */
MPM_API int mpm_grid_collider_preset(int mpm_bc, float sdf_friction, mpm_grid_collider_t *out, size_t capacity,
                                     size_t *n_out);

/* GpuMpmSolver::GridToParticle (cuda_mpm_solver.cu:153-161). */
MPM_API int mpm_grid_to_particle(mpm_handle_t h, float dt);

/* GpuMpmSolver::GpuSync (cuda_mpm_solver.cu:163-166); also surfaces sticky
 * device error flags as MPM_ERR_DRIFT / MPM_ERR_CAPACITY / MPM_ERR_DOMAIN. */
MPM_API int mpm_sync(mpm_handle_t h);

/* GpuMpmSolver::GpuSync() as the reference declares and calls it -- no state argument, a
 * cudaDeviceSynchronize() (cuda_mpm_solver.cu:164-166, cuda_mpm_test.cc:73): every engine the calling
 * thread's current HIP device holds is brought up to date (substeps that mpm_run_substeps deferred are run),
 * then the whole device is synchronised.  Like the reference's call it reports nothing about the state of a
 * simulation: the sticky errors of an engine (MPM_ERR_DRIFT, _CAPACITY, ...) stay with that engine's own
 * mpm_sync / mpm_get_stats, only a failure of the HIP runtime is an error here.  The engine list is walked under
 * the lock mpm_create / mpm_destroy take; the handles' own rule applies besides: no other thread may be inside
 * another call on one of them. */
MPM_API int mpm_device_synchronize(void);

/* Active blocks whose x block coordinate lies in [bx_lo, bx_hi]: what mpm_halo_pack would pack for that zone right
 * now (the set changes with re-sorts only).  For sizing exchange buffers after mpm_dist_init / mpm_finalize: the
 * native chain sends a fixed capacity per substep (a RCCL send needs its size when it is enqueued), so that capacity
 * should be a small multiple of this count, agreed between the neighbours, not a guess. */
MPM_API int mpm_halo_zone_blocks(mpm_handle_t h, int bx_lo, int bx_hi, uint32_t *count_out);

/* Tests: the frame of a contact with unit normal u (rows of J: two tangents, then u) exactly as the contact
 * kernels build it -- the same inline function, compiled for the host (math_tools.cuh:599-638 with axis_index 2,
 * a clone of RotationMatrix::MakeFromOneUnitVector; pinned on the reference's test vectors of
 * math/test/rotation_matrix_test.cc:1215-1252 in tests/test_contact_frame.py).  No device is touched. */
MPM_API int mpm_contact_frame(const float u[3], float J[9]);

/* bench.py, the contact leg's roofline: the four kernels of one backtracking Newton iteration of the solve that
 * mpm_update_contact has just finished (contact gradients / Hessians per cell, direction per node, line-search
 * energies, decision), each launched `reps` times back to back and timed with HIP events on the engine's stream:
 * kernel_ms[4] = duration per launch of (k_ct_tile, k_ct_node_dir, k_ct_ls, k_ct_decide).  Grid velocities are
 * left alone; call it before the next mpm_update_contact. */
MPM_API int mpm_profile_contact_iteration(mpm_handle_t h, int reps, float kernel_ms[4]);

/* Blocking copies between caller-owned host memory and device memory, ordered on the engine's stream (a plain
 * hipMemcpy on the null stream is not ordered with it).  For transports that stage the engine's exchange buffers
 * through the host (mpm_dist_set_transport): they must not bring a second HIP runtime into the process. */
MPM_API int mpm_memcpy_d2h(mpm_handle_t h, void *dst_host, const void *src_device, size_t bytes);
MPM_API int mpm_memcpy_h2d(mpm_handle_t h, void *dst_device, const void *src_host, size_t bytes);

/* Tests: substeps that mpm_run_substeps has enqueued but that found a re-sort pending and are still owed
 * (run by the next call that synchronises).  Waits for the stream; does NOT run them. */
MPM_API int mpm_debug_owed_substeps(mpm_handle_t h, uint32_t *out);

/* Bitwise reproducibility from run to run (off by default; the environment variable
 * MPM_DETERMINISTIC=1 turns it on at mpm_create).  A substep never uses float atomics, but the
 * engine's re-sort orders the particles inside a cell by the arrival of integer atomics; with
 * this switch every re-sort additionally sorts each cell's particles by their previous slot
 * (one more kernel per substep, ~2 us when idle, ~15 us per re-sort at 1M particles), and
 * ParticleToGrid accumulates its node sums in 64-bit fixed point (integer sums: exact whatever the
 * order in which the waves arrive) instead of in double precision (~1 us per substep at 1M
 * particles; the double sums are exact too unless the contributions to one node span more than
 * 2^29 in magnitude, in which case their last bit, 2^-53 of the sum, depends on the order). */
MPM_API int mpm_set_deterministic(mpm_handle_t h, int on);

/* The arithmetic of CalcFemStateAndForce's 19 divisions and 10 square roots per face (the QR / polar / SVD steps of
 * cuda_mpm_kernels.cuh:72-181 through math_tools.cuh:456-597).
 *   on = 0 (default): correctly rounded, as the reference's expressions read.  One substep then agrees with a plain-C
 *     restatement of the reference at north_star's "float state within 1e-5 relative" wherever float arithmetic can
 *     deliver that (tests/test_fast_math_gpu.py; tests/helpers.py: NOISE_FLOOR, floor_decides).
 *   on = 1: the hardware's 1-ulp reciprocal / reciprocal square root followed by one Newton step (results within
 *     ~0.6 ulp): 2.4 us per substep less at 1M particles (k_fem 910 -> 630 vector instructions per face); the
 *     velocities of one substep then differ from the strict path's by about the float rounding noise of that substep
 *     (1.4 x the distance between a float and a double evaluation; 1.0 - 1.7e-5 of max|v| on the parity scenes).
 *     The reference itself is built with -use_fast_math (tools/skylark/cuda.bzl:66-80), i.e. with CUDA's
 *     approximate division and square root: this switch is the closer analogue of what the reference RUNS, the default
 *     of what its source SAYS.
 * Also MPM_FAST_MATH=1 in the environment at mpm_create.  May be changed between substeps. */
MPM_API int mpm_set_fast_math(mpm_handle_t h, int on);
MPM_API int mpm_get_fast_math(mpm_handle_t h, int *on_out);

/* GpuMpmSolver::SyncParticleStateToCpu (cuda_mpm_solver.cu:185-191):
 * pos_out: float[3*n_particles] in slot order (the reference fills
 * positions_host()). */
MPM_API int mpm_sync_particle_state_to_cpu(mpm_handle_t h, float *pos_out);

/* GpuMpmSolver::Dump (cuda_mpm_solver.cu:168-183): Wavefront OBJ. */
MPM_API int mpm_dump_obj(mpm_handle_t h, const char *filename);

/* GpuMpmSolver::CopyContactPairs (cuda_mpm_solver.cu:193-212) taking the
 * MpmParticleContactPairs SoA (cpu_mpm_model.h:73-114): particle slot index,
 * rigid body index, signed distance (<0), normal, contact position, rigid
 * point velocity, body origin p_WB. */
MPM_API int mpm_copy_contact_pairs(mpm_handle_t h, size_t n_contacts, const uint32_t *particle_in_contact_index,
                                   const uint32_t *non_mpm_id, const float *penetration_distance,
                                   const float *normal, const float *particle_in_contact_position,
                                   const float *rigid_v, const float *rigid_p_WB);

/* Device-side replacement of DeformableDriver::CalcMpmContactPairs (deformable_driver.h:120-194)
 * followed by CopyContactPairs, for rigid bodies with analytic signed distance fields: no particle
 * positions travel to the host.  For every particle slot s (ascending) and collider j (ascending)
 * with phi_j(x_s) < 0 one contact is produced with exactly the fields of MpmParticleContactPairs
 * (cpu_mpm_model.h:73-114; deformable_driver.h:181-187): particle_in_contact_index = s,
 * non_mpm_id = body, penetration_distance = phi, normal = -grad(phi)/|grad(phi)|,
 * position = x_s, rigid_v = v + w x (x_s - p_WB), rigid_p_WB = p_WB.
 *   kind 0 half-space: inside is z_B <= 0 (Drake's HalfSpace), plane through p_WB, normal = R_WB[:,2]
 *   kind 1 sphere:     radius dims[0], centre p_WB
 *   kind 2 box:        half extents dims[0..2] in the body frame
 *   kind 3 capsule:    radius dims[0], half length dims[1] along z_B
 * The pairs are left in the engine as if mpm_copy_contact_pairs had been called;
 * mpm_download_contact_pairs returns them (any pointer may be NULL). */
typedef struct mpm_collider {
    int32_t kind;
    uint32_t body;      /* index into the external-body accumulators */
    float p_WB[3];      /* body origin in the world */
    float R_WB[9];      /* rotation body -> world, row major */
    float dims[3];
    float v[3], w[3];   /* spatial velocity of the body frame, world */
} mpm_collider_t;
MPM_API int mpm_generate_contact_pairs(mpm_handle_t h, size_t n_colliders, const mpm_collider_t *colliders,
                                       size_t *n_contacts_out);
/* n_contacts_out may be NULL: the call then waits for nothing -- the pairs are counted on the device and STAY counted
 * there; mpm_update_contact's launches have fixed grids and read the count on the device (the reference's driver reads
 * every position back, loops over the particles on the host and uploads the pairs, per substep:
 * deformable_driver.h:120-194, cuda_mpm_solver.cu:185-212).  mpm_update_contact reports the count afterwards
 * (mpm_contact_stats_t::contacts); mpm_get_contact_pair_count reads it back on request (a synchronisation point), as do
 * mpm_download_contact_pairs and a non-NULL n_contacts_out.  Up to 16 colliders travel as a kernel argument (no upload);
 * more through a device array that is refreshed only when the colliders have changed. */
MPM_API int mpm_get_contact_pair_count(mpm_handle_t h, size_t *n_contacts_out);
/* What the last mpm_update_contact worked on, without touching the device: contacts, grid nodes reached by a contact
 * stencil, and whether its set-up was the previous solve's, reused (a settled scene whose pair list repeats: no sort, no
 * per-cell runs, no node list -- verified on the device entry by entry).  Any pointer may be NULL. */
MPM_API int mpm_last_contact_counts(mpm_handle_t h, uint32_t *contacts_out, uint32_t *nodes_out, int *setup_reused_out);
/* Tests: {mpm_update_contact calls that solved, of them on a reused set-up, solves refused on the device because a guess of
 * the host's was wrong (repeated with the full set-up), pair generations / solves repeated after a buffer overflow,
 * substeps of mpm_run_coupled_substeps that were enqueued contact-free (no particle in any collider when the substep before
 * ended), of them skipped on the device and repeated as coupled substeps}. */
MPM_API int mpm_debug_contact_counters(mpm_handle_t h, uint64_t out6[6]);
/* Tests of the invariant that no kernel of the solve indexes a per-pair array with a count it has not clamped to the
 * capacity that sized the array (the reference sizes its launches from a host count, cuda_mpm_solver.cu:222-232; here the
 * count of device-made pairs never leaves the device).  Overwrites the count mpm_generate_contact_pairs left on the device
 * (count >= 0) and / or shifts the number of the pair generation that wrote it (stamp_delta != 0: "a count some other
 * generation left", i.e. a stale one).  The next mpm_update_contact must refuse: MPM_ERR_CAPACITY for a count outside
 * [0, capacity], MPM_ERR_INTERNAL for a stale one -- nothing indexed, nothing solved; pairs made again afterwards solve as
 * if nothing had happened.  Needs pairs counted on the device (n_contacts_out == NULL). */
MPM_API int mpm_debug_contact_count(mpm_handle_t h, int count, int stamp_delta);
MPM_API int mpm_download_contact_pairs(mpm_handle_t h, uint32_t *particle_in_contact_index, uint32_t *non_mpm_id,
                                       float *penetration_distance, float *normal, float *position, float *rigid_v,
                                       float *rigid_p_WB);

/* GpuMpmSolver::UpdateContact (cuda_mpm_solver.cu:214-621).  iterations_out,
 * residual_out may be NULL.  frame/substep/dump only name the optional JSON
 * statistics file (written to dump_dir set by mpm_set_dump_dir, default ".").
 * max_newton_iterations <= 0 selects the reference's 2000. */
MPM_API int mpm_update_contact(mpm_handle_t h, int frame, int substep, float dt, float friction_mu, float stiffness,
                               float damping, int dump, int exact_line_search, int max_newton_iterations,
                               int *iterations_out, float *residual_out);
MPM_API int mpm_set_dump_dir(mpm_handle_t h, const char *dir);

/* The numbers the reference prints and dumps per UpdateContact call (cuda_mpm_solver.cu:577-612:
 * iteration count, residual, line-search count, energy), plus the step and energies of the LAST
 * Newton iteration so that a single iteration (max_newton_iterations = 1) can be compared:
 * alpha = accepted step, energy = E(alpha) as the line search evaluated it, E0 = E(0),
 * norm_dir_sq / dofs = sum |Dir|^2 before relaxation and the DoF count (cuda_mpm_solver.cu:567-570). */
typedef struct {
    int32_t iterations;
    int32_t line_search_evals;   /* summed over the iterations */
    uint32_t contacts;
    uint32_t nodes;              /* grid nodes reached by a contact stencil */
    float residual;
    float alpha;
    float energy;
    float E0;
    float norm_dir_sq;
    float dofs;
} mpm_contact_stats_t;
MPM_API int mpm_get_contact_stats(mpm_handle_t h, mpm_contact_stats_t *out);

/* The decisions of the last mpm_update_contact, one row per Newton iteration -- what the reference's host loop decides per
 * iteration (cuda_mpm_solver.cu:472-528 backtracking, :383-471 exact search, :567-570 stopping test) and dumps as JSON
 * for the exact search (:587-612).  Row = MPM_CONTACT_LOG_FLOATS floats:
 *   [0] residual sqrt(sum |Dir|^2) / DoFs   [1] line-search evaluations   [2] E(alpha)   [3] accepted alpha   [4] E(0)
 *   [5] sum |Dir|^2 (before relaxation)     [6] DoFs                      [7] 0
 * At most 2048 iterations are kept.  rows_out may be NULL (count only). */
#define MPM_CONTACT_LOG_FLOATS 8
MPM_API int mpm_download_contact_log(mpm_handle_t h, float *rows_out, size_t capacity_rows, size_t *n_rows_out);

/* ---- conveniences on top of the reference interface ---------------------- */

/* The five solver calls of one contact-free substep (cuda_mpm_test.cc:66-72,
 * deformable_driver.h:244-258 without the contact part), without host syncs. */
MPM_API int mpm_substep(mpm_handle_t h, float dt, int mpm_bc);
MPM_API int mpm_run_substeps(mpm_handle_t h, int n, float dt, int mpm_bc);

/* The same for COUPLED substeps: the body of DeformableDriver::CalcAbstractStates' loop
 * (multibody/plant/deformable_driver.h:240-258) n times in one call, for rigid bodies with analytic signed distance
 * fields (mpm_collider_t, see mpm_generate_contact_pairs): RebuildMapping, CalcFemStateAndForce, ParticleToGrid,
 * UpdateGrid(mpm_bc), contact pairs on the device, UpdateContact(dt, mu, stiffness, damping, exact), GridToParticle.
 * The results are those of the seven calls (the tests compare them bit for bit); the host waits once per substep, for
 * the word that says the solve has converged.  mpm_reallocate_external_bodies must have been called (the per-body
 * impulses accumulate over the n substeps, as over the substeps of one plant step: deformable_driver.h:196-219).
 * While nothing touches a collider the call goes without pair generation and solve: after a substep without pairs a watch
 * kernel asks whether the NEXT substep would have any (a pair is a particle with phi < 0 where the substep starts: exact),
 * and the substeps that follow are enqueued contact-free, each checked the same way; the ones that turn out to have pairs
 * skip themselves on the device and are run again as coupled substeps.  Their results report 0 contacts and iterations.
 * results: n entries or NULL. */
typedef struct {
    float dt;
    int32_t mpm_bc;
    float friction_mu, stiffness, damping;
    int32_t exact_line_search;
    int32_t max_newton_iterations;   /* <= 0: the reference's 2000 */
} mpm_coupled_params_t;
typedef struct {
    int32_t iterations;
    uint32_t contacts, nodes;
    float residual;
    int32_t setup_reused;
} mpm_coupled_result_t;
MPM_API int mpm_run_coupled_substeps(mpm_handle_t h, int n, const mpm_coupled_params_t *params, size_t n_colliders,
                                     const mpm_collider_t *colliders, mpm_coupled_result_t *results);
/* (partitioned domains: see mpm_team_prepare below) */
MPM_API int mpm_world_coupled_substeps(mpm_handle_t *handles, int n_local, int n, const mpm_coupled_params_t *params,
                                       size_t n_colliders, const mpm_collider_t *colliders, mpm_coupled_result_t *const *results);

/* Runs n substeps with HIP events around every kernel group on the engine's
 * stream and returns the mean milliseconds per substep of each phase
 * (phase_ms[MPM_PHASE_COUNT]) and of the whole substep. */
MPM_API int mpm_profile_substeps(mpm_handle_t h, int n, float dt, int mpm_bc, float *phase_ms, float *total_ms);

/* ---- multi-GPU: one engine per GPU, domains tiled along x ------------------
 * Each rank runs its own engine on its own local grid; neighbouring ranks' grids overlap in a
 * few block layers next to the cut.  Per substep, between mpm_particle_to_grid and the grid
 * update:  mpm_grid_gather -> mpm_halo_pack (one buffer per neighbour) -> exchange the buffers
 * (RCCL send/recv on the same stream, done by the caller) -> mpm_halo_add for every received
 * buffer -> mpm_update_grid_from_sums.  Nothing here synchronises with the host.
 * The reference has no multi-GPU path (settings.h:40 G_DEVICE_COUNT = 1); this is new surface. */

/* Raw node sums (mass, momentum) of all active blocks, without the update. */
MPM_API int mpm_grid_gather(mpm_handle_t h);
/* Size in bytes of a halo buffer for `capacity_blocks` blocks. */
MPM_API size_t mpm_halo_buffer_bytes(size_t capacity_blocks);
/* Packs the sums of the active blocks whose x block coordinate is in [bx_lo, bx_hi] into
 * dev_buf (device memory), relabelled by shift_bx blocks (the neighbour's local coordinates). */
MPM_API int mpm_halo_pack(mpm_handle_t h, int bx_lo, int bx_hi, int shift_bx, void *dev_buf, size_t capacity_blocks);
/* Adds a neighbour's packed sums to the local ones (blocks that are not active here are skipped). */
MPM_API int mpm_halo_add(mpm_handle_t h, const void *dev_buf, size_t capacity_blocks);
/* UpdateGrid (cuda_mpm_solver.cu:107-151) on sums that already include the neighbours'. */
MPM_API int mpm_update_grid_from_sums(mpm_handle_t h, int mpm_bc);
/* The two halves of a multi-GPU substep: RebuildMapping + CalcFemStateAndForce + ParticleToGrid +
 * mpm_grid_gather, and mpm_update_grid_from_sums + GridToParticle. */
MPM_API int mpm_substep_begin(mpm_handle_t h, float dt);
MPM_API int mpm_substep_end(mpm_handle_t h, float dt, int mpm_bc);
/* The same with the halo kernels folded in (one host call either side of the exchange):
 * begin + mpm_halo_pack for each of the n_zones <= 2 (bx_lo[i], bx_hi[i], shift_bx[i]) -> send_bufs[i];
 * mpm_halo_add for each of the n_bufs received buffers + end. */
MPM_API int mpm_substep_begin_halo(mpm_handle_t h, float dt, int n_zones, const int *bx_lo, const int *bx_hi,
                                   const int *shift_bx, void *const *send_bufs, size_t capacity_blocks);
/* The same chain driven by the library itself: RCCL point-to-point on the engine's stream (no
 * event between pack, transfer and add; a cross-stream dependency costs ~20 us on this stack) and
 * one host call for a batch of substeps.  Rank 0 makes the id (mpm_chain_unique_id), the caller
 * distributes its 128 bytes by any means, every rank calls mpm_chain_init with its own handle.
 * Rank r's local frame is shifted by r * pitch_blocks blocks along x; cut_lo / cut_hi are the local
 * x block indices of the left / right cut planes, zone_blocks the halo depth either side of a cut.
 * periodic != 0 closes the chain into a ring (rank world-1's right neighbour is rank 0): used to
 * exercise the transport with a single rank, whose neighbours are then itself.  id == NULL: the geometry only, without
 * an RCCL communicator (for the direct transport below). */
MPM_API int mpm_chain_unique_id(char id_out[128]);
MPM_API int mpm_chain_init(mpm_handle_t h, const char id[128], int rank, int world, int cut_lo_block, int cut_hi_block,
                           int pitch_blocks, int zone_blocks, size_t capacity_blocks, int periodic);
MPM_API int mpm_chain_substeps(mpm_handle_t h, int n_substeps, float dt, int mpm_bc);
/* Partitioned domain (mpm_dist_init + mpm_chain_init with pitch_blocks = 0): mpm_chain_substeps also
 * swaps migration records with the neighbours every `every` substeps (mpm_dist_migrate_pack / _apply);
 * every = 0: when the ranks' common quiet-time estimate says so (mpm_dist_migration_quiet_time). */
MPM_API int mpm_chain_enable_migration(mpm_handle_t h, int every, size_t capacity_particles);
MPM_API int mpm_chain_destroy(mpm_handle_t h);
/* DIRECT halo exchange (round 5; the reference has no multi-GPU path, settings.h:40): instead of RCCL send / recv -- a
 * kernel of its own plus a staging copy, ~21 us of every chain substep on this stack -- the kernel that gathers a rank's
 * zone sums stores them straight into the NEIGHBOUR's receive buffer (device memory of the peer, mapped through a HIP IPC
 * handle; xGMI on one node), a one-thread kernel raises a sequence flag over there, and a one-wave kernel on the
 * receiving side waits for it (bounded: MPM_HALO_TIMEOUT_S, default 5 s, then MPM_ERR_HALO) before the grid update reads
 * the sums.  Receive buffers alternate between two parities, so nobody overwrites what a neighbour may still read.
 *   mpm_chain_init(h, NULL, rank, world, ...)     the geometry alone (no RCCL communicator; or with an id: RCCL stays
 *                                                 available for the rare exchanges -- migration, the contact solve)
 *   mpm_chain_direct_prepare(h, handle_out)       allocates this rank's receive buffers, returns their IPC handle
 *   (the caller hands every rank its neighbours' 64 bytes, by any means)
 *   mpm_chain_direct_connect(h, left, right)      maps the neighbours' buffers (NULL where there is no neighbour; a rank
 *                                                 that is its own neighbour -- a ring of one -- needs no handle);
 *                                                 both NULL on a rank that HAS neighbours: the direct path is switched
 *                                                 off again (what a caller does on every rank when one of them could
 *                                                 not map its neighbour: all ranks must use the same transport)
 * mpm_chain_substeps then uses the direct path for the per-substep halo.  Validated on one GPU only (two processes
 * sharing it; the ring of one): the protocol, the indexing and the time-out -- NOT the ordering of peer stores across
 * two devices, which this build has never run on. */
MPM_API int mpm_chain_direct_prepare(mpm_handle_t h, char handle_out[64]);
MPM_API int mpm_chain_direct_connect(mpm_handle_t h, const char left_handle[64], const char right_handle[64]);
/* Round 6: the region must be FINE-GRAINED device memory -- mpm_chain_direct_prepare returns MPM_ERR_HIP when the runtime
 * refuses it (a coarse-grained region would let stale lines of the receiver's L2 shadow a peer's stores: wrong sums, no
 * error flag), and the caller then keeps every rank on RCCL; MPM_DIRECT_COARSE_OK=1 accepts plain device memory for
 * rehearsals on ONE device.  The pack counts its entries in this device's memory and the count travels with the signal: no
 * returning atomic on peer memory.
 * Neighbours that live in THIS process (an in-process world: several ranks of one partition on one stream) are named by
 * pointer -- a HIP IPC handle of one's own process cannot be opened: mpm_chain_direct_base hands a rank's region out,
 * mpm_chain_direct_connect_local takes the neighbours' (NULL where there is none). */
MPM_API int mpm_chain_direct_base(mpm_handle_t h, void **base_out);
MPM_API int mpm_chain_direct_connect_local(mpm_handle_t h, void *left_base, void *right_base);

/* TEAM transport of the distributed contact solve (round 6; the reference has one device, settings.h:40, and reads its
 * global scalars back per Newton iteration, cuda_mpm_solver.cu:318-319, 359-363, 515-517, 567-570): on a partitioned
 * domain the per-node Hessian / gradient sums of the zone blocks go to the two neighbours and the line-search sums to
 * every rank as stores into each other's memory + sequence flags ON THE ENGINE'S STREAM; every rank adds all ranks' sums
 * in rank order (identical bits, identical `E1 <= E0` decisions, no collective library), whether any rank has a pair at
 * all and whether any rank must refuse its solve (buffers overflowed, a wrong guess of its host) is agreed the same way,
 * and the host only polls the mailbox, as on one GPU.  Needs mpm_dist_init; at most 8 ranks (one node).
 *   mpm_team_prepare(h, zone_capacity_blocks, handle_out, base_out)   allocates this rank's region (fine-grained, as
 *        above), returns its IPC handle (64 bytes, may be NULL) and its address (for ranks of the same process, may be NULL)
 *   mpm_team_connect(h, handles, local_bases)   handles: world x 64 bytes in rank order (own entry ignored) or NULL;
 *        local_bases: world pointers or NULL -- a rank of this process is named by its region's address, any other by
 *        its handle.
 * mpm_chain_destroy -- and mpm_chain_init, which starts from a clean chain -- release the team's region too: set the chain up
 * first (mpm_chain_init, mpm_chain_direct_*), then the team.
 * With the team connected, mpm_update_contact on a partitioned engine runs the device-resident solve (EVERY rank must
 * call it in every coupled substep, with or without pairs of its own), and mpm_run_coupled_substeps accepts a partitioned
 * engine whose halo runs over the direct transport: n coupled substeps per call, no migration inside (the caller runs the
 * substeps between two migrations in one call), no contact-free speculation.  mpm_world_coupled_substeps does the same for
 * the n_local ranks of an in-process world (all on one stream), enqueued phase by phase across the ranks; results: n_local
 * arrays of n entries, or NULL.  Validated between processes sharing one GPU and in in-process worlds: the protocol, the
 * indexing, the rank-order sums -- NOT the ordering of peer stores across two devices. */
MPM_API int mpm_team_prepare(mpm_handle_t h, size_t zone_capacity_blocks, char handle_out[64], void **base_out);
MPM_API int mpm_team_connect(mpm_handle_t h, const char *handles, void *const *local_bases);

/* ---- multi-GPU, ONE domain cut into x slabs (strong scaling) ------------------------------
 * Every rank is created and finalised with the WHOLE scene (same mpm_add_qr_cloth calls: replicated
 * topology, full-capacity arrays) and then keeps only its share: rank r owns the particles whose
 * base cell lies in x blocks [own_lo_block, own_hi_block) (the outermost ranks extend to the walls),
 * plus ghost copies of the neighbours' particles within ghost_cells of a cut.  Ghosts take part in
 * CalcFemStateAndForce and GridToParticle on both ranks (same inputs, same arithmetic: the copies
 * stay bit-identical without communication) and scatter nothing in ParticleToGrid.  Per substep the
 * ranks exchange the node sums of the blocks within zone_blocks of a cut: mpm_substep_begin_halo with
 * zones [cut - zone_blocks, cut + zone_blocks - 1] and shift 0, then mpm_substep_end_halo (or
 * mpm_chain_substeps with pitch 0).  Every few substeps particles change hands:
 *   mpm_dist_migrate_pack -> exchange the two record buffers with the neighbours ->
 *   mpm_dist_migrate_apply -> (the next RebuildMapping merges / drops them).
 * Requirements, checked: 4 * zone_blocks >= ghost_cells + ghost_margin_cells + 2; a slab with two
 * neighbours is at least 2 * zone_blocks wide; a particle whose stencil leaves the shared zone, or a
 * face that misses a corner vertex (mesh edge longer than ghost_margin_cells), raises MPM_ERR_HALO.
 * Arrays downloaded from a rank hold NaN / -1 for particles it does not have; mpm_dist_roles tells
 * which are owned (1), ghosts (2) or absent (0), per slot.  The reference has no multi-GPU path.
 * Meshes: any valence.  The per-slot topology that migrates with a vertex holds eight (face, corner) ids (cloth meshes
 * have six faces around a vertex); a mesh with a vertex of more keeps the scene's adjacency (4 bytes per vertex + 12 per
 * face) on every rank, and such a vertex sums its force over it.  After the call a rank's particle arrays are sized for
 * slot_headroom x what it holds (default 1.5; MPM_DIST_HEADROOM, 0 = keep the whole scene's size); a rank whose share
 * outgrows that is re-allocated at the migration that would overflow it (the stream is idle there), up to
 * the whole scene's size. */
typedef struct {
    int32_t rank, world;
    int32_t own_lo_block, own_hi_block;     /* this rank's x blocks */
    int32_t left_lo_block, right_hi_block;  /* outer ends of the neighbours' ranges (ignored without neighbour) */
    int32_t zone_blocks;                    /* depth of the exchanged zone either side of a cut */
    int32_t ghost_cells;                    /* width of the ghost band for face particles (>= longest mesh edge) */
    int32_t ghost_margin_cells;             /* the band for vertex particles is wider by this much (>= longest
                                               mesh edge), so that a ghost face always has its corners */
} mpm_dist_config_t;
MPM_API int mpm_dist_init(mpm_handle_t h, const mpm_dist_config_t *config);
/* ghost_cells = ghost_margin_cells = 0 selects BANDS FROM THE MESH: fractional widths (in cells, measured on x / dx - 0.5)
 * chosen so that the drift every particle may accumulate along x between two migrations is as large as the zone allows:
 *   reach = 0.75 x longest mesh edge (cells);   drift = (4 zone_blocks - 2 - 2 reach) / 3;
 *   face band = reach + 2 drift;   vertex band = 2 reach + 2 drift.
 * (zone_blocks = 2 and 0.62-cell edges: drift 1.7 cells against 0.8 for the integer bands 2 + 2; zone_blocks = 1, the
 * only depth a slab two blocks wide allows: 0.36 cells, where integer bands 1 + 1 leave none.)  Given integer bands
 * allow drift = min((ghost_cells - reach) / 2, 4 zone_blocks - 2 - ghost_cells - ghost_margin_cells). */
typedef struct {
    float face_band_cells, vertex_band_cells;   /* as in use */
    float drift_budget_cells;                   /* what a particle may drift along x between two migrations */
    float longest_edge_cells;
    uint32_t slot_resizes;                      /* re-allocations of the rank's slot space (1 = mpm_dist_init's own) */
    uint32_t migrations;                        /* mpm_dist_migrate_pack calls so far */
    uint32_t retunes;                           /* times mpm_dist_retune changed the band widths */
} mpm_dist_geometry_t;
MPM_API int mpm_dist_get_geometry(mpm_handle_t h, mpm_dist_geometry_t *out);
/* Slot space of a partitioned rank = factor x what it holds after the partition (default 1.5, or MPM_DIST_HEADROOM;
 * 0 = keep the whole scene's size).  Before mpm_dist_init.  Whatever the factor, a migration that would overflow
 * the slot space re-allocates it first (mpm_dist_migrate_apply), up to the whole scene's size. */
MPM_API int mpm_dist_set_headroom(mpm_handle_t h, float factor);
/* Adaptive migration cadence.  mpm_dist_migrate_pack also estimates how long no particle this rank holds can drift
 * further along x than drift_budget_cells if all keep moving ballistically (a particle far from both cuts has to get
 * near one first); this call waits for the stream and returns that time (seconds, may be infinity).  The ranks take
 * the minimum over all of them and migrate again when a share of it has passed (mpm_chain_enable_migration with
 * every = 0 does exactly that: ncclAllReduce(min), half of the common estimate). */
MPM_API int mpm_dist_migration_quiet_time(mpm_handle_t h, float *seconds_out);
/* Bands from the mesh follow the motion: called by every rank with the ranks' COMMON estimate right after a migration
 * (mpm_chain_substeps does it itself), this re-sizes the bands so that at the speed that estimate implies a migration is
 * due every ~16 substeps (MPM_DIST_INTERVAL): wide bands for a cloth that moves along x, the narrowest (0.125 cells of
 * drift: fewest ghost particles) for one that does not; within what the zone allows, in steps of an eighth of a cell.
 * All ranks compute the same widths from the same numbers; they take effect at the next migration.  No-op for given
 * integer bands or with MPM_DIST_RETUNE=0. */
MPM_API int mpm_dist_retune(mpm_handle_t h, float quiet_time_all, float dt, int *changed_out);
MPM_API size_t mpm_dist_migration_buffer_bytes(size_t capacity_particles);
/* send_left / send_right: device buffers of mpm_dist_migration_buffer_bytes(capacity) each (both
 * required; a rank without that neighbour gets an empty one).  recv_*: what the neighbours packed, or NULL. */
MPM_API int mpm_dist_migrate_pack(mpm_handle_t h, void *send_left, void *send_right, size_t capacity_particles);
MPM_API int mpm_dist_migrate_apply(mpm_handle_t h, const void *recv_left, const void *recv_right,
                                   size_t capacity_particles);
MPM_API int mpm_dist_roles(mpm_handle_t h, uint8_t *roles_out /* n_particles */);
/* What mpm_dist_migrate_apply decides from the two 16-byte headers of the received buffers ([0] records, [1] how many of
 * them are faces, [2..3] zero; NULL = no such neighbour) before it sizes anything: the headers are CHECKED against the
 * buffers' capacity and the scene (more records than the buffers hold: MPM_ERR_CAPACITY, the sender overflowed; more
 * faces than records or non-zero padding: MPM_ERR_INVALID, a corrupt header; inconsistent own counts:
 * MPM_ERR_INTERNAL), then out6 = {arriving faces, arriving vertices, face slots needed, vertex slots needed, face
 * slots to re-allocate to, vertex slots to re-allocate to} (the last two equal the current slot space when it
 * suffices; never more than the scene has).  Host arithmetic only: works without a GPU (tests/test_error_paths.py). */
MPM_API int mpm_dist_plan_migration(const uint32_t *hdr_left, const uint32_t *hdr_right, size_t capacity_particles,
                                    size_t scene_faces, size_t scene_vertices, size_t held_faces, size_t held_vertices,
                                    size_t face_slots, size_t vertex_slots, float headroom, size_t out6[6]);
/* Re-allocation of a partitioned rank's slot space is TWO-PHASE: every new array is allocated before anything of the
 * engine is touched, so a failed allocation (MPM_ERR_NOMEM) leaves the handle exactly as it was -- usable, at its old
 * size, with the arriving records not applied (the caller may free memory and call mpm_dist_migrate_apply again with
 * the same buffers). */

/* Tests of the error contract.  mpm_debug_throw raises a C++ exception inside an entry point (kind 0: std::bad_alloc,
 * 1: std::length_error from a container asked for an absurd size, 2: a non-std exception) and returns what the
 * boundary's catch-all made of it: MPM_ERR_NOMEM / MPM_ERR_INTERNAL, message in mpm_last_error(); host code only.
 * mpm_debug_fail_alloc: the engine's nth device allocation from now fails as if the device were out of memory
 * (0 = off). */
MPM_API int mpm_debug_throw(int kind);
MPM_API int mpm_debug_fail_alloc(mpm_handle_t h, int nth);

/* UpdateContact on a partitioned domain.  Every rank solves for the grid nodes it holds with the
 * contacts of the particles it OWNS (pass only those to mpm_copy_contact_pairs;
 * mpm_generate_contact_pairs does so by itself).  Per Newton iteration the ranks (1) exchange the
 * contact Hessian / gradient sums of the nodes in the zones next to the cuts (a node there can be
 * reached by contacts of both ranks), after which both compute the same direction for it, and (2)
 * all-reduce the line-search energies, sum |Dir|^2 and the DoF count -- the scalars the reference
 * reads back per iteration (cuda_mpm_solver.cu:318-319, 359-363, 515-517, 567-570); shared nodes are
 * counted on their owner.  The step and the stopping decision then agree on all ranks.
 * Transport: the native chain when mpm_chain_init was called (RCCL on the engine's stream), else the
 * two callbacks below.  exchange: send_* / recv_* are device buffers of `bytes` bytes each, the
 * engine's stream has been synchronised before the call and the received bytes must be in place on
 * return (a missing neighbour's buffers are to be ignored).  allreduce: sum `n` doubles in place over
 * all ranks (host memory).  Per-body impulses (mpm_external_body_force_to_host) are per-rank partial
 * sums: add them over the ranks. */
typedef int (*mpm_exchange_fn)(void *user, void *send_left, void *send_right, void *recv_left, void *recv_right,
                               size_t bytes);
typedef int (*mpm_allreduce_fn)(void *user, double *values, size_t n);
MPM_API int mpm_dist_set_transport(mpm_handle_t h, mpm_exchange_fn exchange, mpm_allreduce_fn allreduce, void *user,
                                   size_t zone_capacity_blocks);

/* Optional, between the two: the part of UpdateGrid and GridToParticle that does not depend on the
 * neighbours' sums (blocks outside the zones given to begin, work items whose tiles do not touch a
 * zone), to be enqueued while the exchange is in flight; mpm_substep_end_halo then only does the rest. */
MPM_API int mpm_substep_mid_halo(mpm_handle_t h, float dt, int mpm_bc);
MPM_API int mpm_substep_end_halo(mpm_handle_t h, float dt, int mpm_bc, int n_bufs, const void *const *recv_bufs,
                                 size_t capacity_blocks);

/* Use a caller-provided hipStream_t (e.g. torch's current stream); NULL
 * restores the engine's own stream. */
MPM_API int mpm_set_stream(mpm_handle_t h, void *hip_stream);

MPM_API int mpm_get_stats(mpm_handle_t h, mpm_stats_t *out);

/* Diagnostic build support: 16 device counters that kernels fill only when the
 * MPM_DBG environment variable has bit 2 set (see DESIGN.md "Diagnostics"). */
MPM_API int mpm_debug_counters(mpm_handle_t h, uint64_t *out16, int reset);

/* Copies one engine array to the host (see mpm_array_id).  `bytes` is the
 * size of `out`; the call fails if it is too small. *written gets the number
 * of bytes produced (may be NULL). */
MPM_API int mpm_download_array(mpm_handle_t h, int which, void *out, size_t bytes, size_t *written);

/* Overwrites particle state -- checkpoint restore and test injection.  pos,
 * vel (float[3*np]), affine (float[9*np]), volumes (float[np]) are in slot
 * order; deformation_gradients (float[9*nf]) in original face order, like
 * mpm_download_array returns them.  Any pointer may be NULL (left unchanged). */
MPM_API int mpm_upload_particle_state(mpm_handle_t h, const float *pos, const float *vel, const float *affine,
                                      const float *volumes, const float *deformation_gradients);

/* The root finder inside UpdateContact's exact line search (cuda_mpm_solver.cu:383-471), a float
 * clone of Drake's DoNewtonWithBisectionFallback (multibody/contact_solvers/
 * newton_with_bisection.cc:14-119), exposed so that the reference's own root-finding cases
 * (multibody/contact_solvers/test/newton_with_bisection_test.cc:55-225) can be run through the
 * product's code.  fn(user, x, &f, &df) evaluates the function and its derivative.  flags = 0 gives
 * Drake's semantics; the solver uses MPM_RF_SIGN3 | MPM_RF_NO_ENDS | MPM_RF_STEP_LAST (the clone's
 * deviations, see drake_amd/csrc/mpm_rootfind.h).  Returns 0 on convergence, 1 when max_evals ran
 * out (Drake throws), negative on bad arguments.  Host code only: works without a GPU. */
enum { MPM_RF_SIGN3 = 1, MPM_RF_NO_ENDS = 2, MPM_RF_STEP_LAST = 4 };
typedef void (*mpm_rootfind_fn)(void *user, double x, double *f, double *df);
MPM_API int mpm_newton_bisect_f64(mpm_rootfind_fn fn, void *user, double x_lo, double x_hi, double guess,
                                  double x_tol, double f_tol, int max_evals, int flags, double *root_out,
                                  int *evals_out);
MPM_API int mpm_newton_bisect_f32(mpm_rootfind_fn fn, void *user, float x_lo, float x_hi, float guess, float x_tol,
                                  float f_tol, int max_evals, int flags, float *root_out, int *evals_out);

#ifdef __cplusplus
}
#endif
#endif /* MPM_HIP_H_ */
