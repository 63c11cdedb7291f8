// Host-side engine object shared by the C-ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mpm_hip.h"
#include "mpm_device.h"
#include "mpm_rebuild.h"
#include "mpm_step.h"
#include "mpm_contact_dev.h"
#include "mpm_team.h"
#include "mpm_trace.h"

using namespace mpm;

static thread_local std::string g_last_error;

// (never throws: the message of a failure must not become a second failure -- the catch-all of the C ABI calls this)
static int fail(int code, const std::string& msg) noexcept {
    try {
        g_last_error = msg;
    } catch (...) {
        g_last_error.clear();
    }
    return code;
}
static int fail(int code, const char* msg) noexcept {
    try {
        g_last_error = msg;
    } catch (...) {
        g_last_error.clear();
    }
    return code;
}

// The exception barrier of the C ABI (include/mpm_hip.h: "errors never kill the caller", the reference's
// settings.h:11-25 contract turned round: CUDA_SAFE_CALL throws only in DEBUG builds, nothing else does).  Every
// extern "C" entry point is a function-try-block that ends in MPM_CATCH_ALL: a C++ exception raised anywhere below it
// (std::bad_alloc / std::length_error from a container sized by a count that came off the device or out of a buffer, a
// std::function that is empty, an ofstream failure with exceptions on ...) becomes a status code and a message instead
// of std::terminate -- which, inside Drake, would take the whole simulation process down.
static int exception_to_status() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(MPM_ERR_NOMEM, "host memory exhausted (std::bad_alloc) inside the engine");
    } catch (const std::exception& ex) {
        try {
            return fail(MPM_ERR_INTERNAL, std::string("C++ exception inside the engine: ") + ex.what());
        } catch (...) {
            return fail(MPM_ERR_INTERNAL, "C++ exception inside the engine");
        }
    } catch (...) {
        return fail(MPM_ERR_INTERNAL, "unknown C++ exception inside the engine");
    }
}
#define MPM_CATCH_ALL \
    catch (...) { return exception_to_status(); }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(e__ == hipErrorOutOfMemory ? MPM_ERR_NOMEM : MPM_ERR_HIP,                       \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                            \
    } while (0)

#define REQUIRE(cond, msg)                                   \
    do {                                                     \
        if (!(cond)) return fail(MPM_ERR_INVALID, (msg));    \
    } while (0)

struct mpm_engine {
    std::atomic<int> pins{0};   // mpm_device_synchronize of another thread is working on this engine (mpm_destroy waits)
    int device = 0;
    int bits = 7;
    mpm_material_t mat{};
    hipStream_t stream = nullptr, own_stream = nullptr;
    bool finalized = false;
    // staged cloth (AddQRCloth)
    std::vector<float> h_pos, h_vel;
    std::vector<int> h_idx;  // vertex ids local to the vertex array
    size_t nv = 0, nf = 0, np = 0;
    DP dp{};
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;   // (parallel to allocs)
    // slot (API) order bookkeeping: slot -> original id and its inverse
    int* d_pids_api = nullptr;
    int* d_apimap = nullptr;
    // RebuildMapping(sort = true): scratch of the device radix sort, second pids buffer
    uint32_t *d_sort_keys = nullptr, *d_sort_vals = nullptr, *d_sort_keys2 = nullptr, *d_sort_vals2 = nullptr;
    int* d_sort_hist = nullptr;
    float last_dt = 0.f;   // length of the last substep (anticipatory binning of the re-sort)
    // gated substeps of mpm_run_substeps (see gated_out in mpm_step.h)
    // the re-sort launches precede every check_every-th substep of mpm_run_substeps (1 = all); read from
    // MPM_RESORT_EVERY when the engine is created, like graph_len (MPM_GRAPH): per handle, not per process
    int check_every = 4;
    int graph_len = 0;         // > 0: mpm_run_substeps replays captured graphs of this many substeps
    unsigned step_phase = 0;
    bool force_check = true;   // the next substep gets them whatever its number (after any other call)
    float quiet_left = 0.f;    // seconds of Ctl::quiet_time left as of the last settle() (0: unknown)
    // phase calls of a substep in the reference's order that have been accepted but not launched (mpm_engine.hip,
    // mpm_rebuild_mapping): n of RebuildMapping(false), CalcFemStateAndForce(dt), ParticleToGrid(dt), UpdateGrid(bc)
    struct PendingPhases {
        int n = 0;
        float dt = 0.f;
        int bc = 0;
    } pend;
    int chain_lean = 0;             // mpm_chain_substeps: another substep of the batch follows (DP::lean_g2p)
    bool defer_phases = true;       // MPM_DEFER_PHASES=0: every phase call launches its kernels at once
    uint64_t checks_launched = 0;   // (diagnostics)
    Ctl* h_ctl = nullptr;           // pinned landing place of the control block (settle)
    bool settle_read_ctl = false;   // settle() handed its copy of the control block to the caller (mpm_sync)
    float quiet_factor = .5f;  // share of it that is trusted (MPM_QUIET_FACTOR; 0 = check launches as before)
    bool maybe_owed = false;   // gated substeps were enqueued since the last settle()
    float owed_dt = 0.f;
    int owed_bc = 0;
    uint64_t owed_gcv = 0;
    int* d_pids_api2 = nullptr;
    int* d_iota = nullptr;     // identity map, created on first use (views in original order)
    bool api_identity = true;
    bool deterministic = getenv("MPM_DETERMINISTIC") != nullptr;  // see mpm_set_deterministic
    // mpm_set_fast_math / MPM_FAST_MATH=1: k_fem's divisions and square roots by the hardware approximation + one Newton
    // step (mpm_math.h, FM = 1) instead of correctly rounded (the default)
    bool fast_math = getenv("MPM_FAST_MATH") != nullptr && atoi(getenv("MPM_FAST_MATH")) != 0;
    bool p2g_fixed_point = getenv("MPM_P2G_FIXED") != nullptr;    // the fixed-point LDS tile of k_p2g outside deterministic mode too (A/B)
    // native chain (mpm_chain_*): communicator, neighbours, device buffers
    struct Chain {
        void* comm = nullptr;
        int rank = 0, world = 1, left = -1, right = -1;
        int zone_lo[2] = {0, 0}, zone_hi[2] = {0, 0}, pitch = 0;
        size_t cap = 0, bytes = 0;
        void *send_l = nullptr, *send_r = nullptr, *recv_l = nullptr, *recv_r = nullptr;
        // partitioned domain: migration records every mig_every substeps
        int mig_every = 0;         // 0 with mig_cap > 0: adaptive (see mpm_chain_substeps)
        float mig_budget = 0.f, mig_elapsed = 0.f;   // adaptive cadence: seconds until the next migration / since the last
        float* mig_quiet_all = nullptr;              // device: the ranks' common estimate (ncclMin)
        size_t mig_cap = 0, mig_bytes = 0;
        void *mig_send_l = nullptr, *mig_send_r = nullptr, *mig_recv_l = nullptr, *mig_recv_r = nullptr;
        uint64_t steps = 0;
        // DIRECT halo (mpm_chain_direct_*): the pack kernel stores the zone sums straight into the NEIGHBOUR's receive
        // buffer (peer memory mapped through an IPC handle), a one-thread kernel raises a sequence flag over there, the
        // receiver's one-workgroup kernel waits for it: no RCCL kernel, no staging copy on the per-substep path.
        // One allocation per rank: [from-left parity 0 | from-left parity 1 | from-right 0 | from-right 1 | flags].
        bool direct = false;
        void* direct_base = nullptr;            // this rank's allocation (fine-grained device memory)
        void* peer_base[2] = {nullptr, nullptr};   // the left / right neighbour's, as mapped here
        bool peer_mapped[2] = {false, false};      // ... through hipIpcOpenMemHandle (else: this rank itself)
        float direct_timeout_s = 5.f;
        bool direct_mute = false;   // MPM_HALO_DEBUG_MUTE (tests): the signal kernel is left out
        uint32_t* direct_cnt = nullptr;   // [2] entry counters of the two zones a substep packs, in THIS device's memory
        bool direct_coarse = false;       // the region is NOT fine-grained (MPM_DIRECT_COARSE_OK=1: one-device rehearsals only)
    } chain;
    // TEAM transport of the distributed contact solve (mpm_team.h, mpm_team_prepare / _connect): this rank's region, every
    // rank's region as mapped here, the device-side exchange counters
    struct Team {
        bool on = false;
        int rank = 0, world = 1;
        void* base = nullptr;                 // this rank's region (fine-grained device memory)
        void* peer[TEAM_MAX] = {};            // every rank's, as this rank addresses it (peer[rank] == base)
        bool mapped[TEAM_MAX] = {};           // ... through hipIpcOpenMemHandle (else: a pointer of this process)
        size_t zone_cap = 0, zone_bytes = 0;
        TeamState* ts = nullptr;
        bool coarse = false;                  // NOT fine-grained (MPM_DIRECT_COARSE_OK=1: one-device rehearsals only)
        float timeout_s = 5.f;
    } team;
    bool halo_mid_done = false;   // mpm_substep_mid_halo ran in this substep
    int halo_nz = 0, halo_zlo[2] = {0, 0}, halo_zhi[2] = {0, 0};
    unsigned g_rb = 2048;  // workgroups of the particle-parallel re-sort kernels
    int grid_state = 0;  // 0 nothing, 1 slabs valid (after P2G), 2 grid updated
    uint64_t substeps = 0;
    // one captured substep (run_substeps replays it): launch arguments are all by-value constants
    hipGraphExec_t step_graph = nullptr;
    float step_graph_dt = 0.f;
    int step_graph_bc = 0;
    uint64_t step_graph_gcv = 0;
    // analytic colliders of the grid update selected by mpm_bc = MPM_BC_TABLE (mpm_set_grid_colliders)
    GridColliders grid_colliders{};
    uint64_t grid_colliders_version = 0;
    int step_graph_len = 1;
    hipStream_t step_graph_stream = nullptr;
    // the two halves of the multi-GPU substep as replayable graphs (host enqueue time matters there)
    // (one pair per parity of the direct transport's receive slots: its buffer addresses alternate with every substep,
    // and a single cached graph would be re-captured every time -- ADVICE r5)
    struct KeyedGraph {
        hipGraphExec_t exec = nullptr;
        std::vector<uint64_t> key;
    } halo_graph[4];
    int halo_graph_parity = 0;   // which pair the next begin / end halo call replays (mpm_chain_substeps, direct transport)
    int max_valence = 0;       // most faces around one vertex of the mesh (Finalize): above 8, k_vforce stays (see fused_forces)
    int last_tile_kernel = 0;  // 1 = P2G, 2 = G2P (see launch_p2g)
    // launch geometry
    unsigned g_np = 0, g_nf = 0, g_nv = 0, g_tile = 0, g_grid = 0;
    // contacts / rigid bodies
    ContactBuffers cb{};
    mpm_contact_stats_t last_contact{};   // of the last mpm_update_contact
    bool last_contact_on_device = false;  // ... of which only what the mailbox carries has reached the host (contact_stats_from_device)
    bool last_contact_exact = false;
    bool last_contact_gated = false;      // ... skipped itself with its whole substep (CT_DONE_GATED)
    float ct_quiet_left = 0.f;            // Ctl::quiet_time left as of the last solve's publication
    std::function<void()> ct_before_impulse;   // (mpm_run_coupled_substeps) launched between the solve's last update and its impulses
    bool last_contact_reused = false;     // ... ran on the previous solve's sorted order and node list (settled scene)
    // MPM_CT_NO_REUSE=1: every solve runs the full set-up (sort, per-cell runs, node list), also when the pair list repeats
    bool ct_no_reuse = getenv("MPM_CT_NO_REUSE") != nullptr;
    // MPM_CT_INITIAL_CAPACITY: contacts the buffers of device-made pairs hold at first (tests: a small value makes the
    // overflow-and-repeat path run; default: a sixteenth of the particles, at least 4096)
    size_t ct_initial_capacity = getenv("MPM_CT_INITIAL_CAPACITY") ? (size_t)std::max(1, atoi(getenv("MPM_CT_INITIAL_CAPACITY"))) : 0;
    // MPM_CT_GATE_ALWAYS=1 (tests): mpm_run_coupled_substeps never sends the re-sort launches ahead of a substep on its own
    // estimate: every re-sort is found by a substep skipping itself
    bool ct_gate_always = getenv("MPM_CT_GATE_ALWAYS") != nullptr;
    double ct_wait_us = 0;   // (MPM_CT_DEBUG) host time spent polling the mailbox
    uint64_t ct_counters[6] = {0, 0, 0, 0, 0, 0};   // solves, of them on a reused set-up, refused as stale, repeated after an overflow,
                                                     // coupled substeps that ran contact-free on a watch, of them skipped and repeated
    unsigned watch_seq = 0;                          // number of the last k_ct_watch launch (Ctl::watch_hit)
    bool ct_no_watch = getenv("MPM_CT_NO_WATCH") != nullptr;   // every coupled substep generates pairs and solves (A/B, tests)
    float last_contact_dt = 0.f, last_contact_mu = 0.f, last_contact_k = 0.f, last_contact_d = 0.f;   // its parameters
    mpm_dist_config_t dist_cfg{};         // partitioned domain (mpm_dist_init)
    // slot space of a partitioned rank = headroom x what it holds (mpm_dist_set_headroom, MPM_DIST_HEADROOM; 0 = the whole
    // scene's size, no shrink); grown at the migration that would overflow it (dist_resize)
    float dist_headroom = getenv("MPM_DIST_HEADROOM") ? std::max(0.f, (float)atof(getenv("MPM_DIST_HEADROOM"))) : 1.5f;
    // bands from the mesh: the drift (cells along x) between two migrations that the ghost bands are sized for, if the
    // zone allows that much (MPM_DIST_DRIFT).  Wider bands: fewer migrations, more ghost particles to advance.
    float dist_drift_target = getenv("MPM_DIST_DRIFT") ? std::max(.06f, (float)atof(getenv("MPM_DIST_DRIFT"))) : .5f;
    float dist_longest_edge = 0.f;        // cells, from the mesh handed to AddQRCloth
    // bands from the mesh are RE-TUNED at migrations (mpm_dist_retune): the drift budget follows the speed the ranks
    // measure, so that a migration is due every dist_target_interval substeps or so -- wide bands for a cloth that moves
    // along x, narrow ones (few ghosts) for one that does not.  MPM_DIST_RETUNE=0 keeps the bands of mpm_dist_init.
    bool dist_auto = false;               // the bands came from the mesh
    bool dist_retune = getenv("MPM_DIST_RETUNE") ? atoi(getenv("MPM_DIST_RETUNE")) != 0 : true;
    float dist_reach = 0.f, dist_hyst = 0.f, dist_delta_max = 0.f;   // cells: a face's reach, the hysteresis, the zone's limit
    float dist_target_interval = getenv("MPM_DIST_INTERVAL") ? std::max(2.f, (float)atof(getenv("MPM_DIST_INTERVAL"))) : 16.f;
    uint32_t dist_retunes = 0;
    // share of the ranks' common quiet-time estimate after which they migrate again (the estimate is ballistic, elastic
    // forces are not in it); MPM_MIG_SAFETY
    float mig_safety = getenv("MPM_MIG_SAFETY") ? std::min(1.f, std::max(.05f, (float)atof(getenv("MPM_MIG_SAFETY")))) : .5f;
    uint32_t dist_resizes = 0, dist_migrations = 0;
    // transport of the distributed contact solve when there is no native chain (mpm_dist_set_transport)
    mpm_exchange_fn dist_exchange = nullptr;
    mpm_allreduce_fn dist_allreduce = nullptr;
    void* dist_user = nullptr;
    size_t dist_zone_cap = 1024;
    std::string dump_dir = ".";
    // scratch for downloads
    void* d_stage = nullptr;
    size_t stage_bytes = 0;

    // Environment switches are read once per HANDLE, when the engine object is made (mpm_create): two engines of one
    // process can differ, and nothing is cached per process.
    // MPM_POISON=1 (tests): buffers that are not zero-initialised start as 0xFF bytes (NaN floats,
    // -1 ints), so that a read of something never written shows up on every box
    bool poison_fill = getenv("MPM_POISON") != nullptr;
    bool poison() const { return poison_fill; }
    // MPM_ANTICIPATE: horizon (substeps) of the re-sort's anticipatory binning (launch_rebuild)
    float anticipate_horizon = getenv("MPM_ANTICIPATE") ? (float)atof(getenv("MPM_ANTICIPATE")) : 32.f;
    // MPM_HALO_GRAPH=1: the two halves of a chain substep are replayed from captured graphs
    bool use_halo_graphs = getenv("MPM_HALO_GRAPH") != nullptr && atoi(getenv("MPM_HALO_GRAPH")) != 0;
    // MPM_CT_EAGER=1: the contact solve applies every accepted step with a kernel of its own (update_contact)
    bool ct_eager = getenv("MPM_CT_EAGER") != nullptr;
    // MPM_CT_RELAX: Jacobi relaxation of the contact solve instead of the reference's 0.3 (tests: overshoot on purpose)
    float ct_relax = getenv("MPM_CT_RELAX") ? (float)atof(getenv("MPM_CT_RELAX")) : 0.f;
    // MPM_CT_BATCH="first,next": Newton iterations enqueued per batch of the contact solve (measurements)
    int ct_batch[2] = {0, 0};
    bool ct_debug = getenv("MPM_CT_DEBUG") != nullptr;
    bool ct_force_dist = getenv("MPM_CT_FORCE_DIST") != nullptr;   // (tests, see update_contact)
    mpm_engine() {
        if (const char* t = getenv("MPM_CT_BATCH")) sscanf(t, "%d,%d", &ct_batch[0], &ct_batch[1]);
    }
    // tests (mpm_debug_fail_alloc): the n-th dalloc from now reports hipErrorOutOfMemory without asking the runtime
    int fail_alloc_countdown = 0;
    template <class T>
    int dalloc(T** out, size_t n, bool zero) {
        void* ptr = nullptr;
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        if (fail_alloc_countdown > 0 && --fail_alloc_countdown == 0)
            return fail(MPM_ERR_NOMEM, "device allocation failed (injected by mpm_debug_fail_alloc)");
        HIP_TRY(hipMalloc(&ptr, bytes));
        allocs.push_back(ptr);
        alloc_bytes.push_back(bytes);
        if (zero) HIP_TRY(hipMemsetAsync(ptr, 0, bytes, stream));
        else if (poison()) HIP_TRY(hipMemsetAsync(ptr, 0xFF, bytes, stream));
        *out = static_cast<T*>(ptr);
        return 0;
    }
    // releases one dalloc'ed array (mpm_dist_init: the whole-scene sized arrays a partitioned rank gives back)
    template <class T>
    void dfree(T*& ptr) {
        if (!ptr) return;
        for (size_t k = 0; k < allocs.size(); ++k)
            if (allocs[k] == (void*)ptr) {
                (void)hipFree(allocs[k]);
                allocs.erase(allocs.begin() + (long)k);
                alloc_bytes.erase(alloc_bytes.begin() + (long)k);
                break;
            }
        ptr = nullptr;
    }
    size_t bytes_of(const void* ptr) const {
        for (size_t k = 0; k < allocs.size(); ++k)
            if (allocs[k] == ptr) return alloc_bytes[k];
        return 0;
    }
    int stage(size_t bytes) {
        if (bytes > stage_bytes) {
            if (d_stage) HIP_TRY(hipFree(d_stage));
            d_stage = nullptr;
            HIP_TRY(hipMalloc(&d_stage, bytes));
            if (poison()) HIP_TRY(hipMemsetAsync(d_stage, 0xFF, bytes, stream));
            stage_bytes = bytes;
        }
        return 0;
    }
};

// Blocking copies ORDERED ON THE ENGINE'S STREAM.  That stream is non-blocking: the legacy null
// stream behind plain hipMemcpy/hipMemset does not wait for it and is not waited for by it, so
// a plain hipMemcpy can overtake a queued hipMemsetAsync or kernel (seen as rare garbage in
// freshly finalised engines).  Nothing in the engine may use the synchronous calls.
#define H2D(e, dst, src, bytes)                                                                    \
    do {                                                                                           \
        HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, (e)->stream));        \
        HIP_TRY(hipStreamSynchronize((e)->stream));                                                \
    } while (0)
#define D2H(e, dst, src, bytes)                                                                    \
    do {                                                                                           \
        HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, (e)->stream));        \
        HIP_TRY(hipStreamSynchronize((e)->stream));                                                \
    } while (0)

static int use(mpm_engine* e) {
    HIP_TRY(hipSetDevice(e->device));
    return 0;
}

