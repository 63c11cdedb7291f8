// Host side of the engine and the C ABI declared in include/mpm_hip.h.
// One translation unit: the kernels are header-only so hipcc can inline the
// device math into them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mpm_hip.h"
#include "mpm_host.h"
#include "mpm_io.h"
#include "mpm_contact.h"
#include "mpm_chain.h"
#include "mpm_feedback.h"
#include "mpm_dist.h"

// every engine alive in this process (mpm_device_synchronize)
static std::mutex g_live_mutex;
static std::vector<mpm_engine*> g_live;

extern "C" {

const char* mpm_last_error(void) { return g_last_error.c_str(); }

int mpm_default_material(mpm_material_t* m) try {
    REQUIRE(m, "null material");
    m->youngs_modulus = 400000.f;
    m->poisson_ratio = .3f;
    m->density = 2000.f;
    m->gamma = 0.f;
    m->K = 100000.f;
    m->V = .8f;
    m->c_F = 0.f;
    m->sdf_friction = .3f;
    m->gravity = -9.8f;
    m->epsv = 1e-3f;
    m->gravity_axis = 2;
    m->wall_cells = 3;
    return 0;
} MPM_CATCH_ALL

int mpm_create(int domain_bits, const mpm_material_t* material, int device, mpm_handle_t* out) try {
    REQUIRE(out, "null handle pointer");
    REQUIRE(domain_bits >= 4 && domain_bits <= 8, "domain_bits must be in [4,8] (16^3 .. 256^3 cells)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MPM_ERR_NO_DEVICE, "no HIP device visible: the engine has no CPU fallback");
    REQUIRE(device >= 0 && device < ndev, "device ordinal out of range");
    mpm_engine* e = new mpm_engine();
    e->device = device;
    e->bits = domain_bits;
    if (material) e->mat = *material; else mpm_default_material(&e->mat);
    if (e->mat.gravity_axis < 0 || e->mat.gravity_axis > 2) {
        delete e;
        return fail(MPM_ERR_INVALID, "gravity_axis must be 0, 1 or 2");
    }
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        delete e;
        return fail(MPM_ERR_HIP, std::string("mpm_create: ") + hipGetErrorString(err));
    }
    e->stream = e->own_stream;
    if (getenv("MPM_RESORT_EVERY")) e->check_every = std::max(1, atoi(getenv("MPM_RESORT_EVERY")));
    if (getenv("MPM_DEFER_PHASES")) e->defer_phases = atoi(getenv("MPM_DEFER_PHASES")) != 0;
    if (getenv("MPM_QUIET_FACTOR")) e->quiet_factor = std::min(1.f, std::max(0.f, (float)atof(getenv("MPM_QUIET_FACTOR"))));
    if (getenv("MPM_GRAPH")) e->graph_len = std::max(0, atoi(getenv("MPM_GRAPH")));
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.push_back(e);
    }
    *out = e;
    return 0;
} MPM_CATCH_ALL

int mpm_add_qr_cloth(mpm_handle_t e, const float* pos, const float* vel, size_t n_verts, const int32_t* indices,
                     size_t n_faces) try {
    REQUIRE(e, "null handle");
    REQUIRE(!e->finalized, "AddQRCloth after Finalize");
    REQUIRE(pos && vel && (indices || n_faces == 0), "null input array");
    for (size_t i = 0; i < n_faces * 3; ++i) {
        REQUIRE(indices[i] >= 0 && (size_t)indices[i] < n_verts, "triangle index out of range");
        e->h_idx.push_back(indices[i] + (int)e->nv);
    }
    e->h_pos.insert(e->h_pos.end(), pos, pos + 3 * n_verts);
    e->h_vel.insert(e->h_vel.end(), vel, vel + 3 * n_verts);
    e->nv += n_verts;
    e->nf += n_faces;
    e->np = e->nv + e->nf;
    return 0;
} MPM_CATCH_ALL

// ---- analytic colliders of the grid update ---------------------------------------------------
// The reference's compile-time scenes (cuda_mpm_kernels.cuh:673-774) as tables; MPM_BC_TABLE selects
// the table set with mpm_set_grid_colliders.
static GridCollider gc_sphere(float x, float y, float z, float r, int mode, float friction) {
    GridCollider c{};
    c.shape = 0; c.mode = mode;
    c.p[0] = x; c.p[1] = y; c.p[2] = z;
    c.radius = r; c.friction = friction;
    return c;
}
static int grid_collider_preset(int bc, float sdf_friction, GridColliders* out) {
    out->n = 0;
    switch (bc) {
        case -1: return 0;
        case 0:   // one sphere, slip, active while approaching (:673-692)
            out->c[out->n++] = gc_sphere(.5f, .5f, .5f, .08f, 1, sdf_friction);
            return 0;
        case 1:   // two fixed spheres (:694-734)
            out->c[out->n++] = gc_sphere(.38f, .38f, .75f, .04f, 0, sdf_friction);
            out->c[out->n++] = gc_sphere(.38f, .62f, .75f, .04f, 0, sdf_friction);
            return 0;
        case 2: { // plane z < 0.11, slip, active whenever inside (:737-749)
            GridCollider c{};
            c.shape = 1; c.mode = 2;
            c.p[2] = .11f; c.n[2] = 1.f; c.friction = sdf_friction;
            out->c[out->n++] = c;
            return 0;
        }
        case 3: { // four pin spheres (:752-774)
            const float span[2] = {.3f, .7f};
            for (int k = 0; k < 4; ++k) out->c[out->n++] = gc_sphere(span[k % 2], span[k / 2], .5f, .02f, 0, sdf_friction);
            return 0;
        }
        default: return -1;
    }
}
static int grid_colliders_for(mpm_engine* e, int bc, GridColliders* out) {
    if (bc == MPM_BC_TABLE) {
        *out = e->grid_colliders;
        return 0;
    }
    if (grid_collider_preset(bc, e->mat.sdf_friction, out))
        return fail(MPM_ERR_INVALID, "mpm_bc must be -1, 0, 1, 2, 3 or MPM_BC_TABLE");
    return 0;
}

static void launch_rebuild(mpm_engine* e) {
    // anticipatory binning over the next `horizon` substeps of the last known length (not in a partitioned
    // domain, where ownership and ghost bands are defined by the position itself)
    e->dp.anticip = e->dp.dist.on ? 0.f : e->anticipate_horizon * e->last_dt * e->dp.dxinv;
    const DP& p = e->dp;
    e->checks_launched += 1;
    TraceRange tr("mpm:RebuildMapping (conditional re-sort)");
    // (host state only -- nothing below may depend on this function running: captured graphs replay the launches
    // without it.  Callers that enqueue a possible re-sort call may_resort() themselves.)
    hipLaunchKernelGGL(k_rb_count, dim3(std::min(e->g_np, e->g_rb)), dim3(256), 0, e->stream, p);
    hipLaunchKernelGGL(k_rb_tables, dim3(33), dim3(1024), 0, e->stream, p);
    hipLaunchKernelGGL(k_rb_scatter, dim3(std::min(e->g_np, e->g_rb)), dim3(256), 0, e->stream, p);
    if (e->deterministic) hipLaunchKernelGGL(k_rb_canon, dim3(512), dim3(256), 0, e->stream, p);
    hipLaunchKernelGGL(k_rb_finish, dim3((std::min(e->g_np, e->g_rb) + 7u) & ~7u), dim3(256), 0, e->stream, p);
}
// A re-sort may be about to run (directly or inside a replayed graph): the block tables may change, so contact
// pairs handed over before are re-keyed with the full width, and the anticipatory binning uses this substep length.
static void may_resort(mpm_engine* e, float dt) {
    // (the contact solve's guess of how many blocks are active survives a re-sort: k_ct_keys verifies it on the device)
    if (dt > 0.f) e->last_dt = dt;
}
static void drop_step_graph(mpm_engine* e) {
    if (e->step_graph) (void)hipGraphExecDestroy(e->step_graph);
    e->step_graph = nullptr;
}

static void launch_fem_faces(mpm_engine* e, float dt) {
    TraceRange tr("mpm:CalcFemStateAndForce (faces)");
    if (!e->nf) return;
    if (e->fast_math) hipLaunchKernelGGL(k_fem<1>, dim3((e->g_nf + 7u) & ~7u), dim3(256), 0, e->stream, e->dp, dt);
    else hipLaunchKernelGGL(k_fem<0>, dim3((e->g_nf + 7u) & ~7u), dim3(256), 0, e->stream, e->dp, dt);
}
static void launch_fem_vertices(mpm_engine* e) {
    TraceRange tr("mpm:CalcFemStateAndForce (vertex forces)");
    if (e->nv) hipLaunchKernelGGL(k_vforce, dim3((e->g_nv + 7u) & ~7u), dim3(256), 0, e->stream, e->dp);
}
static void launch_fem(mpm_engine* e, float dt) {
    e->last_dt = dt;
    launch_fem_faces(e, dt);
    launch_fem_vertices(e);
}
// `forces`: where k_p2g's vertex lanes find the force on their vertex (the kernel's template parameter: 0 = in p.f,
// k_vforce ran; 1 = summed inside the kernel)
static void launch_p2g(mpm_engine* e, float dt, int forces = 0) {
    TraceRange tr(forces ? "mpm:ParticleToGrid (+ vertex forces)" : "mpm:ParticleToGrid");
    const dim3 g(e->g_tile), b(P2G_THREADS);
    // (deterministic mode accumulates in fixed point: exact sums whatever the order of arrival, see k_p2g's EXACT)
    const bool exact = e->deterministic || e->p2g_fixed_point;
#define MPM_P2G_LAUNCH(F)                                                                          \
    do {                                                                                           \
        if (exact) hipLaunchKernelGGL((k_p2g<F, 1>), g, b, 0, e->stream, e->dp, dt);               \
        else hipLaunchKernelGGL((k_p2g<F, 0>), g, b, 0, e->stream, e->dp, dt);                     \
    } while (0)
    if (forces == 1) MPM_P2G_LAUNCH(1);
    else MPM_P2G_LAUNCH(0);
#undef MPM_P2G_LAUNCH
    e->last_tile_kernel = 1;
}
// the vertex forces inside k_p2g, from the vertices' entries of DP::VF (single and partitioned domains alike since round
// 5); a mesh with a vertex of more than eight faces keeps the k_vforce launch
static int fused_forces(const mpm_engine* e) { return e->max_valence <= 8 ? 1 : 0; }
// FEM faces, then P2G with the vertex forces of every work item computed inside it (no k_vforce launch): the
// batched substeps use this; the phase-by-phase calls keep the two FEM kernels, whose forces a caller may read
static void launch_fem_p2g(mpm_engine* e, float dt) {
    e->last_dt = dt;
    launch_fem_faces(e, dt);
    const int forces = fused_forces(e);
    if (!forces) launch_fem_vertices(e);
    launch_p2g(e, dt, forces);
}
// (`p` may carry a halo class restriction)
static void launch_g2p_with(mpm_engine* e, DP p, float dt) {
    TraceRange tr("mpm:GridToParticle");
    hipLaunchKernelGGL(k_g2p, dim3(std::min(768u, p.capI) * G2P_SPLIT), dim3(G2P_THREADS), 0, e->stream, p, dt);
    e->last_tile_kernel = 2;
}
static void launch_grid(mpm_engine* e, const GridColliders& gc) {
    TraceRange tr("mpm:UpdateGrid");
    hipLaunchKernelGGL(k_grid<1>, dim3(e->g_grid), dim3(256), 0, e->stream, e->dp, gc);
}
static void launch_g2p(mpm_engine* e, float dt) { launch_g2p_with(e, e->dp, dt); }

// The P2G tiles accumulate in 64-bit fixed point.  Scales are powers of two chosen from the
// total particle mass M: a node can never hold more than M, and its momentum is allowed
// |v| < 2^15 length units per time unit.  Resolution: M * 2^-61 (mass), M * 2^-47 (momentum).
static int recover_slab_overflow(mpm_engine* e, Ctl& c);
static int set_fixed_point_scales(mpm_engine* e) {
    DP& p = e->dp;
    // (slot space: the whole scene, or -- rare: volumes uploaded to a partitioned engine -- what this rank holds;
    // the scales of mpm_finalize are the whole scene's on every rank, which keeps ghost copies bit-identical)
    const size_t slots = (size_t)p.Np;
    std::vector<float> q0(slots * 4);
    HIP_TRY(hipStreamSynchronize(e->stream));
    Ctl c;
    D2H(e, &c, p.ctl, sizeof(Ctl));
    D2H(e, q0.data(), p.set[c.cur & 1].q[0], slots * 16);
    double mass = 0;
    for (size_t i = 0; i < slots; ++i) mass += std::fabs((double)q0[i * 4 + 3]) * p.M.density;
    if (!(mass > 0) || !std::isfinite(mass)) mass = 1.0;
    const int k = 61 - (int)std::ceil(std::log2(mass));
    p.fix_m = std::ldexp(1.0, k);
    p.fix_p = std::ldexp(1.0, k - 14);
    p.unfix_m = 1.0 / p.fix_m;
    p.unfix_p = 1.0 / p.fix_p;
    // captured launches carry the scales by value
    if (e->step_graph) (void)hipGraphExecDestroy(e->step_graph);
    e->step_graph = nullptr;
    for (auto& kg : e->halo_graph) {
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
        kg.exec = nullptr;
    }
    return 0;
}

int mpm_finalize(mpm_handle_t e) try {
    REQUIRE(e, "null handle");
    REQUIRE(!e->finalized, "Finalize called twice");
    REQUIRE(e->np > 0, "no particles: call mpm_add_qr_cloth first");
    if (int rc = use(e)) return rc;
    const size_t np = e->np, nf = e->nf, nv = e->nv;
    REQUIRE(np < (size_t)1 << 30, "too many particles");
    DP& p = e->dp;
    p.Np = (int)np; p.Nf = (int)nf; p.Nv = (int)nv;
    p.NpG = (int)np; p.NfG = (int)nf;
    p.bits = e->bits;
    p.dbg = getenv("MPM_DBG") ? atoi(getenv("MPM_DBG")) : 0;
    p.nb = 1 << (e->bits - 2);
    p.ncells = 1u << (3 * e->bits);
    p.nblocks = p.ncells >> 6;
    p.capH = (unsigned)std::min<size_t>(p.nblocks, np);
    p.capA = (unsigned)std::min<size_t>(p.nblocks, (size_t)27 * p.capH);
    p.halo_cls = -1;
    p.fem_fast = e->fast_math ? 1 : 0;
    p.item_groups = getenv("MPM_ITEM_GROUPS") ? std::max(1, atoi(getenv("MPM_ITEM_GROUPS"))) : 48;
    p.item_groups_small = getenv("MPM_ITEM_GROUPS_SMALL") ? std::max(1, atoi(getenv("MPM_ITEM_GROUPS_SMALL"))) : 16;
    p.item_small_below = getenv("MPM_ITEM_SMALL_BELOW") ? atoi(getenv("MPM_ITEM_SMALL_BELOW")) : 6500;
    p.capI = p.capH + (unsigned)(np / (64 * (size_t)std::min(p.item_groups, p.item_groups_small))) + 16u;
    // Slabs (16 KB each) are allocated for the blocks a cloth of this size typically occupies, not for
    // the worst case of one particle per block (capI: 4 GB at 256^3, 33 GB at 512^3); mpm_sync and
    // mpm_get_stats double the pool when it is more than half full (slab_pool_grow).
    p.capS = (unsigned)std::min<size_t>(p.capI, std::max<size_t>(4096, np / 256 + 1024 + np / (64 * (size_t)p.item_groups)));
    if (getenv("MPM_SLAB_CAPACITY")) p.capS = (unsigned)std::min<size_t>(p.capI, std::max(1, atoi(getenv("MPM_SLAB_CAPACITY"))));
    p.dxinv = (float)(1 << e->bits);
    p.dx = 1.f / p.dxinv;
    p.Dinv = 4.f * p.dxinv * p.dxinv;
    const mpm_material_t& m = e->mat;
    p.M.mu = m.youngs_modulus / (2.f * (1.f + m.poisson_ratio));
    p.M.lambda = m.youngs_modulus * m.poisson_ratio / ((1.f + m.poisson_ratio) * (1.f - 2.f * m.poisson_ratio));
    p.M.density = m.density; p.M.gamma = m.gamma; p.M.K = m.K; p.M.V = m.V; p.M.cF = m.c_F;
    p.M.sdf_friction = m.sdf_friction; p.M.gravity = m.gravity; p.M.epsv = m.epsv;
    p.M.gravity_axis = m.gravity_axis; p.M.wall = m.wall_cells;

    // ---- device buffers ---------------------------------------------------
    int rc = 0;
    p.q_stride = (unsigned)((np + 63) & ~(size_t)63);
    p.f_stride = (unsigned)((np + 63) & ~(size_t)63);
#define ALLOC(ptr, n, zero)                         \
    if ((rc = e->dalloc(&(ptr), (n), (zero)))) return rc
    ALLOC(p.ctl, 1, true);
    ALLOC(p.dbgbuf, 16, true);
    for (int s = 0; s < 2; ++s) {
        PSet& S = p.set[s];
        // (the four particle planes in one allocation, q_stride apart: k_p2g addresses them from ONE base
        // pointer -- it runs out of scalar registers otherwise)
        {
            float4* base = nullptr;
            ALLOC(base, 4 * (size_t)p.q_stride, true);
            for (int d = 0; d < 4; ++d) S.q[d] = base + (size_t)d * p.q_stride;
        }
        for (int d = 0; d < 4; ++d) ALLOC(S.fq[d], nf, true);
        ALLOC(S.f8, nf, true);
        ALLOC(S.c8, nf, true);
        ALLOC(S.pid, np, true);
    }
    ALLOC(p.ta, std::max<size_t>(nf, 1), true);   // (k_p2g's vertex lanes read element 0 when an item has no face)
    ALLOC(p.G3, 3 * nf, true);
    ALLOC(p.VF, 3 * vf_entry((unsigned)nv + VF_CHUNK, 0), true);   // (whole chunks; k_p2g's face lanes read vertex 0's entries)
    {
        float* base = nullptr;
        ALLOC(base, 3 * (size_t)p.f_stride, true);
        for (int d = 0; d < 3; ++d) p.f[d] = base + (size_t)d * p.f_stride;
    }
    int* idx_orig[3];
    int *adj_off, *adj_fc;
    for (int d = 0; d < 3; ++d) { ALLOC(idx_orig[d], nf, false); p.idx_orig[d] = idx_orig[d]; }
    ALLOC(adj_off, nv + 1, false);
    ALLOC(adj_fc, 3 * nf, false);
    p.adj_off = adj_off; p.adj_fc = adj_fc;
    ALLOC(p.imap, np, false);
    ALLOC(p.pkey, np, false);
    ALLOC(p.prank, np, false);
    ALLOC(p.src_of, np, false);
    ALLOC(p.dst_of, np, false);
    ALLOC(p.tickets, 32 * 32, true);
    for (int t = 0; t < 2; ++t) {
        ALLOC(p.cellcnt[t], p.ncells, true);
        ALLOC(p.blkcnt[t], p.nblocks, true);
        ALLOC(p.blkstart[t], p.nblocks, true);
    }
    ALLOC(p.lut_home, p.nblocks, false);
    ALLOC(p.lut_act, p.nblocks, false);
    HIP_TRY(hipMemsetAsync(p.lut_home, 0xFF, (size_t)p.nblocks * 4, e->stream));  // -1 = not a home block
    HIP_TRY(hipMemsetAsync(p.lut_act, 0xFF, (size_t)p.nblocks * 4, e->stream));
    ALLOC(p.home_bits, p.nblocks / 32 + 1, true);
    ALLOC(p.home_block, p.capH, true);
    ALLOC(p.home_range, p.capH, true);
    ALLOC(p.home_nbr_act, (size_t)p.capH * 27, true);
    ALLOC(p.item_desc, p.capI, true);
    ALLOC(p.item_order, p.capI, true);
    ALLOC(p.item_flat, (size_t)p.capI * 2, true);
    ALLOC(p.item_rng, p.capI, true);
    ALLOC(p.item_pos, p.capI, true);
    ALLOC(p.home_items, p.capH, true);
    ALLOC(p.act_nbr_items, (size_t)p.capA * 27, true);
    ALLOC(p.home_ngroups, p.capH, true);
    ALLOC(p.home_groups, np / 64 + p.capH + 2, false);
    ALLOC(p.act_block, p.capA, true);
    ALLOC(p.act_nbr_home, (size_t)p.capA * 27, true);
    HIP_TRY(hipMalloc((void**)&p.slab, (size_t)p.capS * TILE_N * sizeof(float4)));   // (not in `allocs`: regrown)
    HIP_TRY(hipMemsetAsync(p.slab, 0, (size_t)p.capS * TILE_N * sizeof(float4), e->stream));
    ALLOC(p.slab_mask, p.capI, true);
    ALLOC(p.gv, (size_t)p.capA * 64, true);
    ALLOC(p.gvs, (size_t)p.capA * 64, true);
    ALLOC(e->d_pids_api, np, false);
    ALLOC(e->d_apimap, np, false);
    float4* dm_orig;
    ALLOC(dm_orig, nf, false);
    p.dm_orig = dm_orig;
#undef ALLOC

    // ---- host-side layout: [faces | verts], indices offset by +nf ---------
    // (cuda_mpm_model.cu:40-45)
    PSet& S0 = p.set[0];
    {
        std::vector<float> q0(np * 4, 0.f), q1(np * 4, 0.f);   // (x,y,z,vol) and (vx,vy,vz,C8); faces are filled on the device
        for (size_t i = 0; i < nv; ++i)
            for (int d = 0; d < 3; ++d) {
                q0[(nf + i) * 4 + d] = e->h_pos[i * 3 + d];
                q1[(nf + i) * 4 + d] = e->h_vel[i * 3 + d];
            }
        H2D(e, S0.q[0], q0.data(), np * 16);
        H2D(e, S0.q[1], q1.data(), np * 16);
    }
    std::vector<int> iota(np);
    std::iota(iota.begin(), iota.end(), 0);
    H2D(e, S0.pid, iota.data(), np * 4);
    H2D(e, p.imap, iota.data(), np * 4);
    H2D(e, e->d_pids_api, iota.data(), np * 4);
    H2D(e, e->d_apimap, iota.data(), np * 4);
    std::vector<int> col(nf);
    for (int d = 0; d < 3; ++d) {
        for (size_t f = 0; f < nf; ++f) col[f] = e->h_idx[f * 3 + d] + (int)nf;
        H2D(e, idx_orig[d], col.data(), nf * 4);
    }
    // corner vertex slots ride in fq[3].yzw (slot == original id before the first sort)
    std::vector<int> f3(nf * 4, 0);
    for (size_t f = 0; f < nf; ++f)
        for (int d = 0; d < 3; ++d) f3[f * 4 + 1 + d] = e->h_idx[f * 3 + d] + (int)nf;
    // vertex -> (face, corner) adjacency, ascending face id
    std::vector<int> off(nv + 1, 0), fc(3 * nf);
    for (size_t k = 0; k < 3 * nf; ++k) off[e->h_idx[k] + 1]++;
    for (size_t v = 0; v < nv; ++v) off[v + 1] += off[v];
    {
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (size_t f = 0; f < nf; ++f)
            for (int c = 0; c < 3; ++c) {
                const int v = e->h_idx[f * 3 + c];
                const int at = fill[v]++;
                fc[at] = (int)(f << 2) | c;
                // the face's rank around this corner's vertex (DP::VF), 15 = the vertex has more than eight faces
                const int rank = off[v + 1] - off[v] <= 8 ? at - off[v] : 15;
                f3[f * 4] |= rank << (4 * c);
            }
    }
    if (nf) H2D(e, S0.fq[3], f3.data(), nf * 16);
    e->max_valence = 0;
    for (size_t v = 0; v < nv; ++v) e->max_valence = std::max(e->max_valence, off[v + 1] - off[v]);
    H2D(e, adj_off, off.data(), (nv + 1) * 4);
    if (nf) H2D(e, adj_fc, fc.data(), 3 * nf * 4);

    // ---- launch geometry --------------------------------------------------
    e->g_np = (unsigned)((np + 255) / 256);
    e->g_nf = (unsigned)((nf + 255) / 256);
    e->g_nv = (unsigned)((nv + 255) / 256);
    e->g_tile = std::min(512u, p.capI);  // 2 resident workgroups per CU pulling blocks from a queue
    if (getenv("MPM_P2G_WGS")) e->g_tile = (unsigned)std::max(1, atoi(getenv("MPM_P2G_WGS")));   // (tuning experiments)
    e->g_grid = std::min(1024u, (p.capA + 3) / 4);
    e->g_rb = getenv("MPM_RB_WGS") ? (unsigned)atoi(getenv("MPM_RB_WGS")) : 2048u;

    // ---- FEM initialisation (cuda_mpm_kernels.cuh:13-70) + first sort -----
    if (nf) hipLaunchKernelGGL(k_init_faces, dim3(e->g_nf), dim3(256), 0, e->stream, p);
    if (nv) hipLaunchKernelGGL(k_init_vertex_adjacency, dim3(e->g_nv), dim3(256), 0, e->stream, p);
    if (nv) hipLaunchKernelGGL(k_init_vertex_volumes, dim3(e->g_nv), dim3(256), 0, e->stream, p);
    if (int rc2 = set_fixed_point_scales(e)) return rc2;
    Ctl c0{};
    c0.cur = 0;
    c0.need_rebuild = 1;
    c0.nfa = (int)nf;
    c0.nva = (int)nv;
    HIP_TRY(hipMemcpyAsync(p.ctl, &c0, sizeof(Ctl), hipMemcpyHostToDevice, e->stream));
    launch_rebuild(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    {
        // a scene with few particles per block needs more slabs than the estimate above: grow and re-sort
        Ctl c;
        D2H(e, &c, p.ctl, sizeof(Ctl));
        if (int rc2 = recover_slab_overflow(e, c)) return rc2;
    }
    if (p.dbg) {
        int a = 0, b = 0, c = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_p2g<1, 0>, P2G_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_g2p, G2P_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, k_fem<0>, 256, 0);
        std::fprintf(stderr, "[mpm_hip] resident workgroups per CU: p2g %d, g2p %d, fem %d\n", a, b, c);
    }
    e->finalized = true;
    if (int rc2 = mpm_sync(e)) return rc2;
    {
        // the first re-sort's quiet time: the first batch of substeps starts with it (launch_substep)
        Ctl c;
        D2H(e, &c, p.ctl, sizeof(Ctl));
        e->quiet_left = c.error || c.need_rebuild ? 0.f : c.quiet_time;
    }
    return 0;
} MPM_CATCH_ALL

int mpm_destroy(mpm_handle_t e) try {
    if (!e) return 0;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), e), g_live.end());
    }
    while (e->pins.load() > 0) std::this_thread::yield();   // (a GpuSync() of another thread is settling this engine)
    hipSetDevice(e->device);
    if (e->own_stream) hipStreamSynchronize(e->own_stream);
    (void)mpm_chain_destroy(e);
    drop_step_graph(e);
    for (auto& kg : e->halo_graph)
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
    for (void* a : e->allocs) hipFree(a);
    if (e->dp.slab) hipFree(e->dp.slab);
    if (e->d_stage) hipFree(e->d_stage);
    if (e->h_ctl) (void)hipHostFree(e->h_ctl);
    e->cb.release();
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    delete e;
    return 0;
} MPM_CATCH_ALL

int mpm_counts(mpm_handle_t e, size_t* nv, size_t* nf, size_t* np) try {
    REQUIRE(e, "null handle");
    if (nv) *nv = e->nv;
    if (nf) *nf = e->nf;
    if (np) *np = e->np;
    return 0;
} MPM_CATCH_ALL

static int settle(mpm_engine* e, Ctl* fresh = nullptr);

int mpm_set_deterministic(mpm_handle_t e, int on) try {
    REQUIRE(e, "null handle");
    if (e->finalized) {
        if (int rc = use(e)) return rc;
        if (int rc = settle(e)) return rc;
    }
    if (e->deterministic != (on != 0)) drop_step_graph(e);   // the captured substep has one kernel more or less
    e->deterministic = on != 0;
    return 0;
} MPM_CATCH_ALL

int mpm_set_fast_math(mpm_handle_t e, int on) try {
    REQUIRE(e, "null handle");
    if (e->finalized) {
        if (int rc = use(e)) return rc;
        if (int rc = settle(e)) return rc;   // (owed substeps run with the arithmetic they were enqueued with)
    }
    if (e->fast_math != (on != 0)) {
        drop_step_graph(e);   // captured launches name the kernel
        for (auto& kg : e->halo_graph) {
            if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
            kg.exec = nullptr;
        }
    }
    e->fast_math = on != 0;
    e->dp.fem_fast = e->fast_math ? 1 : 0;
    return 0;
} MPM_CATCH_ALL

int mpm_get_fast_math(mpm_handle_t e, int* on_out) try {
    REQUIRE(e && on_out, "null argument");
    *on_out = e->fast_math ? 1 : 0;
    return 0;
} MPM_CATCH_ALL

int mpm_set_stream(mpm_handle_t e, void* s) try {
    REQUIRE(e, "null handle");
    if (int rc = use(e)) return rc;
    if (e->finalized)
        if (int rc = settle(e)) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stream = s ? static_cast<hipStream_t>(s) : e->own_stream;
    return 0;
} MPM_CATCH_ALL

// The stream is idle and `c` is the current control block: double the slab pool when the work items
// fill more than half of it.  Slabs only live from ParticleToGrid to UpdateGrid, so between substeps
// there is nothing to copy; between those two calls the old contents are kept.
static int slab_pool_grow(mpm_engine* e, const Ctl& c) {
    DP& p = e->dp;
    const unsigned items = std::max(c.n_items, c.n_items_wanted);
    if (p.capS >= p.capI || items * 2u <= p.capS) return 0;
    unsigned want = p.capS;
    while (want < p.capI && items * 2u > want) want = (unsigned)std::min<size_t>(p.capI, (size_t)want * 2);
    float4* bigger = nullptr;
    HIP_TRY(hipMalloc((void**)&bigger, (size_t)want * TILE_N * sizeof(float4)));
    HIP_TRY(hipMemcpyAsync(bigger, p.slab, (size_t)p.capS * TILE_N * sizeof(float4), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipFree(p.slab));
    p.slab = bigger;
    p.capS = want;
    drop_step_graph(e);   // captured launches carry the old pointer
    for (auto& kg : e->halo_graph) {
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
        kg.exec = nullptr;
    }
    return 0;
}

// The stream is idle and `c` is the control block.  If the last re-sort made more work items than the slab pool
// holds and nothing else is wrong, grow the pool to twice the wanted count, clear the flag and repeat the re-sort;
// `c` is refreshed.  Only legitimate while no substep has used the truncated tables: right after a re-sort
// (mpm_finalize, mpm_dist_init) or when every substep since has skipped itself (DP::gated bit 1).
static int recover_slab_overflow(mpm_engine* e, Ctl& c) {
    for (int round = 0; round < 4 && (c.error & ERR_SLABS); ++round) {
        if (c.error & ~ERR_SLABS) break;        // something else failed too: report it
        if (e->dp.capS >= e->dp.capI) break;    // (cannot happen: capI covers every legal item count)
        if (int rc = slab_pool_grow(e, c)) return rc;
        const unsigned zero = 0;
        const int one = 1;
        H2D(e, &e->dp.ctl->error, &zero, sizeof(unsigned));
        H2D(e, &e->dp.ctl->need_rebuild, &one, sizeof(int));
        may_resort(e, 0.f);
        launch_rebuild(e);
        D2H(e, &c, e->dp.ctl, sizeof(Ctl));
    }
    return 0;
}

// The stream is idle and `c` is the control block as it stands: what is left of the quiet time the last re-sort
// estimated.  Substeps enqueued right after go without check launches while it lasts (launch_substep); any call that
// may change the state resets it (settle_owed).
static void note_quiet_time(mpm_engine* e, const Ctl& c) {
    e->quiet_left = c.skipped || c.need_rebuild || c.error ? 0.f : std::max(0.f, c.quiet_time - c.time_since_resort);
}

int mpm_sync(mpm_handle_t e) try {
    REQUIRE(e, "null handle");
    if (int rc = use(e)) return rc;
    Ctl c;
    e->settle_read_ctl = false;
    if (e->finalized)
        if (int rc = settle(e, &c)) return rc;
    if (!e->settle_read_ctl) {   // (otherwise settle() has just synchronised and read the control block: once is enough)
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipGetLastError());
        if (!e->finalized && !e->dp.ctl) return 0;
        D2H(e, &c, e->dp.ctl, sizeof(Ctl));
    }
    if (e->finalized && !c.error)
        if (int rc = slab_pool_grow(e, c)) return rc;
    if (e->finalized) note_quiet_time(e, c);   // (also when settle() had nothing to do and this call read the block itself)
    if (c.error & ERR_DRIFT)
        return fail(MPM_ERR_DRIFT,
                    "a face particle was re-centred on its corners out of its block's tile: the corner "
                    "velocities differ by cells per substep, the state is diverging -- reduce dt");
    if (c.error & ERR_DOMAIN)
        return fail(MPM_ERR_DOMAIN, "a particle left the grid; results are invalid");
    if (c.error & ERR_HALO)
        return fail(MPM_ERR_HALO,
                    "partitioned domain: a particle left the zone shared with the neighbouring rank (or reached a "
                    "rank that is not a neighbour): migrate more often, or widen the zone / ghost band");
    if (c.error & ERR_RANGE)
        return fail(MPM_ERR_RANGE,
                    "a node sum of ParticleToGrid was not finite or exceeded the fixed-point range "
                    "(|node momentum| >= total mass * 2^15 length units per time unit): the state has diverged");
    // (last: a diverging state also scatters particles over so many blocks that tables overflow)
    if (c.error & (ERR_CAPACITY | ERR_SLABS))
        return fail(MPM_ERR_CAPACITY,
                    "a table of the engine overflowed (home / active blocks, slabs of the work items -- the pool is "
                    "grown at synchronisation points: call mpm_sync more often while a cloth spreads out, or set "
                    "MPM_SLAB_CAPACITY --, slabs over one block, halo or migration buffers)");
    return 0;
} MPM_CATCH_ALL

// ---- the solver calls -----------------------------------------------------
static void launch_substep(mpm_engine* e, float dt, const GridColliders& gc, bool allow_gate, bool lean = false);

// Substeps that mpm_run_substeps enqueued without their re-sort launches and that found a re-sort pending
// did nothing (Ctl::skipped): run them now, each with the re-sort in front.  Called by every entry point
// before it looks at or changes the state.
// `fresh`: receives the control block if this call read it and enqueued nothing afterwards (the stream is idle and
// *fresh is the state); untouched otherwise.
static int flush_phases(mpm_engine* e);
static int settle_owed(mpm_engine* e, Ctl* fresh);
static int settle(mpm_engine* e, Ctl* fresh) {
    if (int rc = settle_owed(e, fresh)) return rc;
    // phase calls that were waiting for the rest of their substep (see PendingPhases) come after everything owed
    if (e->pend.n) {
        e->settle_read_ctl = false;   // (what settle_owed read is no longer the state)
        return flush_phases(e);
    }
    return 0;
}
static int settle_owed(mpm_engine* e, Ctl* fresh) {
    e->force_check = true;   // whatever comes next starts with the re-sort launches
    e->dp.gated = 0;
    e->quiet_left = 0.f;     // (the call that settles may change the state: the hint is only kept from a settle that
                             // has just read it, below, to the substeps enqueued right after)
    if (!e->maybe_owed) return 0;
    // (a few rounds: a substep that is run again may itself overflow the slab pool -- a cloth that keeps spreading)
    for (int round = 0; round < 8 && e->maybe_owed; ++round) {
        e->maybe_owed = false;
        Ctl c;
        // (synchronises the stream.  Into pinned memory: a copy into pageable memory is staged by the runtime and
        // costs ~10 us more, which a caller that synchronises after every frame of substeps pays every time)
        if (!e->h_ctl) HIP_TRY(hipHostMalloc((void**)&e->h_ctl, sizeof(Ctl), hipHostMallocDefault));
        D2H(e, e->h_ctl, e->dp.ctl, sizeof(Ctl));
        c = *e->h_ctl;
        // every substep since the overflowing re-sort has skipped itself: the pool can be grown and the re-sort repeated
        if (int rc = recover_slab_overflow(e, c)) return rc;
        const unsigned owed = c.skipped;
        // what is left of the quiet time the last re-sort estimated (Ctl::quiet_time): substeps enqueued from here
        // on go without check launches while it lasts (launch_substep).  Nothing left when a re-sort is pending.
        e->quiet_left = owed || c.need_rebuild || c.error ? 0.f : std::max(0.f, c.quiet_time - c.time_since_resort);
        if (e->dp.dbg & 32)
            std::fprintf(stderr, "[mpm_hip] settle: quiet time %.4g s, %.4g s since the re-sort, %u owed, re-sort pending %d, checks launched %llu\n",
                         c.quiet_time, c.time_since_resort, owed, c.need_rebuild, (unsigned long long)e->checks_launched);
        if (!owed) {
            if (fresh) {
                *fresh = c;
                e->settle_read_ctl = true;
            }
            break;
        }
        const unsigned zero = 0;
        H2D(e, &e->dp.ctl->skipped, &zero, sizeof(unsigned));
        GridColliders gc;
        if (int rc = grid_colliders_for(e, e->owed_bc, &gc)) return rc;
        may_resort(e, e->owed_dt);
        for (unsigned k = 0; k < owed; ++k) launch_substep(e, e->owed_dt, gc, false);
        e->dp.gated = 0;
    }
    e->force_check = true;
    return 0;
}

#define READY_NO_SETTLE(e)                              \
    REQUIRE(e, "null handle");                          \
    REQUIRE((e)->finalized, "call mpm_finalize first"); \
    if (int rc__ = use(e)) return rc__
#define READY(e)         \
    READY_NO_SETTLE(e);  \
    if (int rc2__ = settle(e)) return rc2__

int mpm_halo_zone_blocks(mpm_handle_t e, int bx_lo, int bx_hi, uint32_t* count_out) try {
    READY(e);
    REQUIRE(count_out, "null output");
    HIP_TRY(hipStreamSynchronize(e->stream));
    Ctl c;
    D2H(e, &c, e->dp.ctl, sizeof(Ctl));
    std::vector<uint32_t> blocks(c.n_active);
    if (c.n_active) D2H(e, blocks.data(), e->dp.act_block, (size_t)c.n_active * 4);
    auto compact3 = [](uint32_t v) {   // (host copy of mpm_math.h's: every third bit)
        v &= 0x09249249u;
        v = (v ^ (v >> 2)) & 0x030C30C3u;
        v = (v ^ (v >> 4)) & 0x0300F00Fu;
        v = (v ^ (v >> 8)) & 0xFF0000FFu;
        v = (v ^ (v >> 16)) & 0x000003FFu;
        return v;
    };
    uint32_t n = 0;
    for (uint32_t b : blocks) {
        const int bx = (int)compact3(b >> 2);
        n += bx >= bx_lo && bx <= bx_hi;
    }
    *count_out = n;
    return 0;
} MPM_CATCH_ALL

int mpm_contact_frame(const float u[3], float J[9]) try {
    REQUIRE(u && J, "null argument");
    frame_from_normal(u, J);
    return 0;
} MPM_CATCH_ALL

int mpm_profile_contact_iteration(mpm_handle_t e, int reps, float kernel_ms[4]) try {
    READY(e);
    REQUIRE(kernel_ms && reps > 0 && reps <= 1000, "bad arguments");
    return profile_contact_iteration(e, reps, kernel_ms);
} MPM_CATCH_ALL

int mpm_memcpy_d2h(mpm_handle_t e, void* dst_host, const void* src_device, size_t bytes) try {
    REQUIRE(e, "null handle");
    REQUIRE(bytes == 0 || (dst_host && src_device), "null pointer");
    if (int rc = use(e)) return rc;
    if (bytes) D2H(e, dst_host, src_device, bytes);
    return 0;
} MPM_CATCH_ALL

int mpm_memcpy_h2d(mpm_handle_t e, void* dst_device, const void* src_host, size_t bytes) try {
    REQUIRE(e, "null handle");
    REQUIRE(bytes == 0 || (dst_device && src_host), "null pointer");
    if (int rc = use(e)) return rc;
    if (bytes) H2D(e, dst_device, src_host, bytes);
    return 0;
} MPM_CATCH_ALL

int mpm_debug_owed_substeps(mpm_handle_t e, uint32_t* out) try {
    READY_NO_SETTLE(e);
    REQUIRE(out, "null output");
    unsigned owed = 0;
    D2H(e, &owed, &e->dp.ctl->skipped, sizeof(unsigned));
    *out = owed;
    return 0;
} MPM_CATCH_ALL

int mpm_device_synchronize(void) try {
    // The reference's GpuSync() is a plain cudaDeviceSynchronize (cuda_mpm_solver.cu:164-166): it completes the work
    // of the device and says nothing about the state of any simulation.  Here "the work" includes substeps that
    // mpm_run_substeps deferred, so every engine of the device is settled and flushed first; the sticky SIMULATION
    // errors of an engine (diverged, capacity, ...) stay with that engine -- its own mpm_sync / mpm_get_stats report
    // them -- and only a failure of the runtime itself is an error of this call.  The list is walked under the lock
    // that mpm_create / mpm_destroy take: an engine cannot be destroyed by another thread while it is being settled.
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int first = 0;
    std::string first_msg;
    //  the engines of this device are PINNED under the lock and settled outside it: mpm_create / mpm_destroy of
    // other threads do not stall behind a GPU drain, and mpm_destroy of a pinned engine waits for the pin to go
    std::vector<mpm_engine*> mine;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        for (mpm_engine* e : g_live)
            if (e->device == dev) {
                e->pins.fetch_add(1);
                mine.push_back(e);
            }
    }
    for (mpm_engine* e : mine) {
        const int rc = mpm_sync(e);
        if ((rc == MPM_ERR_HIP || rc == MPM_ERR_NOMEM || rc == MPM_ERR_INTERNAL) && !first) {
            first = rc;
            first_msg = g_last_error;
        }
        e->pins.fetch_sub(1);
    }
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipDeviceSynchronize());
    return first ? fail(first, first_msg) : 0;
} MPM_CATCH_ALL

// The reference's five calls per substep (cuda_mpm_test.cc:64-72, deformable_driver.h:244-258), taken one by one, are
// nine launches with the re-sort check in front of every substep and the vertex forces as a kernel of their own.
// When they arrive in the standard order -- RebuildMapping(false), CalcFemStateAndForce(dt), ParticleToGrid(dt),
// UpdateGrid(bc), GridToParticle(dt) -- nothing is launched until the last one, and the substep then goes through the
// same path as mpm_run_substeps(1): check launches only when a re-sort can be due, vertex forces inside ParticleToGrid,
// deferral if a re-sort is pending.  Any other call in between (a download, the contact calls, a different order, a
// different dt) first launches what was held back, call by call as before, so every intermediate state a caller can
// look at is the one the reference would show.  MPM_DEFER_PHASES=0 turns this off.
static bool can_defer(const mpm_engine* e) { return e->defer_phases && e->finalized && !e->dp.dist.on; }
static int flush_phases(mpm_engine* e) {
    const int n = e->pend.n;
    e->pend.n = 0;
    if (n >= 1) {
        may_resort(e, 0.f);
        launch_rebuild(e);
    }
    if (n == 2) launch_fem(e, e->pend.dt);
    if (n >= 3) {
        // (nobody looked at the forces between CalcFemStateAndForce and ParticleToGrid: the vertex forces are
        // gathered inside k_p2g, which also writes them out -- one launch less, the same numbers)
        launch_fem_p2g(e, e->pend.dt);
        e->grid_state = 1;
    }
    if (n >= 4) {
        GridColliders gc;
        if (int rc = grid_colliders_for(e, e->pend.bc, &gc)) return rc;
        launch_grid(e, gc);
        e->grid_state = 2;
    }
    return 0;
}

int mpm_rebuild_mapping(mpm_handle_t e, int sort) try {
    if (e && can_defer(e) && !sort && e->pend.n == 0) {
        READY_NO_SETTLE(e);
        e->pend.n = 1;
        return 0;
    }
    READY(e);
    may_resort(e, 0.f);
    launch_rebuild(e);
    if (sort) {
        REQUIRE(!e->dp.dist.on, "RebuildMapping(sort = true) is not available on a partitioned domain");
        return api_sort(e);
    }
    return 0;
} MPM_CATCH_ALL

int mpm_calc_fem_state_and_force(mpm_handle_t e, float dt) try {
    if (e && can_defer(e) && e->pend.n == 1) {
        READY_NO_SETTLE(e);
        e->pend.n = 2;
        e->pend.dt = dt;
        return 0;
    }
    READY(e);
    launch_fem(e, dt);
    return 0;
} MPM_CATCH_ALL

int mpm_particle_to_grid(mpm_handle_t e, float dt) try {
    if (e && can_defer(e) && e->pend.n == 2 && dt == e->pend.dt) {
        READY_NO_SETTLE(e);
        e->pend.n = 3;
        return 0;
    }
    READY(e);
    launch_p2g(e, dt);
    e->grid_state = 1;
    return 0;
} MPM_CATCH_ALL

int mpm_update_grid(mpm_handle_t e, int bc) try {
    if (e && can_defer(e) && e->pend.n == 3) {
        READY_NO_SETTLE(e);
        GridColliders probe;
        if (int rc = grid_colliders_for(e, bc, &probe)) return rc;   // (a bad mpm_bc is reported by this call)
        e->pend.n = 4;
        e->pend.bc = bc;
        return 0;
    }
    READY(e);
    REQUIRE(e->grid_state >= 1, "UpdateGrid before ParticleToGrid");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    launch_grid(e, gc);
    e->grid_state = 2;
    return 0;
} MPM_CATCH_ALL

int mpm_grid_gather(mpm_handle_t e) try {
    READY(e);
    REQUIRE(e->grid_state >= 1, "grid gather before ParticleToGrid");
    hipLaunchKernelGGL(k_grid<0>, dim3(e->g_grid), dim3(256), 0, e->stream, e->dp, GridColliders{});
    e->grid_state = 3;  // raw sums in gv
    return 0;
} MPM_CATCH_ALL

size_t mpm_halo_buffer_bytes(size_t cap) { return (((4 + cap) * 4 + 15) / 16) * 16 + cap * 64 * 16; }

int mpm_halo_pack(mpm_handle_t e, int bx_lo, int bx_hi, int shift_bx, void* dev_buf, size_t cap) try {
    READY(e);
    REQUIRE(e->grid_state == 3, "halo pack needs mpm_grid_gather first");
    REQUIRE(dev_buf && cap > 0 && cap < (1u << 24), "bad halo buffer");
    HIP_TRY(hipMemsetAsync(dev_buf, 0, 16, e->stream));
    HaloZones z{};
    z.lo[0] = bx_lo; z.hi[0] = bx_hi; z.shift[0] = shift_bx;
    z.buf[0] = static_cast<uint32_t*>(dev_buf);
    hipLaunchKernelGGL(k_halo_pack2, dim3(e->g_grid, 1), dim3(256), 0, e->stream, e->dp, z, (unsigned)cap);
    return 0;
} MPM_CATCH_ALL

int mpm_halo_add(mpm_handle_t e, const void* dev_buf, size_t cap) try {
    READY(e);
    REQUIRE(e->grid_state == 3, "halo add needs mpm_grid_gather first");
    REQUIRE(dev_buf && cap > 0, "bad halo buffer");
    HaloBufs hb{};
    hb.buf[0] = static_cast<const uint32_t*>(dev_buf);
    hipLaunchKernelGGL(k_halo_add2, dim3(64, 1), dim3(256), 0, e->stream, e->dp, hb, (unsigned)cap);
    return 0;
} MPM_CATCH_ALL

int mpm_update_grid_from_sums(mpm_handle_t e, int bc) try {
    READY(e);
    REQUIRE(e->grid_state == 3, "needs mpm_grid_gather first");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    hipLaunchKernelGGL(k_grid<2>, dim3(e->g_grid), dim3(256), 0, e->stream, e->dp, gc);
    e->grid_state = 2;
    return 0;
} MPM_CATCH_ALL

int mpm_substep_begin(mpm_handle_t e, float dt) try {
    READY(e);
    may_resort(e, dt);
    launch_rebuild(e);
    launch_fem_p2g(e, dt);
    hipLaunchKernelGGL(k_grid<0>, dim3(e->g_grid), dim3(256), 0, e->stream, e->dp, GridColliders{});
    e->grid_state = 3;
    return 0;
} MPM_CATCH_ALL

int mpm_substep_end(mpm_handle_t e, float dt, int bc) try {
    READY(e);
    REQUIRE(e->grid_state == 3, "mpm_substep_end without mpm_substep_begin");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    hipLaunchKernelGGL(k_grid<2>, dim3(e->g_grid), dim3(256), 0, e->stream, e->dp, gc);
    e->grid_state = 2;
    launch_g2p(e, dt);
    e->substeps += 1;
    return 0;
} MPM_CATCH_ALL

// Replays `body` (a sequence of launches on e->stream) from a graph cached under `key`.
static int replay_keyed(mpm_engine* e, mpm_engine::KeyedGraph& kg, const std::vector<uint64_t>& key,
                        const std::function<void()>& body) {
    if (!kg.exec || kg.key != key) {
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
        kg.exec = nullptr;
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        body();
        HIP_TRY(hipStreamEndCapture(e->stream, &g));
        const hipError_t err = hipGraphInstantiate(&kg.exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        HIP_TRY(err);
        kg.key = key;
    }
    HIP_TRY(hipGraphLaunch(kg.exec, e->stream));
    return 0;
}
static uint64_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

static int substep_begin_halo(mpm_handle_t e, float dt, int n, const int* bx_lo, const int* bx_hi, const int* shift_bx,
                              void* const* send_bufs, size_t cap, uint32_t* const* counters);
int mpm_substep_begin_halo(mpm_handle_t e, float dt, int n, const int* bx_lo, const int* bx_hi, const int* shift_bx,
                           void* const* send_bufs, size_t cap) try {
    return substep_begin_halo(e, dt, n, bx_lo, bx_hi, shift_bx, send_bufs, cap, nullptr);
} MPM_CATCH_ALL
// counters: per zone the word the pack counts its entries in (null: word 0 of the send buffer itself)
static int substep_begin_halo(mpm_handle_t e, float dt, int n, const int* bx_lo, const int* bx_hi, const int* shift_bx,
                              void* const* send_bufs, size_t cap, uint32_t* const* counters) {
    READY(e);
    may_resort(e, dt);
    REQUIRE(n >= 0 && n <= 2 && (n == 0 || (bx_lo && bx_hi && shift_bx && send_bufs)), "bad halo zone list");
    REQUIRE(n == 0 || (cap > 0 && cap < (1u << 24)), "bad halo buffer");
    DP p = e->dp;   // per-launch copy: k_grid<0> packs the zones' sums into the send buffers as it gathers them
    DP pp = e->dp;  // ... and k_p2g, in front of it, resets their entry counters
    p.halo_pn = n;
    p.halo_pcap = (unsigned)cap;
    // (mpm_chain_substeps: another substep of the same batch follows, see DP::lean_g2p)
    const int lean = e->chain_lean && !e->dp.dist.on;
    std::vector<uint64_t> key = {1, bits_of(dt), (uint64_t)n, (uint64_t)cap, (uint64_t)(uintptr_t)e->stream, (uint64_t)lean};
    for (int i = 0; i < n; ++i) {
        REQUIRE(send_bufs[i], "null halo buffer");
        pp.halo_hdr[i] = counters && counters[i] ? counters[i] : static_cast<uint32_t*>(send_bufs[i]);
        p.halo_plo[i] = bx_lo[i]; p.halo_phi[i] = bx_hi[i]; p.halo_pshift[i] = shift_bx[i];
        p.halo_pbuf[i] = static_cast<uint32_t*>(send_bufs[i]);
        p.halo_pcnt[i] = counters ? counters[i] : nullptr;
        key.insert(key.end(), {(uint64_t)(uint32_t)bx_lo[i], (uint64_t)(uint32_t)bx_hi[i], (uint64_t)(uint32_t)shift_bx[i],
                               (uint64_t)(uintptr_t)send_bufs[i], (uint64_t)(uintptr_t)p.halo_pcnt[i]});
    }
    auto body = [&]() {
        e->dp.lean_resort = !e->dp.dist.on;   // (CalcFemStateAndForce follows at once)
        launch_rebuild(e);
        e->dp.lean_resort = 0;
        e->dp.lean_g2p = lean;
        for (int i = 0; i < 2; ++i) e->dp.halo_hdr[i] = pp.halo_hdr[i];
        launch_fem_p2g(e, dt);
        for (int i = 0; i < 2; ++i) e->dp.halo_hdr[i] = nullptr;
        e->dp.lean_g2p = 0;
        hipLaunchKernelGGL(k_grid<0>, dim3(e->g_grid), dim3(256), 0, e->stream, p, GridColliders{});
    };
    if (e->use_halo_graphs) {
        if (int rc = replay_keyed(e, e->halo_graph[2 * e->halo_graph_parity], key, body)) return rc;
    } else {
        body();
    }
    e->grid_state = 3;
    e->halo_mid_done = false;
    e->halo_nz = n;
    for (int i = 0; i < n; ++i) { e->halo_zlo[i] = bx_lo[i]; e->halo_zhi[i] = bx_hi[i]; }
    return 0;
}

// Between begin and end: the part of the grid update and of G2P that does not depend on the
// neighbours' sums, to be overlapped with the exchange.  Zones = the ranges given to begin.
int mpm_substep_mid_halo(mpm_handle_t e, float dt, int bc) try {
    READY(e);
    REQUIRE(e->grid_state == 3 && !e->halo_mid_done, "mpm_substep_mid_halo needs mpm_substep_begin_halo first");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    DP p = e->dp;
    p.halo_cls = 0;
    p.halo_nz = e->halo_nz;
    for (int i = 0; i < e->halo_nz; ++i) { p.halo_zlo[i] = e->halo_zlo[i]; p.halo_zhi[i] = e->halo_zhi[i]; }
    hipLaunchKernelGGL(k_grid<2>, dim3(e->g_grid), dim3(256), 0, e->stream, p, gc);
    launch_g2p_with(e, p, dt);
    e->halo_mid_done = true;
    return 0;
} MPM_CATCH_ALL

static int substep_end_halo(mpm_handle_t e, float dt, int bc, int n, const void* const* recv_bufs, size_t cap, bool with_g2p);
int mpm_substep_end_halo(mpm_handle_t e, float dt, int bc, int n, const void* const* recv_bufs, size_t cap) try {
    return substep_end_halo(e, dt, bc, n, recv_bufs, cap, true);
} MPM_CATCH_ALL
// with_g2p = false: the grid update only -- a coupled substep puts the contact solve between it and GridToParticle
static int substep_end_halo(mpm_handle_t e, float dt, int bc, int n, const void* const* recv_bufs, size_t cap, bool with_g2p) {
    READY(e);
    REQUIRE(n >= 0 && n <= 2 && (n == 0 || recv_bufs), "bad halo buffer list");
    REQUIRE(e->grid_state == 3, "mpm_substep_end_halo without mpm_substep_begin_halo");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    HaloBufs b{};
    const int lean = e->chain_lean && !e->dp.dist.on;
    std::vector<uint64_t> key = {2, bits_of(dt), (uint64_t)(uint32_t)bc, (uint64_t)n, (uint64_t)cap,
                                 (uint64_t)(uintptr_t)e->stream, e->grid_colliders_version, (uint64_t)lean, (uint64_t)with_g2p};
    for (int i = 0; i < n; ++i) {
        REQUIRE(recv_bufs[i], "null halo buffer");
        b.buf[i] = static_cast<const uint32_t*>(recv_bufs[i]);
        key.push_back((uint64_t)(uintptr_t)recv_bufs[i]);
    }
    const bool split = e->halo_mid_done;
    DP p = e->dp;
    p.lean_g2p = lean;
    if (split) {   // the interior is done: only what the received sums touch is left
        p.halo_cls = 1;
        p.halo_nz = e->halo_nz;
        for (int i = 0; i < e->halo_nz; ++i) { p.halo_zlo[i] = e->halo_zlo[i]; p.halo_zhi[i] = e->halo_zhi[i]; }
    }
    // The received sums are added inside the grid update (k_grid<2> looks its zone blocks up in the buffers): one launch
    // less per substep than add + update.  Needs the zones of the matching mpm_substep_begin_halo, buffer i <-> zone i.
    DP pg = p;
    const bool folded = n > 0 && n == e->halo_nz;
    if (folded) {
        pg.halo_pn = n;
        pg.halo_pcap = (unsigned)cap;
        for (int i = 0; i < n; ++i) {
            pg.halo_plo[i] = e->halo_zlo[i]; pg.halo_phi[i] = e->halo_zhi[i];
            pg.halo_pbuf[i] = const_cast<uint32_t*>(b.buf[i]);
        }
    }
    auto body = [&]() {
        if (n > 0 && !folded) hipLaunchKernelGGL(k_halo_add2, dim3(64, n), dim3(256), 0, e->stream, p, b, (unsigned)cap);
        hipLaunchKernelGGL(k_grid<2>, dim3(e->g_grid), dim3(256), 0, e->stream, pg, gc);
        if (with_g2p) launch_g2p_with(e, p, dt);
    };
    if (e->use_halo_graphs && !split) {
        if (int rc = replay_keyed(e, e->halo_graph[2 * e->halo_graph_parity + 1], key, body)) return rc;
    } else {
        body();
    }
    e->halo_mid_done = false;
    e->grid_state = 2;
    if (with_g2p) e->substeps += 1;
    return 0;
}

int mpm_chain_unique_id(char id_out[128]) try {
    REQUIRE(id_out, "null argument");
    const rccl_rt::Api* a = rccl_rt::api();
    if (!a) return fail(MPM_ERR_HIP, "RCCL (librccl.so) is not available in this process");
    rccl_rt::UniqueId id;
    RCCL_TRY(a->get_unique_id(&id));
    std::memcpy(id_out, id.internal, rccl_rt::kIdBytes);
    return 0;
} MPM_CATCH_ALL

int mpm_chain_destroy(mpm_handle_t e) try {
    REQUIRE(e, "null handle");
    mpm_engine::Chain& c = e->chain;
    if (int rc = use(e)) return rc;
    (void)hipStreamSynchronize(e->stream);
    if (c.comm) {   // (RCCL is only looked up by engines that use it)
        const rccl_rt::Api* a = rccl_rt::api();
        if (a) (void)a->comm_destroy(c.comm);
    }
    for (void* q : {c.send_l, c.send_r, c.recv_l, c.recv_r, c.mig_send_l, c.mig_send_r, c.mig_recv_l, c.mig_recv_r,
                    (void*)c.mig_quiet_all})
        if (q) (void)hipFree(q);
    for (int k = 0; k < 2; ++k)
        if (c.peer_mapped[k] && c.peer_base[k] && !(k == 1 && c.peer_base[1] == c.peer_base[0])) (void)hipIpcCloseMemHandle(c.peer_base[k]);
    if (c.direct_base) (void)hipFree(c.direct_base);
    if (c.direct_cnt) (void)hipFree(c.direct_cnt);
    c = mpm_engine::Chain();
    {
        mpm_engine::Team& t = e->team;
        for (int r = 0; r < TEAM_MAX; ++r)
            if (t.mapped[r] && t.peer[r]) (void)hipIpcCloseMemHandle(t.peer[r]);
        if (t.base) (void)hipFree(t.base);
        if (t.ts) (void)hipFree(t.ts);
        t = mpm_engine::Team();
    }
    return 0;
} MPM_CATCH_ALL

int mpm_chain_enable_migration(mpm_handle_t e, int every, size_t capacity_particles) try {
    READY(e);
    mpm_engine::Chain& c = e->chain;
    REQUIRE(c.comm, "mpm_chain_init first");
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    REQUIRE(c.pitch == 0 && c.rank == e->dp.dist.rank && c.world == e->dp.dist.world,
            "the chain of a partitioned domain has pitch 0 and the rank / world given to mpm_dist_init");
    REQUIRE(every >= 0 && capacity_particles > 0 && capacity_particles < (1u << 28), "bad migration parameters");
    REQUIRE(every > 0 || rccl_rt::api()->all_reduce, "adaptive migration (every = 0) needs ncclAllReduce");
    for (void** q : {&c.mig_send_l, &c.mig_send_r, &c.mig_recv_l, &c.mig_recv_r, (void**)&c.mig_quiet_all}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    HIP_TRY(hipMalloc((void**)&c.mig_quiet_all, 16));
    c.mig_budget = 0.f;    // (the first substep starts with a migration: it yields the first estimate)
    c.mig_elapsed = 0.f;
    c.mig_every = every;
    c.mig_cap = capacity_particles;
    c.mig_bytes = mpm_dist_migration_buffer_bytes(capacity_particles);
    for (void** q : {&c.mig_send_l, &c.mig_send_r, &c.mig_recv_l, &c.mig_recv_r}) {
        HIP_TRY(hipMalloc(q, c.mig_bytes));
        HIP_TRY(hipMemsetAsync(*q, 0, c.mig_bytes, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
} MPM_CATCH_ALL

int mpm_chain_init(mpm_handle_t e, const char id[128], int rank, int world, int cut_lo_block, int cut_hi_block,
                   int pitch_blocks, int zone_blocks, size_t capacity_blocks, int periodic) try {
    READY(e);
    REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    REQUIRE(zone_blocks >= 1 && capacity_blocks > 0 && capacity_blocks < (1u << 24), "bad halo geometry");
    // (id == NULL: the geometry only, no RCCL communicator -- for the direct transport, mpm_chain_direct_prepare)
    const rccl_rt::Api* a = id ? rccl_rt::api() : nullptr;
    if (id && !a) return fail(MPM_ERR_HIP, "RCCL (librccl.so) is not available in this process");
    if (int rc = mpm_chain_destroy(e)) return rc;
    mpm_engine::Chain& c = e->chain;
    c.rank = rank; c.world = world; c.pitch = pitch_blocks; c.cap = capacity_blocks;
    c.left = rank > 0 ? rank - 1 : (periodic ? world - 1 : -1);
    c.right = rank < world - 1 ? rank + 1 : (periodic ? 0 : -1);
    c.zone_lo[0] = cut_lo_block - zone_blocks; c.zone_hi[0] = cut_lo_block + zone_blocks - 1;
    c.zone_lo[1] = cut_hi_block - zone_blocks; c.zone_hi[1] = cut_hi_block + zone_blocks - 1;
    c.bytes = mpm_halo_buffer_bytes(capacity_blocks);
    for (void** q : {&c.send_l, &c.send_r, &c.recv_l, &c.recv_r}) {
        HIP_TRY(hipMalloc(q, c.bytes));
        HIP_TRY(hipMemsetAsync(*q, 0, c.bytes, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (id) {
        rccl_rt::UniqueId uid;
        std::memcpy(uid.internal, id, rccl_rt::kIdBytes);
        RCCL_TRY(a->comm_init_rank(&c.comm, world, uid, rank));
    }
    return 0;
} MPM_CATCH_ALL

// ---- direct halo: peer-to-peer stores + sequence flags -------------------------------------------
static size_t direct_slot_bytes(const mpm_engine::Chain& c) { return (c.bytes + 255) & ~(size_t)255; }
static size_t direct_total_bytes(const mpm_engine::Chain& c) { return 4 * direct_slot_bytes(c) + 256; }
// [side 0 = from the left neighbour, 1 = from the right][parity] inside an allocation laid out as above
static void* direct_buffer(const mpm_engine::Chain& c, void* base, int side, int parity) {
    return (char*)base + (size_t)(side * 2 + parity) * direct_slot_bytes(c);
}
static uint32_t* direct_flag(const mpm_engine::Chain& c, void* base, int side) {
    return reinterpret_cast<uint32_t*>((char*)base + 4 * direct_slot_bytes(c)) + side * 16;   // (64 bytes apart)
}

int mpm_chain_direct_prepare(mpm_handle_t e, char handle_out[64]) try {
    READY(e);
    mpm_engine::Chain& c = e->chain;
    REQUIRE(handle_out, "null argument");
    REQUIRE(c.cap > 0, "mpm_chain_init first (with a NULL id for the geometry alone)");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!c.direct_base) {
        const size_t bytes = direct_total_bytes(c);
        // FINE-GRAINED, or not at all: the protocol's consumer (k_grid<2>) reads the count, the ids and the sums with plain
        // loads and relies on a peer's stores not being shadowed by stale lines of this device's L2.  A coarse-grained
        // region would add stale halo sums without any error flag, so a refused allocation is an error of
        // this call -- the caller stays on RCCL (the connect protocol allows that: mpm_chain_direct_connect(NULL, NULL)).
        // MPM_DIRECT_COARSE_OK=1 accepts plain device memory for rehearsals on ONE device, where both ends share the L2.
        const hipError_t fg = hipExtMallocWithFlags(&c.direct_base, bytes, hipDeviceMallocFinegrained);
        if (fg != hipSuccess) {
            (void)hipGetLastError();
            c.direct_base = nullptr;
            if (!getenv("MPM_DIRECT_COARSE_OK"))
                return fail(MPM_ERR_HIP, std::string("mpm_chain_direct_prepare: fine-grained device memory is not available (") +
                                             hipGetErrorString(fg) + "): the direct halo exchange needs it; stay on the RCCL transport");
            HIP_TRY(hipMalloc(&c.direct_base, bytes));
            c.direct_coarse = true;
        }
        HIP_TRY(hipMemsetAsync(c.direct_base, 0, bytes, e->stream));
        if (!c.direct_cnt) {
            HIP_TRY(hipMalloc((void**)&c.direct_cnt, 64));
            HIP_TRY(hipMemsetAsync(c.direct_cnt, 0, 64, e->stream));
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, c.direct_base));
    std::memcpy(handle_out, &h, 64);
    if (const char* t = getenv("MPM_HALO_TIMEOUT_S")) c.direct_timeout_s = std::max(.001f, (float)atof(t));
    c.direct_mute = getenv("MPM_HALO_DEBUG_MUTE") != nullptr;   // (tests: this rank never signals)
    return 0;
} MPM_CATCH_ALL

int mpm_chain_direct_connect(mpm_handle_t e, const char left_handle[64], const char right_handle[64]) try {
    READY(e);
    mpm_engine::Chain& c = e->chain;
    REQUIRE(c.direct_base, "mpm_chain_direct_prepare first");
    const char* hs[2] = {left_handle, right_handle};
    const int nbr[2] = {c.left, c.right};
    for (int k = 0; k < 2; ++k) {
        if (c.peer_mapped[k] && c.peer_base[k] && !(k == 1 && c.peer_base[1] == c.peer_base[0])) (void)hipIpcCloseMemHandle(c.peer_base[k]);
        c.peer_base[k] = nullptr;
        c.peer_mapped[k] = false;
    }
    const bool foreign = (c.left >= 0 && c.left != c.rank) || (c.right >= 0 && c.right != c.rank);
    if (foreign && !left_handle && !right_handle) {   // switched off again: the chain is RCCL's (or nobody's) once more
        c.direct = false;
        return 0;
    }
    for (int k = 0; k < 2; ++k) {
        if (nbr[k] < 0) continue;
        if (nbr[k] == c.rank) {   // a ring of one: the neighbour is this rank, no mapping
            c.peer_base[k] = c.direct_base;
            continue;
        }
        REQUIRE(hs[k], "missing IPC handle of a neighbour");
        if (k == 1 && c.left == c.right && c.peer_base[0]) {   // a ring of two: both neighbours are the same rank
            c.peer_base[1] = c.peer_base[0];
            c.peer_mapped[1] = c.peer_mapped[0];
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, hs[k], 64);
        const hipError_t err = hipIpcOpenMemHandle(&c.peer_base[k], h, hipIpcMemLazyEnablePeerAccess);
        if (err != hipSuccess) {
            (void)hipGetLastError();   // (not sticky: the engine stays usable, with RCCL or without neighbours)
            c.peer_base[k] = nullptr;
            return fail(MPM_ERR_HIP, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(err) +
                                         " (the neighbour's buffers cannot be mapped: another node, or a handle of this very process)");
        }
        c.peer_mapped[k] = true;
    }
    c.direct = true;
    return 0;
} MPM_CATCH_ALL

// Neighbours that live in THIS process (an in-process world: several engines of one partition on one stream, for tests and
// rehearsals of more ranks than a box admits processes): their regions are plain pointers, a HIP IPC handle of one's own
// process cannot be opened.  mpm_chain_direct_base hands a rank's region out, mpm_chain_direct_connect_local takes the
// neighbours'.
int mpm_chain_direct_base(mpm_handle_t e, void** base_out) try {
    READY(e);
    REQUIRE(base_out, "null argument");
    REQUIRE(e->chain.direct_base, "mpm_chain_direct_prepare first");
    *base_out = e->chain.direct_base;
    return 0;
} MPM_CATCH_ALL

int mpm_chain_direct_connect_local(mpm_handle_t e, void* left_base, void* right_base) try {
    READY(e);
    mpm_engine::Chain& c = e->chain;
    REQUIRE(c.direct_base, "mpm_chain_direct_prepare first");
    REQUIRE((c.left < 0 || left_base) && (c.right < 0 || right_base), "missing region of a neighbour");
    for (int k = 0; k < 2; ++k) {
        if (c.peer_mapped[k] && c.peer_base[k] && !(k == 1 && c.peer_base[1] == c.peer_base[0])) (void)hipIpcCloseMemHandle(c.peer_base[k]);
        c.peer_mapped[k] = false;
    }
    c.peer_base[0] = c.left >= 0 ? left_base : nullptr;
    c.peer_base[1] = c.right >= 0 ? right_base : nullptr;
    c.direct = true;
    return 0;
} MPM_CATCH_ALL

// ---- TEAM transport of the distributed contact solve (mpm_team.h) --------------------------------------------------------
int mpm_team_prepare(mpm_handle_t e, size_t zone_capacity_blocks, char handle_out[64], void** base_out) try {
    READY(e);
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    REQUIRE(e->dp.dist.world <= TEAM_MAX, "the team transport serves the ranks of one node (at most 8)");
    REQUIRE(zone_capacity_blocks > 0 && zone_capacity_blocks < (1u << 20), "bad zone capacity");
    mpm_engine::Team& t = e->team;
    REQUIRE(!t.on, "mpm_team_prepare after mpm_team_connect");
    if (!t.base) {
        t.rank = e->dp.dist.rank;
        t.world = e->dp.dist.world;
        t.zone_cap = zone_capacity_blocks;
        t.zone_bytes = team_zone_slot_bytes(zone_capacity_blocks);
        const size_t bytes = team_region_bytes(t.zone_bytes);
        // fine-grained or not at all, as for the direct halo (mpm_chain_direct_prepare)
        const hipError_t fg = hipExtMallocWithFlags(&t.base, bytes, hipDeviceMallocFinegrained);
        if (fg != hipSuccess) {
            (void)hipGetLastError();
            t.base = nullptr;
            if (!getenv("MPM_DIRECT_COARSE_OK"))
                return fail(MPM_ERR_HIP, std::string("mpm_team_prepare: fine-grained device memory is not available (") + hipGetErrorString(fg) +
                                             "): the team transport needs it; the solve stays on the host-driven transports");
            HIP_TRY(hipMalloc(&t.base, bytes));
            t.coarse = true;
        }
        HIP_TRY(hipMemsetAsync(t.base, 0, bytes, e->stream));
        HIP_TRY(hipMalloc((void**)&t.ts, sizeof(TeamState)));
        HIP_TRY(hipMemsetAsync(t.ts, 0, sizeof(TeamState), e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (const char* s2 = getenv("MPM_HALO_TIMEOUT_S")) t.timeout_s = std::max(.001f, (float)atof(s2));
    }
    if (handle_out) {
        hipIpcMemHandle_t h;
        HIP_TRY(hipIpcGetMemHandle(&h, t.base));
        std::memcpy(handle_out, &h, 64);
    }
    if (base_out) *base_out = t.base;
    return 0;
} MPM_CATCH_ALL

// handles: world x 64 bytes (the IPC handles of all ranks, own slot ignored) or NULL; local_bases: world pointers or NULL --
// a rank of this process is named by its region's pointer (mpm_team_prepare's base_out), every other one by its handle
int mpm_team_connect(mpm_handle_t e, const char* handles, void* const* local_bases) try {
    READY(e);
    mpm_engine::Team& t = e->team;
    REQUIRE(t.base, "mpm_team_prepare first");
    for (int r = 0; r < t.world; ++r) {
        if (r == t.rank) {
            t.peer[r] = t.base;
            continue;
        }
        if (t.mapped[r] && t.peer[r]) (void)hipIpcCloseMemHandle(t.peer[r]);
        t.mapped[r] = false;
        t.peer[r] = nullptr;
        if (local_bases && local_bases[r]) {
            t.peer[r] = local_bases[r];
            continue;
        }
        REQUIRE(handles, "missing IPC handle of a rank");
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        const hipError_t err = hipIpcOpenMemHandle(&t.peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            t.peer[r] = nullptr;
            return fail(MPM_ERR_HIP, std::string("mpm_team_connect: hipIpcOpenMemHandle: ") + hipGetErrorString(err) +
                                         " (a rank's region cannot be mapped: another node, or a handle of this very process)");
        }
        t.mapped[r] = true;
    }
    t.on = true;
    return 0;
} MPM_CATCH_ALL

// The two halves of a chain substep over the DIRECT transport (see mpm_chain_substeps): begin = re-sort checks, FEM,
// ParticleToGrid, raw sums packed into the neighbours' slots of this substep's parity, signal; end = wait, add + grid
// update (+ GridToParticle).  Between them a coupled substep runs nothing that touches the grid.
static int chain_zones(const mpm_engine::Chain& c, int* lo, int* hi, int* sh) {
    int nz = 0;
    if (c.left >= 0) { lo[nz] = c.zone_lo[0]; hi[nz] = c.zone_hi[0]; sh[nz] = +c.pitch; ++nz; }
    if (c.right >= 0) { lo[nz] = c.zone_lo[1]; hi[nz] = c.zone_hi[1]; sh[nz] = -c.pitch; ++nz; }
    return nz;
}
static int chain_direct_begin(mpm_engine* e, float dt) {
    mpm_engine::Chain& c = e->chain;
    REQUIRE(c.direct, "mpm_chain_direct_connect first");
    int lo[2], hi[2], sh[2];
    const int nz = chain_zones(c, lo, hi, sh);
    c.mig_elapsed += dt;
    c.steps += 1;
    const int parity = (int)(c.steps & 1u);
    const uint32_t seq = (uint32_t)c.steps;
    void* dsb[2] = {nullptr, nullptr};
    uint32_t *sig[2] = {nullptr, nullptr}, *cnt[2] = {nullptr, nullptr}, *hdr[2] = {nullptr, nullptr}, *counters[2] = {nullptr, nullptr};
    int k = 0;
    if (c.left >= 0) {   // my left zone goes to the left neighbour's "from the right" buffer
        dsb[k] = direct_buffer(c, c.peer_base[0], 1, parity);
        sig[0] = direct_flag(c, c.peer_base[0], 1);
        cnt[0] = c.direct_cnt; hdr[0] = static_cast<uint32_t*>(dsb[k]);
        counters[k] = cnt[0];
        ++k;
    }
    if (c.right >= 0) {
        dsb[k] = direct_buffer(c, c.peer_base[1], 0, parity);
        sig[1] = direct_flag(c, c.peer_base[1], 0);
        cnt[1] = c.direct_cnt + 8; hdr[1] = static_cast<uint32_t*>(dsb[k]);
        counters[k] = cnt[1];
        ++k;
    }
    e->halo_graph_parity = parity;
    const int rc = substep_begin_halo(e, dt, nz, lo, hi, sh, dsb, c.cap, counters);
    e->halo_graph_parity = 0;
    if (rc) return rc;
    if (!c.direct_mute && nz > 0)
        hipLaunchKernelGGL(k_halo_signal, dim3(1), dim3(64), 0, e->stream, sig[0], sig[1], seq, (const uint32_t*)cnt[0], hdr[0],
                           (const uint32_t*)cnt[1], hdr[1], (unsigned)c.cap);
    return 0;
}
static int chain_direct_end(mpm_engine* e, float dt, int bc, bool with_g2p) {
    mpm_engine::Chain& c = e->chain;
    int lo[2], hi[2], sh[2];
    const int nz = chain_zones(c, lo, hi, sh);
    const int parity = (int)(c.steps & 1u);
    const uint32_t seq = (uint32_t)c.steps;
    void* drb[2] = {nullptr, nullptr};
    uint32_t* mine[2] = {nullptr, nullptr};
    int k = 0;
    if (c.left >= 0) { drb[k++] = direct_buffer(c, c.direct_base, 0, parity); mine[0] = direct_flag(c, c.direct_base, 0); }
    if (c.right >= 0) { drb[k++] = direct_buffer(c, c.direct_base, 1, parity); mine[1] = direct_flag(c, c.direct_base, 1); }
    if (nz > 0)
        hipLaunchKernelGGL(k_halo_wait, dim3(1), dim3(64), 0, e->stream, (const uint32_t*)mine[0], (const uint32_t*)mine[1], seq,
                           (unsigned long long)((double)c.direct_timeout_s * 1e8), e->dp.ctl);
    e->halo_graph_parity = parity;
    const int rc = substep_end_halo(e, dt, bc, nz, drb, c.cap, with_g2p);
    e->halo_graph_parity = 0;
    return rc;
}

int mpm_chain_substeps(mpm_handle_t e, int n, float dt, int bc) try {
    READY(e);
    mpm_engine::Chain& c = e->chain;
    REQUIRE(c.comm || c.direct, "mpm_chain_init first");
    const rccl_rt::Api* a = c.comm ? rccl_rt::api() : nullptr;
    REQUIRE(c.comm || c.mig_cap == 0, "migration needs the RCCL communicator (mpm_chain_init with an id)");
    struct LeanReset {   // (whatever way this function is left)
        mpm_engine* e;
        ~LeanReset() { e->chain_lean = 0; }
    } lean_reset{e};
    // zones / buffers in the order (left, right), leaving out a missing neighbour
    int lo[2], hi[2], sh[2], nz = 0;
    void *sb[2], *rb[2];
    if (c.left >= 0) { lo[nz] = c.zone_lo[0]; hi[nz] = c.zone_hi[0]; sh[nz] = +c.pitch; sb[nz] = c.send_l; rb[nz] = c.recv_l; ++nz; }
    if (c.right >= 0) { lo[nz] = c.zone_lo[1]; hi[nz] = c.zone_hi[1]; sh[nz] = -c.pitch; sb[nz] = c.send_r; rb[nz] = c.recv_r; ++nz; }
    for (int s = 0; s < n; ++s) {
        // Migration: every mig_every substeps, or (mig_every = 0, adaptive) when half of the time has passed in which,
        // by the ranks' common estimate, no particle can have drifted further than the bands allow (Dist::mig_delta).
        bool due = false;
        if (c.mig_cap > 0 && nz > 0) {
            if (c.mig_every > 0) due = c.steps > 0 && c.steps % (uint64_t)c.mig_every == 0;
            else due = !(c.mig_elapsed + dt <= c.mig_budget);
        }
        if (due) {
            // particles change hands (mpm_dist.h): records to / from both neighbours, same pairing as below.  In two
            // rounds: the 16-byte headers first (record counts), then exactly the records that exist -- a RCCL
            // send needs its size when it is enqueued, and the buffers are sized for the worst case (9 MB for 65536
            // records) while a migration of a cloth at rest moves none.  The host reads the counts in between; a
            // migration is a synchronisation point anyway (mpm_dist_migrate_apply).
            if (int rc = mpm_dist_migrate_pack(e, c.mig_send_l, c.mig_send_r, c.mig_cap)) return rc;
            auto exchange = [&](size_t out_l, size_t out_r, size_t in_l, size_t in_r, size_t offset) -> int {
                RCCL_TRY(a->group_start());
                int rc_g = 0;
                if (!rc_g && c.left >= 0 && out_l) rc_g = a->send((char*)c.mig_send_l + offset, out_l, 0, c.left, c.comm, e->stream);
                if (!rc_g && c.right >= 0 && out_r) rc_g = a->send((char*)c.mig_send_r + offset, out_r, 0, c.right, c.comm, e->stream);
                if (!rc_g && c.right >= 0 && in_r) rc_g = a->recv((char*)c.mig_recv_r + offset, in_r, 0, c.right, c.comm, e->stream);
                if (!rc_g && c.left >= 0 && in_l) rc_g = a->recv((char*)c.mig_recv_l + offset, in_l, 0, c.left, c.comm, e->stream);
                const int rc_e = a->group_end();   // (always closes the group, also after a failed call)
                RCCL_TRY(rc_g);
                RCCL_TRY(rc_e);
                return 0;
            };
            if (int rc = exchange(16, 16, 16, 16, 0)) return rc;
            if (c.mig_every == 0) {
                // the ranks agree on the smallest estimate (one float, ncclMin), read back with the headers
                RCCL_TRY(a->all_reduce(&e->dp.ctl->mig_quiet, c.mig_quiet_all, 1, 7 /* ncclFloat32 */, 3 /* ncclMin */, c.comm,
                                       e->stream));
            }
            {
                uint32_t h[4][4] = {};
                void* bufs[4] = {c.mig_send_l, c.mig_send_r, c.mig_recv_l, c.mig_recv_r};
                for (int k = 0; k < 4; ++k) HIP_TRY(hipMemcpyAsync(h[k], bufs[k], 16, hipMemcpyDeviceToHost, e->stream));
                HIP_TRY(hipStreamSynchronize(e->stream));
                auto bytes = [&](int k, bool there) {
                    return there ? (size_t)std::min<size_t>(h[k][0], c.mig_cap) * DIST_REC_F4 * 16 : (size_t)0;
                };
                // (a ring of one or two ranks pairs the k-th send with the k-th receive: what goes left arrives "from the
                // right" over there -- the received headers say how much comes from each side)
                if (int rc = exchange(bytes(0, c.left >= 0), bytes(1, c.right >= 0), bytes(2, c.left >= 0), bytes(3, c.right >= 0), 16))
                    return rc;
            }
            if (int rc = mpm_dist_migrate_apply(e, c.left >= 0 ? c.mig_recv_l : nullptr, c.right >= 0 ? c.mig_recv_r : nullptr,
                                                c.mig_cap))
                return rc;
            if (c.mig_every == 0) {
                float t = 0.f;
                D2H(e, &t, c.mig_quiet_all, sizeof(float));   // (the stream is idle: apply has just synchronised it)
                // (the estimate is ballistic: forces can speed particles up before the next look.  The interval may at
                // most double from one migration to the next -- 4 substeps after the first --, so a change of the
                // velocities is seen after at most as long as they have been watched)
                const float t_est = e->mig_safety * (t >= 0.f ? t : 0.f);   // (NaN -> 0: migrate again next substep)
                c.mig_budget = std::min(t_est, std::max(4.f * dt, 2.f * c.mig_elapsed));
                c.mig_elapsed = 0.f;
                if (int rc = mpm_dist_retune(e, t, dt, nullptr)) return rc;   // (band widths for the migrations to come)
            }
        }
        e->chain_lean = s + 1 < n;   // (reset below; the two calls are public entry points of their own as well)
        if (c.direct && nz > 0) {
            // DIRECT: the pack kernel stores into the neighbours' receive buffers of this substep's parity (two in
            // rotation: a neighbour may still be reading the other one -- it cannot be reading this one: its read of
            // substep s - 2 precedes its signal of s - 1, which this rank's update of s - 1 has waited for), a one-thread
            // kernel raises the flags over there, a one-wave kernel waits for this rank's.
            int rc_d = chain_direct_begin(e, dt);
            if (!rc_d) rc_d = chain_direct_end(e, dt, bc, true);
            e->chain_lean = 0;
            if (rc_d) return rc_d;
            continue;
        }
        c.mig_elapsed += dt;
        c.steps += 1;
        if (int rc = mpm_substep_begin_halo(e, dt, nz, lo, hi, sh, sb, c.cap)) {
            e->chain_lean = 0;
            return rc;
        }
        if (nz > 0) {
            RCCL_TRY(a->group_start());
            // what goes to the left arrives "from the right" over there: when both neighbours are the
            // same rank (ring of one or two) the k-th send pairs with the k-th receive, so the receives
            // are posted right-then-left against sends left-then-right
            int rc_g = 0;
            if (!rc_g && c.left >= 0) rc_g = a->send(c.send_l, c.bytes, 0 /* ncclChar */, c.left, c.comm, e->stream);
            if (!rc_g && c.right >= 0) rc_g = a->send(c.send_r, c.bytes, 0, c.right, c.comm, e->stream);
            if (!rc_g && c.right >= 0) rc_g = a->recv(c.recv_r, c.bytes, 0, c.right, c.comm, e->stream);
            if (!rc_g && c.left >= 0) rc_g = a->recv(c.recv_l, c.bytes, 0, c.left, c.comm, e->stream);
            const int rc_e = a->group_end();   // a failed call must not leave the group open
            RCCL_TRY(rc_g);
            RCCL_TRY(rc_e);
        }
        const int rc_end = mpm_substep_end_halo(e, dt, bc, nz, rb, c.cap);
        e->chain_lean = 0;
        if (rc_end) return rc_end;
    }
    HIP_TRY(hipGetLastError());
    return 0;
} MPM_CATCH_ALL

int mpm_grid_to_particle(mpm_handle_t e, float dt) try {
    if (e && can_defer(e) && e->pend.n == 4 && dt == e->pend.dt) {
        // the substep is complete: one gated substep, as mpm_run_substeps(1) enqueues it
        e->pend.n = 0;
        return mpm_run_substeps(e, 1, dt, e->pend.bc);
    }
    READY(e);
    REQUIRE(e->grid_state == 2, "GridToParticle before UpdateGrid");
    launch_g2p(e, dt);
    e->substeps += 1;
    return 0;
} MPM_CATCH_ALL

int mpm_substep(mpm_handle_t e, float dt, int bc) try { return mpm_run_substeps(e, 1, dt, bc); } MPM_CATCH_ALL

// allow_gate: the substep may go without the re-sort launches (mpm_run_substeps outside graphs)
// lean: another substep follows in the same batch, GridToParticle need not refresh what only a download reads
static void launch_substep(mpm_engine* e, float dt, const GridColliders& gc, bool allow_gate, bool lean) {
    e->last_dt = dt;
    // Re-sort check launches: with every substep that may not skip itself; otherwise none while the quiet time that
    // the last re-sort estimated lasts (half of it: the estimate is ballistic, elastic forces are not in it), then
    // with every check_every-th substep.  A wrong guess costs time, not correctness: substeps that find a re-sort
    // pending without their check skip themselves and are run again by settle().
    // (a quiet time left over from the settle() that has just read it also stands for "no re-sort pending, nothing
    // has touched the state since": the check that otherwise follows every other call is not needed either)
    const bool quiet = allow_gate && e->quiet_factor * e->quiet_left > dt;
    bool check = !allow_gate || (e->force_check && !quiet) || e->check_every <= 1 || e->dp.dist.on;
    if (!check) {
        if (quiet) {
            e->quiet_left -= dt / e->quiet_factor;
        } else {
            e->quiet_left = 0.f;
            check = (e->step_phase % (unsigned)e->check_every) == 0u;
        }
    }
    e->step_phase += 1;
    e->dp.gated = check ? 2 : 3;   // (bit 1: it also skips itself while the slab pool is too small, see DP::gated)
    if (check) {
        e->dp.lean_resort = !e->dp.dist.on;
        launch_rebuild(e);
        e->dp.lean_resort = 0;
        e->force_check = false;
    }
    e->maybe_owed = true;
    e->dp.lean_g2p = lean && !e->dp.dist.on;   // (k_p2g: the forces stay in LDS; k_g2p: no face x / v records)
    launch_fem_p2g(e, dt);
    launch_grid(e, gc);
    launch_g2p(e, dt);
    e->dp.lean_g2p = 0;
}

// A substep is nine dependent kernels with constant arguments and no host decisions (the re-sort
// is decided on the device), so a batch of substeps replays one captured graph: the graph's
// kernel-to-kernel hand-over is cheaper than nine stream dispatches.
static int step_graph_for(mpm_engine* e, float dt, int bc, const GridColliders& gc) {
    if (e->step_graph && e->step_graph_dt == dt && e->step_graph_bc == bc && e->step_graph_stream == e->stream &&
        e->step_graph_gcv == e->grid_colliders_version)
        return 0;
    drop_step_graph(e);
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < e->step_graph_len; ++k) launch_substep(e, dt, gc, false);
    HIP_TRY(hipStreamEndCapture(e->stream, &g));
    const hipError_t err = hipGraphInstantiate(&e->step_graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    HIP_TRY(err);
    e->step_graph_dt = dt;
    e->step_graph_bc = bc;
    e->step_graph_gcv = e->grid_colliders_version;
    e->step_graph_stream = e->stream;
    return 0;
}

int mpm_run_substeps(mpm_handle_t e, int n, float dt, int bc) try {
    READY_NO_SETTLE(e);
    // owed substeps are run with the parameters they were enqueued with: settle before these change (and before phase
    // calls that were held back, which come first)
    if (e->pend.n || (e->maybe_owed && (dt != e->owed_dt || bc != e->owed_bc || e->grid_colliders_version != e->owed_gcv)))
        if (int rc = settle(e)) return rc;
    e->owed_dt = dt;
    e->owed_bc = bc;
    e->owed_gcv = e->grid_colliders_version;
    // MPM_GRAPH=<substeps per graph> replays captured graphs; measured slower than plain stream
    // dispatch on ROCm 7.2, hence opt-in
    const int graph_len = e->graph_len;
    may_resort(e, dt);
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    int s = 0;
    if (graph_len > 0 && n >= graph_len) {
        e->step_graph_len = graph_len;
        if (int rc = step_graph_for(e, dt, bc, gc)) return rc;
        for (; s + graph_len <= n; s += graph_len) {
            HIP_TRY(hipGraphLaunch(e->step_graph, e->stream));
            // (launch_substep sets this while a graph is CAPTURED; a replay enqueues the same self-skipping substeps)
            e->maybe_owed = true;
        }
    }
    for (; s < n; ++s) launch_substep(e, dt, gc, true, s + 1 < n);
    e->grid_state = 2;
    e->substeps += (uint64_t)std::max(n, 0);
    HIP_TRY(hipGetLastError());
    return 0;
} MPM_CATCH_ALL

int mpm_profile_substeps(mpm_handle_t e, int n, float dt, int bc, float* phase_ms, float* total_ms) try {
    READY(e);
    REQUIRE(n > 0 && n <= 4096, "n out of range");
    GridColliders gc;
    if (int rc = grid_colliders_for(e, bc, &gc)) return rc;
    const int NE = MPM_PHASE_COUNT + 1;
    may_resort(e, dt);
    std::vector<hipEvent_t> ev((size_t)n * NE);
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    for (int s = 0; s < n; ++s) {
        hipEvent_t* q = &ev[(size_t)s * NE];
        HIP_TRY(hipEventRecord(q[0], e->stream));
        launch_rebuild(e);
        HIP_TRY(hipEventRecord(q[1], e->stream));
        launch_fem_faces(e, dt);
        HIP_TRY(hipEventRecord(q[2], e->stream));
        // (as in mpm_run_substeps: the vertex forces are part of k_p2g; this phase is empty)
        HIP_TRY(hipEventRecord(q[3], e->stream));
        e->dp.lean_g2p = s + 1 < n && !e->dp.dist.on;   // (as in mpm_run_substeps)
        {
            const int forces = fused_forces(e);
            if (!forces) launch_fem_vertices(e);
            launch_p2g(e, dt, forces);
        }
        HIP_TRY(hipEventRecord(q[4], e->stream));
        launch_grid(e, gc);
        HIP_TRY(hipEventRecord(q[5], e->stream));
        launch_g2p(e, dt);
        e->dp.lean_g2p = 0;
        HIP_TRY(hipEventRecord(q[6], e->stream));
    }
    e->grid_state = 2;
    e->substeps += (uint64_t)n;
    HIP_TRY(hipStreamSynchronize(e->stream));
    double acc[MPM_PHASE_COUNT] = {}, tot = 0;
    for (int s = 0; s < n; ++s) {
        hipEvent_t* q = &ev[(size_t)s * NE];
        for (int k = 0; k < MPM_PHASE_COUNT; ++k) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, q[k], q[k + 1]));
            acc[k] += ms;
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, q[0], q[MPM_PHASE_COUNT]));
        tot += ms;
    }
    for (auto& x : ev) hipEventDestroy(x);
    if (phase_ms)
        for (int k = 0; k < MPM_PHASE_COUNT; ++k) phase_ms[k] = (float)(acc[k] / n);
    if (total_ms) *total_ms = (float)(tot / n);
    return 0;
} MPM_CATCH_ALL

int mpm_get_stats(mpm_handle_t e, mpm_stats_t* out) try {
    READY(e);
    REQUIRE(out, "null stats");
    HIP_TRY(hipStreamSynchronize(e->stream));
    Ctl c;
    D2H(e, &c, e->dp.ctl, sizeof(Ctl));
    out->substeps = e->substeps;
    out->rebuilds = c.rebuilds;
    out->home_blocks = c.n_home;
    out->active_blocks = c.n_active;
    out->error_flags = c.error;
    if (!c.error)
        if (int rc = slab_pool_grow(e, c)) return rc;
    note_quiet_time(e, c);   // (a read-only call: the hint its settle() dropped still holds)
    out->active_faces = (uint32_t)c.nfa;
    out->active_vertices = (uint32_t)c.nva;
    out->face_slots = (uint32_t)e->dp.Nf;
    out->vertex_slots = (uint32_t)e->dp.Nv;
    out->resort_checks = e->checks_launched;
    out->quiet_time_s = c.quiet_time;
    out->since_resort_s = c.time_since_resort;
    {
        const DP& p = e->dp;
        size_t pb = 0, sb = 0;
        for (int s = 0; s < 2; ++s) {
            const PSet& S = p.set[s];
            pb += e->bytes_of(S.q[0]) + e->bytes_of(S.pid) + e->bytes_of(S.f8) + e->bytes_of(S.c8) + e->bytes_of(p.fg[s]);
            for (int d = 0; d < 4; ++d) pb += e->bytes_of(S.fq[d]);
            for (int d = 0; d < 2; ++d) pb += e->bytes_of(p.vg[s][d]);
        }
        pb += e->bytes_of(p.ta) + e->bytes_of(p.G3) + e->bytes_of(p.VF) + e->bytes_of(p.f[0]) + e->bytes_of(p.pkey) +
              e->bytes_of(p.prank) + e->bytes_of(p.src_of) + e->bytes_of(p.dst_of) + e->bytes_of(p.home_groups);
        sb += e->bytes_of(p.imap) + e->bytes_of(e->d_pids_api) + e->bytes_of(e->d_apimap) + e->bytes_of(p.dist.prev);
        for (int d = 0; d < 3; ++d) sb += e->bytes_of(p.idx_orig[d]);
        sb += e->bytes_of(p.adj_off) + e->bytes_of(p.adj_fc) + e->bytes_of(p.dm_orig);
        out->particle_bytes = pb;
        out->scene_index_bytes = sb;
    }
    out->touched_blocks = 0;
    if (e->grid_state >= 1) {
        uint32_t cnt = 0;
        if (int rc = touched_flags(e, nullptr, &cnt)) return rc;
        out->touched_blocks = cnt;
    }
    return 0;
} MPM_CATCH_ALL

int mpm_grid_touched_cnt(mpm_handle_t e, uint32_t* out) try {
    READY(e);
    REQUIRE(out, "null output");
    *out = 0;
    if (e->grid_state < 1) return 0;
    return touched_flags(e, nullptr, out);
} MPM_CATCH_ALL

int mpm_download_array(mpm_handle_t e, int which, void* out, size_t bytes, size_t* written) try {
    READY(e);
    REQUIRE(out, "null output");
    return download_array(e, which, out, bytes, written);
} MPM_CATCH_ALL

int mpm_upload_particle_state(mpm_handle_t e, const float* pos, const float* vel, const float* affine,
                              const float* volumes, const float* deformation_gradients) try {
    READY(e);
    if (int rc = upload_state(e, pos, vel, affine, volumes, deformation_gradients)) return rc;
    return volumes ? set_fixed_point_scales(e) : 0;
} MPM_CATCH_ALL

int mpm_sync_particle_state_to_cpu(mpm_handle_t e, float* pos_out) try {
    READY(e);
    REQUIRE(pos_out, "null output");
    return download_array(e, MPM_ARR_POSITIONS, pos_out, e->np * 12, nullptr);
} MPM_CATCH_ALL

int mpm_dump_cpu_state(mpm_handle_t e, float* pos_out, int32_t* idx_out) try {
    READY(e);
    if (pos_out) {
        // original vertex order = original ids nf..np (cuda_mpm_model.cu:257-260)
        if (int rc = download_original_vertices(e, pos_out)) return rc;
    }
    if (idx_out) std::copy(e->h_idx.begin(), e->h_idx.end(), idx_out);
    return 0;
} MPM_CATCH_ALL

int mpm_dump_obj(mpm_handle_t e, const char* filename) try {
    READY(e);
    REQUIRE(filename, "null filename");
    std::vector<float> pos(e->nv * 3);
    if (int rc = download_original_vertices(e, pos.data())) return rc;
    std::ofstream obj(filename);
    if (!obj) return fail(MPM_ERR_INVALID, std::string("cannot open ") + filename);
    for (size_t i = 0; i < e->nv; ++i) obj << "v " << pos[i * 3] << " " << pos[i * 3 + 1] << " " << pos[i * 3 + 2] << "\n";
    for (size_t f = 0; f < e->nf; ++f)
        obj << "f " << e->h_idx[f * 3] + 1 << " " << e->h_idx[f * 3 + 1] + 1 << " " << e->h_idx[f * 3 + 2] + 1 << "\n";
    return 0;
} MPM_CATCH_ALL

int mpm_debug_counters(mpm_handle_t e, uint64_t* out16, int reset) try {
    READY(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (out16) D2H(e, out16, e->dp.dbgbuf, 16 * 8);
    if (reset) HIP_TRY(hipMemsetAsync(e->dp.dbgbuf, 0, 16 * 8, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
} MPM_CATCH_ALL

int mpm_set_dump_dir(mpm_handle_t e, const char* dir) try {
    REQUIRE(e && dir, "null argument");
    e->dump_dir = dir;
    return 0;
} MPM_CATCH_ALL

int mpm_reallocate_external_bodies(mpm_handle_t e, size_t n) try {
    READY(e);
    if (e->cb.resize_bodies(n, e->stream))
        return fail(MPM_ERR_HIP, "ReallocateExternelBodies: device allocation or reset failed");
    return 0;
} MPM_CATCH_ALL

int mpm_external_body_force_to_host(mpm_handle_t e, float* tau_out, float* f_out) try {
    READY(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->cb.n_bodies == 0) return 0;
    // (accumulated as integers in 64-bit fixed point, k_ct_impulse: the same bits whatever order the contacts arrived in)
    std::vector<long long> acc(e->cb.n_bodies * 6);
    D2H(e, acc.data(), e->cb.body_acc, acc.size() * sizeof(long long));
    const double unfix = e->cb.imp_unfix > 0.0 ? e->cb.imp_unfix : e->dp.unfix_p;
    for (size_t b = 0; b < e->cb.n_bodies; ++b)
        for (int a = 0; a < 3; ++a) {
            if (tau_out) tau_out[b * 3 + a] = (float)((double)acc[b * 6 + a] * unfix);
            if (f_out) f_out[b * 3 + a] = (float)((double)acc[b * 6 + 3 + a] * unfix);
        }
    return 0;
} MPM_CATCH_ALL

// ---- partitioned domain ----------------------------------------------------------------------
namespace mpm {
// after the arrays have moved: vertex slots start at new_nf instead of old_nf
__global__ __launch_bounds__(256) void k_dist_shift_slots(DP p, int old_nf, int new_nf, int nfa) {
    const PSet& S = p.set[p.ctl->cur];
    const int gs = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    auto shifted = [&](int s) { return s >= old_nf ? s - old_nf + new_nf : s; };
    for (int i = t0; i < nfa; i += gs) {
        float4 f3 = S.fq[3][i];
        f3.y = __int_as_float(shifted(__float_as_int(f3.y)));
        f3.z = __int_as_float(shifted(__float_as_int(f3.z)));
        f3.w = __int_as_float(shifted(__float_as_int(f3.w)));
        S.fq[3][i] = f3;
    }
    for (int g = t0; g < p.NpG; g += gs) p.imap[g] = shifted(p.imap[g]);
}
}  // namespace mpm

// Slot space of a partitioned rank.  The rank was finalised with the whole scene (every rank runs the same Finalize, so
// ghost copies start bit-identical to their owners'); the first partitioned re-sort has compacted what it keeps into
// slots [0, nfa) and [Nf, Nf + nva).  dist_resize re-allocates every array indexed by particle slot for `new_nf` face and
// `new_nv` vertex slots, copies the held ranges, shifts the slot references and asks for a re-sort (block tables, work
// items and wave groups hold slot ranges).  mpm_dist_init calls it once to SHRINK the rank to headroom x its share and
// to release the tables of the whole scene's topology (corner ids per face, adjacency CSR, Dm^-1 by original id: per
// slot the rank keeps the same information by original id, DP::fg / vg, and migration records carry it); a migration
// that would overflow the slot space calls it again to GROW (mpm_dist_migrate_apply: the stream is idle there).  What
// stays whole-scene sized is the id -> slot map and the slot-order bookkeeping (13 bytes per particle of the scene).
static size_t dist_slot_capacity(const mpm_engine* e, size_t held, size_t all) {
    if (!(e->dist_headroom > 0.f)) return all;   // (0: keep the whole scene's size)
    return std::min(all, std::max<size_t>((size_t)((double)held * std::max(1.f, e->dist_headroom)) + 256, 1024));
}
// TWO-PHASE: phase 1 allocates every new array and touches nothing of the engine -- a failure there (device
// memory exhausted) frees what phase 1 got and returns with the engine exactly as it was, still usable at its old size;
// phase 2 (copies, swaps, frees) cannot fail for lack of memory.
static int dist_resize(mpm_engine* e, size_t new_nf, size_t new_nv, bool first) {
    DP& p = e->dp;
    HIP_TRY(hipStreamSynchronize(e->stream));
    Ctl c;
    D2H(e, &c, p.ctl, sizeof(Ctl));
    // (what a migration appended behind the active particles and the re-sort has not merged yet is kept too)
    REQUIRE(c.nfa >= 0 && c.nva >= 0 && c.add_f >= 0 && c.add_v >= 0, "dist_resize: corrupt control block");
    const size_t nfa = (size_t)c.nfa + (size_t)c.add_f, nva = (size_t)c.nva + (size_t)c.add_v;
    const size_t old_nf = (size_t)p.Nf;
    REQUIRE(nfa <= old_nf && nva <= (size_t)p.Nv, "dist_resize: the control block counts more particles than the slot space holds");
    REQUIRE(new_nf >= nfa && new_nv >= nva, "dist_resize: smaller than what the rank holds");
    REQUIRE(new_nf + new_nv < ((size_t)1 << 30), "dist_resize: too many particle slots");
    const size_t new_np = new_nf + new_nv;
    const int cur = c.cur & 1;
    const unsigned new_q_stride = (unsigned)((new_np + 63) & ~(size_t)63);

    // ---- phase 1: allocate ----------------------------------------------------------------------------------------
    struct Copy { size_t dst_off, src_off, bytes; };
    struct Plan {
        void** owner;             // where the engine keeps the allocation's base pointer
        void* fresh;              // its replacement
        std::vector<Copy> keep;   // byte ranges carried over (old base -> fresh)
    };
    std::vector<Plan> plans;
    plans.reserve(64);
    const size_t allocs_before = e->allocs.size();
    auto rollback = [&]() {
        for (Plan& pl : plans) {
            void* q = pl.fresh;
            if (q) e->dfree(q);
        }
        (void)allocs_before;
    };
    int rc = 0;
    auto plan = [&](auto*& ptr, size_t n_new, std::vector<Copy> keep_elems) -> int {
        using T = std::remove_pointer_t<std::remove_reference_t<decltype(ptr)>>;
        T* fresh = nullptr;
        if (int r = e->dalloc(&fresh, n_new, true)) return r;
        Plan pl;
        pl.owner = (void**)(&ptr);
        pl.fresh = (void*)fresh;
        for (Copy k : keep_elems)
            if (k.bytes) pl.keep.push_back(Copy{k.dst_off * sizeof(T), k.src_off * sizeof(T), k.bytes * sizeof(T)});
        plans.push_back(std::move(pl));
        return 0;
    };
    float4* q_base[2] = {p.set[0].q[0], p.set[1].q[0]};   // (the four planes of a set are one allocation)
    float* f_base = p.f[0];
    for (int s = 0; s < 2 && !rc; ++s) {
        PSet& S = p.set[s];
        const bool live = s == cur;
        std::vector<Copy> planes;
        if (live)
            for (int d = 0; d < 4; ++d) {
                planes.push_back(Copy{(size_t)d * new_q_stride, (size_t)d * p.q_stride, nfa});
                planes.push_back(Copy{(size_t)d * new_q_stride + new_nf, (size_t)d * p.q_stride + old_nf, nva});
            }
        rc = plan(q_base[s], 4 * (size_t)new_q_stride, planes);
        if (!rc) rc = plan(S.pid, new_np, live ? std::vector<Copy>{{0, 0, nfa}, {new_nf, old_nf, nva}} : std::vector<Copy>{});
        const std::vector<Copy> kf = live ? std::vector<Copy>{{0, 0, nfa}} : std::vector<Copy>{};
        const std::vector<Copy> kv = live ? std::vector<Copy>{{0, 0, nva}} : std::vector<Copy>{};
        for (int d = 0; d < 4 && !rc; ++d) rc = plan(S.fq[d], new_nf, kf);
        if (!rc) rc = plan(S.f8, new_nf, kf);
        if (!rc) rc = plan(S.c8, new_nf, kf);
        if (!rc) rc = plan(p.fg[s], new_nf, kf);
        for (int d = 0; d < 2 && !rc; ++d) rc = plan(p.vg[s][d], new_nv, kv);
    }
    // per-substep outputs and re-sort scratch: nothing to keep
    if (!rc) rc = plan(p.ta, new_nf, {});
    if (!rc) rc = plan(p.G3, 3 * new_nf, {});
    // (the vertices' rows of force triples: zero-filled; the re-sort that follows writes the zeros / marks of every held
    // vertex again and k_fem the rest before anything reads them)
    if (!rc) rc = plan(p.VF, 3 * vf_entry((unsigned)new_nv + VF_CHUNK, 0), {});
    if (!rc) rc = plan(f_base, 3 * (size_t)new_q_stride, {});
    if (!rc) rc = plan(p.pkey, new_np, {});
    if (!rc) rc = plan(p.prank, new_np, {});
    if (!rc) rc = plan(p.src_of, new_np, {});
    if (!rc) rc = plan(p.dst_of, new_np, {});
    if (!rc) rc = plan(p.home_groups, new_np / 64 + p.capH + 2, {});
    if (rc) {
        const std::string why = g_last_error;
        rollback();
        (void)hipStreamSynchronize(e->stream);
        return fail(rc, "dist_resize: " + why + " -- the rank keeps its slot space of " + std::to_string(p.Nf) + " + " +
                            std::to_string(p.Nv) + " slots and stays usable");
    }

    // ---- phase 2: copy, swap, free (no allocation from here on) ------------------------------------------------------
    for (Plan& pl : plans)
        for (const Copy& k : pl.keep)
            HIP_TRY(hipMemcpyAsync((char*)pl.fresh + k.dst_off, (const char*)*pl.owner + k.src_off, k.bytes,
                                   hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (Plan& pl : plans) {
        void* old = *pl.owner;
        e->dfree(old);
        *pl.owner = pl.fresh;
    }
    for (int s = 0; s < 2; ++s)
        for (int d = 0; d < 4; ++d) p.set[s].q[d] = q_base[s] + (size_t)d * new_q_stride;
    for (int d = 0; d < 3; ++d) p.f[d] = f_base + (size_t)d * new_q_stride;
    if (first) {
        // the whole scene's topology tables
        for (int d = 0; d < 3; ++d) {
            int* t = const_cast<int*>(p.idx_orig[d]);
            e->dfree(t);
            p.idx_orig[d] = nullptr;
        }
        float4* dm = const_cast<float4*>(p.dm_orig);
        e->dfree(dm);
        p.dm_orig = nullptr;
        if (e->max_valence <= 8) {
            int* a = const_cast<int*>(p.adj_off);
            int* b = const_cast<int*>(p.adj_fc);
            e->dfree(a); e->dfree(b);
            p.adj_off = nullptr; p.adj_fc = nullptr;
        }
        // (a mesh with a vertex of more than eight faces keeps the scene's adjacency -- 4 bytes per vertex and 12 per face
        // on every rank: such a vertex's record holds a mark instead of face ids, and its force walks the adjacency by
        // original id through the id -> slot map, wherever the vertex migrates: vertex_force_csr)
    }
    p.Np = (int)new_np; p.Nf = (int)new_nf; p.Nv = (int)new_nv;
    p.q_stride = new_q_stride;
    p.f_stride = new_q_stride;
    drop_step_graph(e);
    for (auto& kg : e->halo_graph) {
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
        kg.exec = nullptr;
    }
    hipLaunchKernelGGL(k_dist_shift_slots, dim3(1024), dim3(256), 0, e->stream, p, (int)old_nf, (int)new_nf, (int)nfa);
    // block tables, work items and wave groups hold slot ranges: rebuild them on the new slot space
    const int one = 1;
    H2D(e, &p.ctl->need_rebuild, &one, sizeof(int));
    e->dist_resizes += 1;
    if (!first) return 0;   // (a migration follows, and the re-sort that merges what it brings)
    may_resort(e, 0.f);
    launch_rebuild(e);
    D2H(e, &c, p.ctl, sizeof(Ctl));
    return recover_slab_overflow(e, c);
}

int mpm_dist_init(mpm_handle_t e, const mpm_dist_config_t* cfg) try {
    READY(e);
    REQUIRE(cfg, "null configuration");
    REQUIRE(!e->dp.dist.on, "mpm_dist_init called twice");
    REQUIRE(e->api_identity, "mpm_dist_init must precede RebuildMapping(sort = true)");
    const int nb = e->dp.nb;
    REQUIRE(cfg->world >= 1 && cfg->rank >= 0 && cfg->rank < cfg->world, "bad rank / world");
    REQUIRE(cfg->own_lo_block >= 0 && cfg->own_lo_block < cfg->own_hi_block && cfg->own_hi_block <= nb,
            "bad slab: need 0 <= own_lo_block < own_hi_block <= blocks per axis");
    const bool auto_bands = cfg->ghost_cells == 0 && cfg->ghost_margin_cells == 0;
    REQUIRE(cfg->zone_blocks >= 1 && (auto_bands || (cfg->ghost_cells >= 1 && cfg->ghost_margin_cells >= 1)),
            "zone_blocks must be positive; ghost_cells and ghost_margin_cells both positive, or both 0 (bands from the mesh)");
    // a ghost vertex may sit ghost_cells + ghost_margin_cells beyond the cut and its stencil reaches 2 nodes further
    REQUIRE(auto_bands || cfg->zone_blocks * 4 >= cfg->ghost_cells + cfg->ghost_margin_cells + 2,
            "zone too shallow for the ghost band: need 4 * zone_blocks >= ghost_cells + ghost_margin_cells + 2");
    // Reach of a face: how far a corner vertex can be from the face particle (the centroid), in cells -- 2/3 of a
    // median, bounded by 0.75 x the longest edge of the mesh as it was handed over (a cloth stretches by per cent).
    float longest_edge = 0.f;
    for (size_t f = 0; f < e->nf; ++f)
        for (int k = 0; k < 3; ++k) {
            const float* A = &e->h_pos[(size_t)e->h_idx[f * 3 + k] * 3];
            const float* B = &e->h_pos[(size_t)e->h_idx[f * 3 + (k + 1) % 3] * 3];
            longest_edge = std::max(longest_edge, std::sqrt((A[0] - B[0]) * (A[0] - B[0]) + (A[1] - B[1]) * (A[1] - B[1]) +
                                                            (A[2] - B[2]) * (A[2] - B[2])));
        }
    longest_edge *= e->dp.dxinv;
    const float reach = .75f * longest_edge;
    // Drift budget.  Between two migrations nothing joins or leaves a rank, so what a rank holds must cover what its
    // owned particles need even after every particle has drifted `delta` cells along x:
    //   an owned vertex needs its adjacent faces:  ghost_w >= reach + 2 delta   (both may have moved)
    //   a face needs its corners:                  vert_w  >= ghost_w + reach   (holds at the migration, nothing leaves after)
    //   a ghost's stencil stays inside the zone:   vert_w + delta <= 4 zone_blocks - 2
    //   an owned particle's stencil does:          delta <= 4 zone_blocks - 3
    // bands from the mesh (ghost_cells = ghost_margin_cells = 0): the widths that make delta largest; given widths:
    // the delta they allow (0: migrate with every substep).
    // (with the hysteresis h of ownership and band membership, Dist::hyst: an owned particle may sit h beyond a cut and a
    // held ghost h beyond its band when the drift starts)
    //   ghost_w >= reach + 2 delta + h;  vert_w >= ghost_w + reach;  vert_w + h + delta <= 4 zone_blocks - 2;
    //   h + delta <= 4 zone_blocks - 3
    const float zone_room = (float)(4 * cfg->zone_blocks - 2);
    float ghost_w, vert_w, delta, hyst = .125f;
    if (auto_bands) {
        // (as much drift as the zone allows, but no more than dist_drift_target: every cell of band is a cell of ghost
        // particles that FEM and GridToParticle advance with every substep -- with slabs of 16 cells the widest bands
        // of a two-block zone, 3.8 / 4.3 cells, are half as many ghosts as owned particles)
        delta = std::min({(zone_room - 2.f * reach - 2.f * hyst) / 3.f, zone_room - 1.f - hyst, e->dist_drift_target});
        REQUIRE(delta > .05f, "mpm_dist_init: the exchanged zone is too shallow for this mesh (4 zone_blocks - 2 must exceed 1.5 x "
                              "the longest mesh edge in cells): use more zone_blocks or a finer mesh");
        ghost_w = reach + 2.f * delta + hyst;
        vert_w = ghost_w + reach;
    } else {
        ghost_w = (float)cfg->ghost_cells;
        vert_w = (float)(cfg->ghost_cells + cfg->ghost_margin_cells);
        if (zone_room - vert_w < 2.f * hyst) hyst = 0.f;   // (bands that fill the zone: no room for it)
        delta = std::max(0.f, std::min({(ghost_w - reach - hyst) * .5f, zone_room - vert_w - hyst, zone_room - 1.f - hyst}));
    }
    const bool has_left = cfg->rank > 0, has_right = cfg->rank < cfg->world - 1;
    // the zones of the two cuts must not overlap (a block is shared by at most two ranks)
    REQUIRE(!(has_left && has_right) || cfg->own_hi_block - cfg->own_lo_block >= 2 * cfg->zone_blocks,
            "slab narrower than two zones: use fewer ranks or zone_blocks = 1");
    REQUIRE(!has_left || (cfg->left_lo_block >= 0 && cfg->left_lo_block < cfg->own_lo_block), "bad left neighbour range");
    REQUIRE(!has_right || (cfg->right_hi_block > cfg->own_hi_block && cfg->right_hi_block <= nb), "bad right neighbour range");
    Dist d{};
    d.on = 1;
    d.rank = cfg->rank; d.world = cfg->world;
    d.own_lo = has_left ? cfg->own_lo_block * 4 : 0;
    d.own_hi = has_right ? cfg->own_hi_block * 4 : nb * 4;
    d.nbr_lo = has_left ? (cfg->rank > 1 ? cfg->left_lo_block * 4 : 0) : 0;
    d.nbr_hi = has_right ? (cfg->rank < cfg->world - 2 ? cfg->right_hi_block * 4 : nb * 4) : nb * 4;
    d.has_left = has_left; d.has_right = has_right;
    d.ghost_w = ghost_w;
    d.vert_w = vert_w;
    d.hyst = hyst;
    d.zone_cells = cfg->zone_blocks * 4;
    d.mig_delta = delta;
    d.mig_reach = vert_w + 2.f * reach + 1.f;
    e->dist_longest_edge = longest_edge;
    e->dist_auto = auto_bands;
    e->dist_reach = reach;
    e->dist_hyst = hyst;
    e->dist_delta_max = std::min((zone_room - 2.f * reach - 2.f * hyst) / 3.f, zone_room - 1.f - hyst);
    if (int rc = e->dalloc(&d.prev, e->np, true)) return rc;
    if (int rc = e->dalloc(&d.mig_min, 32 * 32, true)) return rc;
    // per-slot topology by original id, for every particle at first (nobody has been released yet)
    for (int s = 0; s < 2; ++s) {
        if (int rc = e->dalloc(&e->dp.fg[s], e->nf, false)) return rc;
        for (int k = 0; k < 2; ++k)
            if (int rc = e->dalloc(&e->dp.vg[s][k], e->nv, false)) return rc;
    }
    hipLaunchKernelGGL(k_dist_build_topology, dim3(std::min(e->g_np, 2048u)), dim3(256), 0, e->stream, e->dp);
    e->dp.dist = d;
    e->dist_cfg = *cfg;
    drop_step_graph(e);
    hipLaunchKernelGGL(k_dist_init_roles, dim3(e->g_np), dim3(256), 0, e->stream, e->dp);
    int one = 1;
    H2D(e, &e->dp.ctl->need_rebuild, &one, sizeof(int));
    may_resort(e, 0.f);
    launch_rebuild(e);
    {
        Ctl c;
        D2H(e, &c, e->dp.ctl, sizeof(Ctl));
        if (int rc = recover_slab_overflow(e, c)) return rc;
    }
    // the rank keeps its share: the particle arrays shrink to it, the whole scene's topology tables go
    {
        Ctl c;
        D2H(e, &c, e->dp.ctl, sizeof(Ctl));
        if (int rc = dist_resize(e, dist_slot_capacity(e, (size_t)c.nfa, (size_t)e->dp.Nf),
                                 dist_slot_capacity(e, (size_t)c.nva, (size_t)e->dp.Nv), true))
            return rc;
    }
    return mpm_sync(e);
} MPM_CATCH_ALL

int mpm_dist_set_headroom(mpm_handle_t e, float factor) try {
    REQUIRE(e, "null handle");
    REQUIRE(!e->dp.dist.on, "mpm_dist_set_headroom must precede mpm_dist_init");
    REQUIRE(factor == 0.f || factor >= 1.f, "headroom must be 0 (keep the whole scene's size) or >= 1");
    e->dist_headroom = factor;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_get_geometry(mpm_handle_t e, mpm_dist_geometry_t* out) try {
    READY_NO_SETTLE(e);
    REQUIRE(out, "null output");
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    const Dist& d = e->dp.dist;
    out->face_band_cells = d.ghost_w;
    out->vertex_band_cells = d.vert_w;
    out->drift_budget_cells = d.mig_delta;
    out->longest_edge_cells = e->dist_longest_edge;
    out->slot_resizes = e->dist_resizes;
    out->migrations = e->dist_migrations;
    out->retunes = e->dist_retunes;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_retune(mpm_handle_t e, float quiet_time_all, float dt, int* changed_out) try {
    REQUIRE(e, "null handle");
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    if (changed_out) *changed_out = 0;
    if (!e->dist_auto || !e->dist_retune || !(dt > 0.f)) return 0;
    Dist& d = e->dp.dist;
    // The ranks' common estimate says the fastest relevant particle uses up the CURRENT budget in quiet_time_all: that
    // is a speed.  The budget that makes a migration due every dist_target_interval substeps at that speed (half of
    // the estimate is trusted), in steps of an eighth of a cell so that it does not flutter, within what the zone allows:
    float want = e->dist_delta_max;   // (an estimate of 0 or NaN: as wide as the zone allows)
    if (quiet_time_all > 0.f && std::isfinite(quiet_time_all)) {
        const float speed = d.mig_delta / quiet_time_all;   // cells per second
        want = speed * dt * e->dist_target_interval / e->mig_safety;
        want = std::ceil(want * 8.f) * .125f;
    } else if (quiet_time_all > 0.f) {
        want = .125f;   // (an infinite estimate -- nothing moves along x --: the narrowest bands)
    }
    want = std::min(std::max(want, .125f), e->dist_delta_max);
    if (!(want > .05f) || std::fabs(want - d.mig_delta) < .06f) return 0;
    // Every rank computes this from the same numbers at the same migration, so all switch together.  The new widths
    // take effect at the NEXT migration's classification (particles newly inside a wider band are sent then, ghosts
    // outside a narrower one are released then); the interval up to it was budgeted with the old widths.
    d.mig_delta = want;
    d.ghost_w = e->dist_reach + 2.f * want + e->dist_hyst;
    d.vert_w = d.ghost_w + e->dist_reach;
    d.mig_reach = d.vert_w + 2.f * e->dist_reach + 1.f;
    e->dist_retunes += 1;
    drop_step_graph(e);
    for (auto& kg : e->halo_graph) {   // (captured launches carry the Dist by value)
        if (kg.exec) (void)hipGraphExecDestroy(kg.exec);
        kg.exec = nullptr;
    }
    if (changed_out) *changed_out = 1;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_migration_quiet_time(mpm_handle_t e, float* seconds_out) try {
    READY(e);
    REQUIRE(seconds_out, "null output");
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    D2H(e, seconds_out, &e->dp.ctl->mig_quiet, sizeof(float));
    return 0;
} MPM_CATCH_ALL

size_t mpm_dist_migration_buffer_bytes(size_t capacity_particles) { return 16 + capacity_particles * DIST_REC_F4 * 16; }

int mpm_dist_migrate_pack(mpm_handle_t e, void* send_left, void* send_right, size_t capacity_particles) try {
    READY(e);
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    REQUIRE(send_left && send_right && capacity_particles > 0 && capacity_particles < (1u << 28), "bad migration buffers");
    HIP_TRY(hipMemsetAsync(send_left, 0, 16, e->stream));
    HIP_TRY(hipMemsetAsync(send_right, 0, 16, e->stream));
    hipLaunchKernelGGL(k_dist_classify, dim3(std::min(e->g_np, 2048u)), dim3(256), 0, e->stream, e->dp,
                       static_cast<float4*>(send_left), static_cast<float4*>(send_right), (unsigned)capacity_particles);
    // (the time until the next migration is due, as far as this rank can tell: Ctl::mig_quiet)
    hipLaunchKernelGGL(k_dist_mig_reduce, dim3(1), dim3(64), 0, e->stream, e->dp);
    e->dist_migrations += 1;
    return 0;
} MPM_CATCH_ALL

// What a migration brings, from the two 16-byte headers of the received buffers ([0] records, [1] how many of them are
// faces, [2..3] zero), and the slot space it needs.  Host arithmetic on numbers that came out of BUFFERS (another rank
// wrote them, a transport carried them): nothing is sized from them before they have been checked against the
// buffers' capacity and against the scene -- a header that fails the checks is an error code, never an allocation.
struct MigrationPlan {
    size_t in_f = 0, in_v = 0;       // arriving records: faces, vertices
    size_t need_f = 0, need_v = 0;   // slots the rank must have before k_dist_apply runs
    size_t want_f = 0, want_v = 0;   // slot space to re-allocate to (= the current one when it suffices)
};
static int plan_migration(const uint32_t* hdr_left, const uint32_t* hdr_right, size_t capacity, size_t scene_f, size_t scene_v,
                          size_t held_f, size_t held_v, size_t slots_f, size_t slots_v, float headroom, MigrationPlan* out) {
    REQUIRE(capacity > 0 && capacity < ((size_t)1 << 28), "migration: bad buffer capacity");
    REQUIRE(scene_f + scene_v < ((size_t)1 << 30), "migration: bad scene size");
    if (held_f > slots_f || held_v > slots_v || slots_f > scene_f || slots_v > scene_v)
        return fail(MPM_ERR_INTERNAL, "migration: the rank's own counts are inconsistent (holds " + std::to_string(held_f) + " + " +
                                          std::to_string(held_v) + " particles in " + std::to_string(slots_f) + " + " +
                                          std::to_string(slots_v) + " slots of a scene of " + std::to_string(scene_f) + " + " +
                                          std::to_string(scene_v) + ")");
    MigrationPlan m;
    const uint32_t* hdrs[2] = {hdr_left, hdr_right};
    for (int k = 0; k < 2; ++k) {
        const uint32_t* h = hdrs[k];
        if (!h) continue;
        const char* side = k ? "right" : "left";
        // (a sender whose buffer overflowed keeps counting: it raises MPM_ERR_CAPACITY on its side, and the records
        // beyond the capacity were never written -- applying the rest would lose particles silently)
        if (h[0] > capacity)
            return fail(MPM_ERR_CAPACITY, std::string("migration: the ") + side + " neighbour packed " + std::to_string(h[0]) +
                                              " records into buffers of " + std::to_string(capacity) +
                                              " (mpm_chain_enable_migration / DomainChain: raise capacity_particles)");
        if (h[1] > h[0] || h[2] != 0 || h[3] != 0)
            return fail(MPM_ERR_INVALID, std::string("migration: corrupt header from the ") + side + " neighbour (" +
                                             std::to_string(h[0]) + " records, " + std::to_string(h[1]) + " faces, " +
                                             std::to_string(h[2]) + ", " + std::to_string(h[3]) + ")");
        m.in_f += h[1];
        m.in_v += h[0] - h[1];
    }
    // (every arriving record counted as a new particle -- promotions of ghosts the rank already holds need no slot --,
    // but never more than the scene has: a particle has one slot)
    m.need_f = std::min(scene_f, held_f + m.in_f);
    m.need_v = std::min(scene_v, held_v + m.in_v);
    auto capacity_for = [&](size_t held, size_t all) {
        if (!(headroom > 0.f)) return all;   // (0: keep the whole scene's size)
        return std::min(all, std::max<size_t>((size_t)((double)held * std::max(1.f, headroom)) + 256, 1024));
    };
    m.want_f = slots_f;
    m.want_v = slots_v;
    if (m.need_f > slots_f || m.need_v > slots_v) {
        m.want_f = std::max(slots_f, capacity_for(m.need_f, scene_f));
        m.want_v = std::max(slots_v, capacity_for(m.need_v, scene_v));
    }
    *out = m;
    return 0;
}

int mpm_dist_plan_migration(const uint32_t* hdr_left, const uint32_t* hdr_right, size_t capacity_particles, size_t scene_faces,
                            size_t scene_vertices, size_t held_faces, size_t held_vertices, size_t face_slots,
                            size_t vertex_slots, float headroom, size_t out6[6]) try {
    REQUIRE(out6, "null output");
    MigrationPlan m;
    if (int rc = plan_migration(hdr_left, hdr_right, capacity_particles, scene_faces, scene_vertices, held_faces, held_vertices,
                                face_slots, vertex_slots, headroom, &m))
        return rc;
    out6[0] = m.in_f; out6[1] = m.in_v; out6[2] = m.need_f; out6[3] = m.need_v; out6[4] = m.want_f; out6[5] = m.want_v;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_migrate_apply(mpm_handle_t e, const void* recv_left, const void* recv_right, size_t capacity_particles) try {
    READY(e);
    REQUIRE(e->dp.dist.on, "mpm_dist_init first");
    REQUIRE(capacity_particles > 0 && capacity_particles < (1u << 28), "bad migration buffers");
    // A migration is a synchronisation point: the host reads how many particles arrive and what the rank holds, and
    // re-allocates the slot space first if they would not fit (a cloth that slides across a cut, an uneven split).
    {
        DP& p = e->dp;
        uint32_t hdr[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        // (pinned landing places are not needed: the copies are followed by a synchronisation at once)
        Ctl c;
        HIP_TRY(hipMemcpyAsync(&c, p.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, e->stream));
        const void* bufs[2] = {recv_left, recv_right};
        for (int k = 0; k < 2; ++k)
            if (bufs[k]) HIP_TRY(hipMemcpyAsync(hdr[k], bufs[k], 16, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        REQUIRE(c.nfa >= 0 && c.nva >= 0 && c.add_f >= 0 && c.add_v >= 0, "migration: corrupt control block");
        MigrationPlan m;
        if (int rc = plan_migration(recv_left ? hdr[0] : nullptr, recv_right ? hdr[1] : nullptr, capacity_particles, e->nf, e->nv,
                                    (size_t)c.nfa + (size_t)c.add_f, (size_t)c.nva + (size_t)c.add_v, (size_t)p.Nf, (size_t)p.Nv,
                                    e->dist_headroom, &m))
            return rc;
        if (m.want_f != (size_t)p.Nf || m.want_v != (size_t)p.Nv)
            if (int rc = dist_resize(e, m.want_f, m.want_v, false)) return rc;
    }
    for (const void* b : {recv_left, recv_right})
        if (b)
            hipLaunchKernelGGL(k_dist_apply, dim3(256), dim3(256), 0, e->stream, e->dp, static_cast<const float4*>(b),
                               (unsigned)capacity_particles);
    return 0;
} MPM_CATCH_ALL

// Tests of the exception barrier and of the allocation-failure paths.  mpm_debug_throw raises a C++ exception of the given
// kind INSIDE an entry point (0: std::bad_alloc, 1: std::length_error via a container asked for an absurd size, 2: a
// non-std exception) and must come back as MPM_ERR_NOMEM / MPM_ERR_INTERNAL; host code only.  mpm_debug_fail_alloc makes the
// engine's n-th device allocation from now fail as if the device were out of memory (0 = off).
int mpm_debug_throw(int kind) try {
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) {
        std::vector<float> v;
        v.resize(v.max_size() + 1);   // (what an unchecked count from a buffer does to a container)
        return (int)v.size();
    }
    if (kind == 2) throw 42;
    return fail(MPM_ERR_INVALID, "mpm_debug_throw: kind must be 0, 1 or 2");
} MPM_CATCH_ALL

int mpm_debug_fail_alloc(mpm_handle_t e, int nth) try {
    REQUIRE(e, "null handle");
    REQUIRE(nth >= 0, "nth must not be negative");
    e->fail_alloc_countdown = nth;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_set_transport(mpm_handle_t e, mpm_exchange_fn exchange, mpm_allreduce_fn allreduce, void* user,
                           size_t zone_capacity_blocks) try {
    REQUIRE(e, "null handle");
    REQUIRE(zone_capacity_blocks > 0 && zone_capacity_blocks < (1u << 24), "bad zone capacity");
    e->dist_exchange = exchange;
    e->dist_allreduce = allreduce;
    e->dist_user = user;
    e->dist_zone_cap = zone_capacity_blocks;
    return 0;
} MPM_CATCH_ALL

int mpm_dist_roles(mpm_handle_t e, uint8_t* out) try {
    READY(e);
    REQUIRE(out, "null output");
    if (int rc = e->stage(e->np)) return rc;
    hipLaunchKernelGGL(k_dist_roles, dim3(e->g_np), dim3(256), 0, e->stream, e->dp, (const int*)e->d_pids_api,
                       (unsigned char*)e->d_stage);
    HIP_TRY(hipMemcpyAsync(out, e->d_stage, e->np, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
} MPM_CATCH_ALL

int mpm_set_grid_colliders(mpm_handle_t e, size_t n, const mpm_grid_collider_t* colliders) try {
    static_assert(sizeof(GridCollider) == sizeof(mpm_grid_collider_t), "grid collider layouts differ");
    REQUIRE(e, "null handle");
    REQUIRE(n <= (size_t)MAX_GRID_COLLIDERS, "too many grid colliders (at most 16)");
    REQUIRE(n == 0 || colliders, "null collider array");
    if (e->finalized) {
        // substeps that mpm_run_substeps deferred are owed with the table they were enqueued with
        if (int rc = use(e)) return rc;
        if (int rc = settle(e)) return rc;
    }
    GridColliders gc{};
    for (size_t k = 0; k < n; ++k) {
        const mpm_grid_collider_t& c = colliders[k];
        REQUIRE(c.shape == MPM_GC_SPHERE || c.shape == MPM_GC_HALF_SPACE, "unknown grid collider shape");
        REQUIRE(c.mode >= MPM_GC_FIXED && c.mode <= MPM_GC_SLIP, "unknown grid collider mode");
        if (c.shape == MPM_GC_SPHERE) REQUIRE(c.radius > 0.f, "sphere radius must be positive");
        if (c.shape == MPM_GC_HALF_SPACE) {
            const float l = c.n[0] * c.n[0] + c.n[1] * c.n[1] + c.n[2] * c.n[2];
            REQUIRE(std::fabs(l - 1.f) < 1e-4f, "half-space normal must be a unit vector");
        }
        std::memcpy(&gc.c[k], &c, sizeof(GridCollider));
        if (c.friction < 0.f) gc.c[k].friction = e->mat.sdf_friction;
    }
    gc.n = (int)n;
    e->grid_colliders = gc;
    e->grid_colliders_version += 1;
    return 0;
} MPM_CATCH_ALL

int mpm_grid_collider_preset(int mpm_bc, float sdf_friction, mpm_grid_collider_t* out, size_t capacity, size_t* n_out) try {
    REQUIRE(n_out, "null argument");
    GridColliders gc{};
    if (grid_collider_preset(mpm_bc, sdf_friction, &gc)) return fail(MPM_ERR_INVALID, "mpm_bc must be -1, 0, 1, 2 or 3");
    *n_out = (size_t)gc.n;
    REQUIRE((size_t)gc.n <= capacity && (gc.n == 0 || out), "output array too small");
    for (int k = 0; k < gc.n; ++k) std::memcpy(&out[k], &gc.c[k], sizeof(GridCollider));
    return 0;
} MPM_CATCH_ALL

int mpm_finalize_external_contact_forces(mpm_handle_t e, float dt, float* tau_out, float* f_out) try {
    READY(e);
    REQUIRE(dt > 0.f, "dt must be positive");
    if (int rc = mpm_external_body_force_to_host(e, tau_out, f_out)) return rc;
    // impulses accumulated over the substeps of one plant step -> forces (deformable_driver.h:214-217)
    for (size_t i = 0; i < e->cb.n_bodies * 3; ++i) {
        if (tau_out) tau_out[i] /= dt;
        if (f_out) f_out[i] /= dt;
    }
    return 0;
} MPM_CATCH_ALL

int mpm_spatial_force_shift(size_t n, const float* tau, const float* f, const float* offset, float* tau_out) try {
    REQUIRE(n == 0 || (tau && f && offset && tau_out), "null argument");
    spatial_force_shift(n, tau, f, offset, tau_out);
    return 0;
} MPM_CATCH_ALL

int mpm_external_forces_at_body_origin(size_t n, const float* R_WB, const float* p_BoBq_B, const float* tau,
                                       const float* f, float* tau_Bo_out) try {
    REQUIRE(n == 0 || (R_WB && p_BoBq_B && tau && f && tau_Bo_out), "null argument");
    forces_at_body_origin(n, R_WB, p_BoBq_B, tau, f, tau_Bo_out);
    return 0;
} MPM_CATCH_ALL

int mpm_copy_contact_pairs(mpm_handle_t e, size_t n, const uint32_t* particle, const uint32_t* body, const float* dist,
                           const float* normal, const float* pos, const float* rigid_v, const float* rigid_p_WB) try {
    READY(e);
    REQUIRE(n == 0 || (particle && body && dist && normal && pos && rigid_v && rigid_p_WB), "null contact array");
    return copy_contacts(e, n, particle, body, dist, normal, pos, rigid_v, rigid_p_WB);
} MPM_CATCH_ALL

int mpm_generate_contact_pairs(mpm_handle_t e, size_t n_colliders, const mpm_collider_t* colliders, size_t* n_out) try {
    READY(e);
    REQUIRE(n_colliders == 0 || colliders, "null collider array");
    REQUIRE(n_colliders <= 1024, "too many colliders");
    return generate_contacts(e, n_colliders, colliders, n_out);
} MPM_CATCH_ALL

int mpm_download_contact_pairs(mpm_handle_t e, uint32_t* particle, uint32_t* body, float* dist, float* normal,
                               float* pos, float* rigid_v, float* p_WB) try {
    READY(e);
    return download_contacts(e, particle, body, dist, normal, pos, rigid_v, p_WB);
} MPM_CATCH_ALL

int mpm_last_contact_counts(mpm_handle_t e, uint32_t* contacts_out, uint32_t* nodes_out, int* setup_reused_out) try {
    REQUIRE(e, "null handle");
    if (contacts_out) *contacts_out = e->last_contact.contacts;
    if (nodes_out) *nodes_out = e->last_contact.nodes;
    if (setup_reused_out) *setup_reused_out = e->last_contact_reused ? 1 : 0;
    return 0;
} MPM_CATCH_ALL

int mpm_debug_contact_counters(mpm_handle_t e, uint64_t out4[6]) try {
    REQUIRE(e && out4, "null argument");
    for (int k = 0; k < 6; ++k) out4[k] = e->ct_counters[k];
    return 0;
} MPM_CATCH_ALL

// Tests of the invariant "no kernel indexes a per-pair array with a count it has not clamped": overwrites the pair count
// that mpm_generate_contact_pairs left on the device (count >= 0) and / or moves the stamp of the generation that wrote
// it (stamp_delta != 0: the count of "another" generation, i.e. a stale one).  The next mpm_update_contact must refuse.
int mpm_debug_contact_count(mpm_handle_t e, int count, int stamp_delta) try {
    READY(e);
    REQUIRE(e->cb.st && e->cb.dev_counted, "mpm_debug_contact_count: needs pairs counted on the device (mpm_generate_contact_pairs without a count)");
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (count >= 0) H2D(e, &e->cb.st->n, &count, sizeof(int));
    if (stamp_delta) {
        unsigned stamp = 0;
        D2H(e, &stamp, &e->cb.st->n_stamp, sizeof(unsigned));
        stamp += (unsigned)stamp_delta;
        H2D(e, &e->cb.st->n_stamp, &stamp, sizeof(unsigned));
    }
    return 0;
} MPM_CATCH_ALL

int mpm_get_contact_pair_count(mpm_handle_t e, size_t* n_out) try {
    READY(e);
    REQUIRE(n_out, "null output");
    if (int rc = resolve_contact_count(e)) return rc;
    *n_out = e->cb.n;
    return 0;
} MPM_CATCH_ALL

int mpm_download_contact_log(mpm_handle_t e, float* rows_out, size_t capacity_rows, size_t* n_rows_out) try {
    READY(e);
    REQUIRE(n_rows_out, "null output");
    static_assert(CT_LOG_F == MPM_CONTACT_LOG_FLOATS, "contact log layouts differ");
    const size_t rows = (size_t)std::min(std::max(e->cb.last_iters, 0), CT_LOG);
    *n_rows_out = rows;
    if (!rows_out || rows == 0) return 0;
    REQUIRE(capacity_rows >= rows, "output array too small");
    REQUIRE(e->cb.it_log, "no contact solve has run");
    D2H(e, rows_out, e->cb.it_log, rows * CT_LOG_F * sizeof(float));
    return 0;
} MPM_CATCH_ALL

int mpm_get_contact_stats(mpm_handle_t e, mpm_contact_stats_t* out) try {
    READY(e);
    REQUIRE(out, "null stats");
    if (int rc = contact_stats_from_device(e)) return rc;
    *out = e->last_contact;
    return 0;
} MPM_CATCH_ALL

int mpm_update_contact(mpm_handle_t e, int frame, int substep, float dt, float mu, float stiffness, float damping,
                       int dump, int exact, int max_iters, int* iters_out, float* residual_out) try {
    READY(e);
    if (iters_out) *iters_out = 0;
    if (residual_out) *residual_out = 0.f;
    e->last_contact = mpm_contact_stats_t{};
    // (a rank of a TEAM takes part in the solve whatever its own pair count: "no contacts" is decided by all ranks together)
    if (!e->cb.dev_counted && e->cb.n == 0 && !(e->dp.dist.on && e->team.on)) return 0;  // cuda_mpm_solver.cu:216-217
    REQUIRE(e->grid_state == 2, "UpdateContact before UpdateGrid");
    return update_contact(e, frame, substep, dt, mu, stiffness, damping, dump, exact, max_iters, iters_out,
                          residual_out);
} MPM_CATCH_ALL

// The body of DeformableDriver::CalcAbstractStates' substep loop (multibody/plant/deformable_driver.h:240-258) for
// rigid bodies with analytic signed distance fields, n times, in one call: RebuildMapping, CalcFemStateAndForce,
// ParticleToGrid, UpdateGrid, CalcMpmContactPairs + CopyContactPairs (on the device), UpdateContact, GridToParticle.
// What the seven calls per substep cost a caller on top of the kernels is the host's share: ~45 launches per coupled
// substep at ~3.5 us each, issued in bursts behind every point where the host has to wait (the solve's end), with the
// caller's own code in between.  Here the host waits exactly once per substep -- for the mailbox word that says the solve
// has converged -- and has the next substep's long kernels (FEM, ParticleToGrid) enqueued before the short ones of the
// contact set-up are due; between two substeps of the call GridToParticle and ParticleToGrid leave out what only a
// download would read (DP::lean_g2p), as mpm_run_substeps does.
// Coupled substeps of a PARTITIONED domain (BASELINE config 5's path): the same loop body on every
// local rank of the team, enqueued phase by phase across them -- halo of the raw node sums over the DIRECT transport, grid
// update, pairs of the particles each rank owns (counted on the device), the TEAM solve (mpm_team.h: zone exchange of the
// per-node Hessian / gradient sums with the neighbours and rank-ordered sums of the line-search rows, all on the engines'
// streams), GridToParticle, impulses (per-rank partial sums).  The host waits once per substep, for the mailbox.  No
// migration in here: the caller runs the substeps between two migrations in one call (DomainChain / LocalWorld decide when
// one is due); no contact-free speculation either -- "does any rank have a pair" is part of the solve's first exchange.
static int team_coupled_substeps(const std::vector<mpm_engine*>& L, int n, const mpm_coupled_params_t* prm, size_t n_colliders,
                                 const mpm_collider_t* colliders, mpm_coupled_result_t* const* results) {
    const float dt = prm->dt;
    for (mpm_engine* e : L) {
        REQUIRE(e->dp.dist.on && e->team.on && e->chain.direct,
                "coupled substeps on a partitioned domain need the direct halo (mpm_chain_direct_connect) and the team transport (mpm_team_connect)");
        REQUIRE(e->chain.pitch == 0, "coupled substeps: a partitioned domain has pitch 0");
    }
    GridColliders gc;
    if (int rc = grid_colliders_for(L[0], prm->mpm_bc, &gc)) return rc;
    for (int s = 0; s < n; ++s) {
        for (mpm_engine* e : L) {
            may_resort(e, dt);
            e->chain_lean = 0;
            if (int rc = chain_direct_begin(e, dt)) return rc;
        }
        for (mpm_engine* e : L)
            if (int rc = chain_direct_end(e, dt, prm->mpm_bc, false)) return rc;
        for (mpm_engine* e : L)
            if (int rc = generate_contacts(e, n_colliders, colliders, nullptr)) return rc;
        std::vector<SolveOutcome> ocs;
        auto g2p = [&](size_t i) { launch_g2p(L[i], dt); };
        if (int rc = team_solve(L, dt, prm->friction_mu, prm->stiffness, prm->damping, prm->exact_line_search, prm->max_newton_iterations,
                                g2p, &ocs))
            return rc;
        for (size_t i = 0; i < L.size(); ++i) {
            mpm_engine* e = L[i];
            e->substeps += 1;
            e->grid_state = 2;
            if (results && results[i]) {
                mpm_coupled_result_t& r = results[i][s];
                r.iterations = ocs[i].mb.iters;
                r.contacts = e->last_contact.contacts;
                r.nodes = e->last_contact.nodes;
                r.residual = ocs[i].mb.residual;
                r.setup_reused = 0;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The ranks of ONE partition that live in this process (an in-process world: tests and rehearsals with more ranks than a
// box admits processes), all on one stream: n coupled substeps of the whole world in one call.  results: n_local arrays of
// n entries, or NULL.
int mpm_world_coupled_substeps(mpm_handle_t* handles, int n_local, int n, const mpm_coupled_params_t* prm, size_t n_colliders,
                               const mpm_collider_t* colliders, mpm_coupled_result_t* const* results) try {
    REQUIRE(handles && n_local >= 1 && n_local <= TEAM_MAX && prm && n >= 0, "bad arguments");
    REQUIRE(n_colliders > 0 && colliders && n_colliders <= 1024, "bad collider array");
    std::vector<mpm_engine*> L(handles, handles + n_local);
    for (mpm_engine* e : L) {
        READY(e);
        REQUIRE(e->stream == L[0]->stream, "an in-process world runs on ONE stream (mpm_set_stream)");
    }
    return team_coupled_substeps(L, n, prm, n_colliders, colliders, results);
} MPM_CATCH_ALL

int mpm_run_coupled_substeps(mpm_handle_t e, int n, const mpm_coupled_params_t* prm, size_t n_colliders,
                             const mpm_collider_t* colliders, mpm_coupled_result_t* results) try {
    READY(e);
    REQUIRE(prm && n >= 0, "bad arguments");
    REQUIRE(n_colliders == 0 || colliders, "null collider array");
    REQUIRE(n_colliders <= 1024, "too many colliders");
    if (e->dp.dist.on) {
        REQUIRE(n_colliders > 0, "coupled substeps on a partitioned domain need colliders (contact-free: mpm_chain_substeps)");
        mpm_coupled_result_t* one[1] = {results};
        return team_coupled_substeps({e}, n, prm, n_colliders, colliders, results ? one : nullptr);
    }
    if (n_colliders == 0) {   // nothing to couple with: contact-free substeps
        if (results)
            for (int s = 0; s < n; ++s) results[s] = mpm_coupled_result_t{};
        return mpm_run_substeps(e, n, prm->dt, prm->mpm_bc);
    }
    GridColliders gc;
    if (int rc = grid_colliders_for(e, prm->mpm_bc, &gc)) return rc;
    const float dt = prm->dt;
    // (MPM_CT_DEBUG: where the HOST's time of the call goes, per substep)
    using clk = std::chrono::steady_clock;
    double t_base = 0, t_gen = 0, t_solve = 0, t_tail = 0;
    auto since = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
    struct Report {
        mpm_engine* e; int n; double *a, *b, *c, *d;
        ~Report() {
            if (e->ct_debug && n > 0)
                std::fprintf(stderr, "[mpm_hip] coupled substeps, host us per substep: base launches %.1f, pairs %.1f, solve (set-up + "
                                     "iterations + waits: %.1f polling) %.1f, GridToParticle %.1f\n", *a / n, *b / n, e->ct_wait_us / n, *c / n, *d / n);
        }
    } report{e, n, &t_base, &t_gen, &t_solve, &t_tail};
    e->ct_wait_us = 0;
    struct HookReset {   // (however the function is left)
        mpm_engine* e;
        ~HookReset() { e->ct_before_impulse = nullptr; e->dp.lean_g2p = 0; e->dp.gated = 0; }
    } hook_reset{e};
    bool g2p_done = false;
    // Re-sort check launches (four kernels that return at once unless GridToParticle has raised need_rebuild): with every
    // substep that may not skip itself; none while the quiet time that the last re-sort estimated lasts -- the host learns
    // what is left of it from every solve's publication.  A substep that goes without them and finds a re-sort pending
    // skips itself as a whole (DP::gated: transfer kernels, contact solve, GridToParticle) and is run again, with the
    // re-sort in front: a wrong guess costs time, not correctness.
    bool force_check = true;
    // Contact-free stretches (round 5): when a coupled substep had no pairs, a watch kernel behind its GridToParticle asks
    // whether the NEXT one would have any (k_ct_watch: exact); the following substeps are then enqueued in chunks WITHOUT
    // pair generation, contact solve or any wait, each with its own watch behind it and the gate "skip yourself if a watch
    // since `watch_base` has seen a particle in a collider" (DP::gated bit 2).  After a chunk the host synchronises once and
    // reads how many of its substeps skipped themselves -- always the last ones: a hit is sticky -- and runs those as
    // coupled substeps.  A cloth that falls towards a body costs a contact-free substep plus the watch until it arrives.
    const bool may_watch = n_colliders > 0 && !e->ct_no_watch;
    bool spec = false, spec_check = true;
    float spec_quiet_left = 0.f;
    unsigned watch_base = 0;
    int chunk = 2;
    for (int s = 0; s < n; ++s) {
        if (spec) {
            const int m = std::min(chunk, n - s);
            // (the re-sort checks of a chunk: none while the quiet time lasts that the last look at the control block left
            // -- a substep that meets a pending re-sort then skips itself like one that meets a hit, and the chunk's tail is
            // repeated with the checks --, else with every substep)
            float quiet = spec_check ? 0.f : e->quiet_factor * spec_quiet_left;
            for (int q = 0; q < m; ++q) {
                const bool unchecked = quiet > 2.f * dt;
                quiet -= dt;
                may_resort(e, dt);
                e->dp.gated = 4 | (unchecked ? 1 : 0);
                e->dp.watch_base = watch_base;
                if (!unchecked) {
                    e->dp.lean_resort = 1;
                    launch_rebuild(e);
                    e->dp.lean_resort = 0;
                }
                e->dp.lean_g2p = s + q + 1 < n;
                launch_fem_p2g(e, dt);
                launch_grid(e, gc);
                launch_g2p(e, dt);
                launch_contact_watch(e, e->dp, ++e->watch_seq);
            }
            e->dp.gated = 0;
            e->dp.lean_g2p = 0;
            e->grid_state = 2;
            HIP_TRY(hipStreamSynchronize(e->stream));
            Ctl c;
            D2H(e, &c, e->dp.ctl, sizeof(Ctl));
            const int skipped = (int)std::min<unsigned>(c.skipped, (unsigned)m);
            if (c.skipped) {
                const unsigned zero = 0;
                H2D(e, &e->dp.ctl->skipped, &zero, sizeof(unsigned));
            }
            const int ran = m - skipped;
            const bool hit = (int)(c.watch_hit - watch_base) >= 0;
            e->ct_counters[4] += (uint64_t)m;
            e->ct_counters[5] += (uint64_t)skipped;
            for (int q = 0; q < ran; ++q) {
                e->substeps += 1;
                if (results) results[s + q] = mpm_coupled_result_t{};
            }
            if (ran > 0) {
                e->last_contact = mpm_contact_stats_t{};
                e->last_contact_reused = false;
            }
            spec_quiet_left = c.need_rebuild || c.error ? 0.f : std::max(0.f, c.quiet_time - c.time_since_resort);
            spec_check = skipped > 0;   // (whatever made them skip: the repeated ones carry their checks)
            // (an error flag ends the speculation too: the coupled substep that follows reports it where it always was)
            if (hit || c.error) {
                spec = false;
                chunk = 2;
                force_check = true;
            } else if (skipped > 0) {
                chunk = 2;           // a re-sort was pending: the tail of the chunk again, contact-free, with its checks
            } else {
                chunk = std::min(chunk * 2, 32);
            }
            s += ran - 1;   // (the loop's own increment makes it `ran`)
            continue;
        }
        auto t0 = clk::now();
        // (without colliders there is no solve whose publication would report a skipped substep: always checked)
        const bool gate = n_colliders > 0 && !force_check && (e->ct_gate_always || e->quiet_factor * e->ct_quiet_left > 2.f * dt);
        may_resort(e, dt);
        e->dp.gated = gate ? 1 : 0;
        if (!gate) {
            e->dp.lean_resort = 1;   // (CalcFemStateAndForce follows at once)
            launch_rebuild(e);
            e->dp.lean_resort = 0;
        }
        force_check = false;
        e->dp.lean_g2p = s + 1 < n;
        launch_fem_p2g(e, dt);
        launch_grid(e, gc);
        e->grid_state = 2;
        int iters = 0;
        float residual = 0.f;
        t_base += since(t0); t0 = clk::now();
        int rc = generate_contacts(e, n_colliders, colliders, nullptr);
        t_gen += since(t0); t0 = clk::now();
        g2p_done = false;
        e->ct_before_impulse = [&]() {
            launch_g2p(e, dt);
            g2p_done = true;
        };
        e->last_contact_gated = false;
        if (!rc && (e->cb.dev_counted || e->cb.n > 0))
            rc = update_contact(e, 0, s, dt, prm->friction_mu, prm->stiffness, prm->damping, 0, prm->exact_line_search,
                                prm->max_newton_iterations, &iters, &residual);
        else if (!rc)
            e->last_contact = mpm_contact_stats_t{};
        e->ct_before_impulse = nullptr;
        t_solve += since(t0); t0 = clk::now();
        if (rc) return rc;
        if (!g2p_done) launch_g2p(e, dt);
        e->dp.lean_g2p = 0;
        t_tail += since(t0);
        if (e->last_contact_gated) {
            // nothing of this substep has run (its GridToParticle counted it in Ctl::skipped, which belongs to
            // mpm_run_substeps' bookkeeping: cleared): once more, with the re-sort in front
            REQUIRE(gate, "contact solve: a substep with its re-sort launches reported itself as skipped");
            HIP_TRY(hipMemsetAsync(&e->dp.ctl->skipped, 0, sizeof(unsigned), e->stream));
            e->ct_counters[2] += 1;
            e->ct_quiet_left = 0.f;
            force_check = true;
            --s;
            continue;
        }
        e->substeps += 1;
        if (results) {
            mpm_coupled_result_t& r = results[s];
            r.iterations = iters;
            r.contacts = e->last_contact.contacts;
            r.nodes = e->last_contact.nodes;
            r.residual = residual;
            r.setup_reused = e->last_contact_reused ? 1 : 0;
        }
        // no pairs in this substep: ask whether the next one has any, and go on without pair generation while it has not
        spec = false;
        if (may_watch && s + 1 < n && e->last_contact.contacts == 0 && !e->cb.dev_counted) {
            watch_base = ++e->watch_seq;
            DP pw = e->dp;
            pw.gated = 0;
            launch_contact_watch(e, pw, watch_base);
            spec = true;
            spec_check = false;
            spec_quiet_left = e->ct_quiet_left;   // (what the solve's publication said is left of the quiet time)
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
} MPM_CATCH_ALL

// The root finder of the exact line search behind a C callback, for the known-answer tests
// (host code only: no device is touched).
int mpm_newton_bisect_f64(mpm_rootfind_fn fn, void* user, double x_lo, double x_hi, double guess, double x_tol,
                          double f_tol, int max_evals, int flags, double* root_out, int* evals_out) try {
    REQUIRE(fn && root_out && evals_out, "null argument");
    REQUIRE(x_lo < x_hi && x_lo <= guess && guess <= x_hi && x_tol > 0 && f_tol > 0, "bad bracket / tolerances");
    double f_lo, f_hi, d;
    fn(user, x_lo, &f_lo, &d);
    fn(user, x_hi, &f_hi, &d);
    RootFinder<double> rf;
    rf.start(x_lo, f_lo, x_hi, f_hi, guess, x_tol, f_tol, max_evals, flags);
    while (rf.status == 0) {
        double f, df;
        fn(user, rf.root, &f, &df);
        rf.feed(f, df);
    }
    *root_out = rf.root;
    *evals_out = rf.evals;
    return rf.status == 1 ? 0 : 1;
} MPM_CATCH_ALL

int mpm_newton_bisect_f32(mpm_rootfind_fn fn, void* user, float x_lo, float x_hi, float guess, float x_tol,
                          float f_tol, int max_evals, int flags, float* root_out, int* evals_out) try {
    REQUIRE(fn && root_out && evals_out, "null argument");
    REQUIRE(x_lo < x_hi && x_lo <= guess && guess <= x_hi && x_tol > 0 && f_tol > 0, "bad bracket / tolerances");
    double f_lo, f_hi, d;
    fn(user, (double)x_lo, &f_lo, &d);
    fn(user, (double)x_hi, &f_hi, &d);
    RootFinder<float> rf;
    rf.start(x_lo, (float)f_lo, x_hi, (float)f_hi, guess, x_tol, f_tol, max_evals, flags);
    while (rf.status == 0) {
        double f, df;
        fn(user, (double)rf.root, &f, &df);
        rf.feed((float)f, (float)df);
    }
    *root_out = rf.root;
    *evals_out = rf.evals;
    return rf.status == 1 ? 0 : 1;
} MPM_CATCH_ALL

}  // extern "C"
