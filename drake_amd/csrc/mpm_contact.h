// Host orchestration of the contact solve (CopyContactPairs / UpdateContact,
// cuda_mpm_solver.cu:193-621).
#pragma once
#include <memory>
#include <cmath>
#include <fstream>
#include <tuple>
#include <vector>

#include "../../include/mpm_hip.h"
#include "mpm_host.h"
#include "mpm_sort.h"
#include "mpm_rootfind.h"
#include "mpm_chain.h"

// (the poison switch of the engine whose buffers are being grown: set by the callers of grow() from their handle)
static thread_local bool g_grow_poison = false;
struct GrowPoison {   // (nests: generate_contacts -> ensure_contact_capacity; the outer setting comes back -- ADVICE r4)
    bool before;
    explicit GrowPoison(const mpm_engine* e) : before(g_grow_poison) { g_grow_poison = e->poison(); }
    ~GrowPoison() { g_grow_poison = before; }
};
template <class T>
static int grow(T** ptr, size_t n) {
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    HIP_TRY(hipMalloc((void**)ptr, std::max<size_t>(n, 1) * sizeof(T)));
    if (g_grow_poison) {  // debugging aid only: full device syncs around a null-stream fill
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(*ptr, 0xFF, std::max<size_t>(n, 1) * sizeof(T)));
        HIP_TRY(hipDeviceSynchronize());
    }
    return 0;
}

// GpuMpmState::ReallocateContacts (cuda_mpm_model.cu:267-317) + the uploads of CopyContactPairs
static int ensure_contact_capacity(mpm_engine* e, size_t n) {
    GrowPoison gp(e);
    ContactBuffers& b = e->cb;
    if (n > b.cap) {
        const size_t cap = n + n / 4;
        int rc;
        if ((rc = grow(&b.api_idx, cap)) || (rc = grow(&b.slot, cap)) || (rc = grow(&b.body, cap)) ||
            (rc = grow(&b.dist, cap)) ||
            (rc = grow(&b.normal, 3 * cap)) || (rc = grow(&b.pos, 3 * cap)) || (rc = grow(&b.rigid_v, 3 * cap)) ||
            (rc = grow(&b.p_WB, 3 * cap)) || (rc = grow(&b.vel, 3 * cap)) || (rc = grow(&b.vel0, 3 * cap)) ||
            (rc = grow(&b.key, cap)) || (rc = grow(&b.order, cap)) || (rc = grow(&b.key2, cap)) ||
            (rc = grow(&b.order2, cap)) || (rc = grow(&b.cnode, 27 * cap)) || (rc = grow(&b.cfx, 3 * cap)) ||
            (rc = grow(&b.cmass, cap)) || (rc = grow(&b.cphi0, cap)) || (rc = grow(&b.cR, 9 * cap)) ||
            (rc = grow(&b.cv0, 3 * cap)) || (rc = grow(&b.crv, 3 * cap)) || (rc = grow(&b.cvel, 3 * cap)) ||
            (rc = grow(&b.seg_part, cap * CT_SEG_F)))
            return rc;
        b.cap = cap;
    }
    {
        const size_t want = sort_hist_ints(n);
        if (want > b.cap_hist) {
            if (int rc = grow(&b.sort_hist, want)) return rc;
            b.cap_hist = want;
        }
    }
    const size_t cells = (size_t)e->dp.capA * 64;
    if (!b.run || b.cap_cells < cells) {
        int rc;
        if ((rc = grow(&b.run, cells)) || (rc = grow(&b.node_flag, cells)) || (rc = grow(&b.node_list, cells)) || (rc = grow(&b.flag_bits, (size_t)e->dp.capA)) ||
            (rc = grow(&b.node_runs, 27 * cells)) || (rc = grow(&b.gD, cells)) ||
            (e->dp.dist.on && (rc = grow(&b.hg, 3 * cells))) ||
            (rc = grow(&b.part, (size_t)(CT_ROWS_CON + CT_ROWS) * CT_PART)) || (rc = grow(&b.part_dir, (size_t)2 * CT_DIR_WG)) ||
            (rc = grow(&b.st, 1)) || (rc = grow(&b.it_log, (size_t)3 * CT_LOG)))
            return rc;
        b.cap_cells = cells;
    }
    return 0;
}

// GpuMpmState::ReallocateContacts (cuda_mpm_model.cu:267-317) + the uploads of CopyContactPairs
static int copy_contacts(mpm_engine* e, size_t n, const uint32_t* particle, const uint32_t* body, const float* dist,
                         const float* normal, const float* pos, const float* rigid_v, const float* p_WB) {
    ContactBuffers& b = e->cb;
    b.n = n;
    if (n == 0) return 0;
    for (size_t k = 0; k < n; ++k) REQUIRE(particle[k] < e->np, "contact particle index out of range");
    for (size_t k = 0; k < n; ++k) REQUIRE(body[k] < std::max<size_t>(b.n_bodies, 1), "contact body index out of range");
    if (int rc = ensure_contact_capacity(e, n)) return rc;
    HIP_TRY(hipMemcpyAsync(b.api_idx, particle, n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.body, body, n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.dist, dist, n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.normal, normal, n * 12, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.pos, pos, n * 12, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.rigid_v, rigid_v, n * 12, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(b.p_WB, p_WB, n * 12, hipMemcpyHostToDevice, e->stream));
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_ct_slots, dim3(g), dim3(256), 0, e->stream, (int)n, (const uint32_t*)b.api_idx,
                       e->d_pids_api, e->dp.imap, b.slot);
    ContactDev c{};
    c.n = (int)n;
    c.slot = b.slot;
    c.vel = b.vel;
    hipLaunchKernelGGL(k_ct_init_vel, dim3(g), dim3(256), 0, e->stream, e->dp, c);
    // (how many blocks are active: bounds the sort keys of the solve, see update_contact)
    HIP_TRY(hipMemcpyAsync(&b.n_active_hint, &e->dp.ctl->n_active, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // the host arrays may be released by the caller
    return 0;
}

// Device-side CalcMpmContactPairs + CopyContactPairs for analytic colliders (include/mpm_hip.h)
static int generate_contacts(mpm_engine* e, size_t n_col, const mpm_collider_t* cols, size_t* n_out) {
    GrowPoison gp(e);
    TraceRange tr("mpm:contact pairs (device)");
    static_assert(sizeof(Collider) == sizeof(mpm_collider_t), "collider layouts differ");
    ContactBuffers& b = e->cb;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    b.n = 0;
    if (n_out) *n_out = 0;
    if (n_col == 0) return 0;
    for (size_t j = 0; j < n_col; ++j) {
        REQUIRE(cols[j].kind >= 0 && cols[j].kind <= 3, "unknown collider kind");
        REQUIRE(cols[j].body < std::max<size_t>(b.n_bodies, 1), "collider body index out of range");
    }
    if (n_col > b.cap_colliders) {
        if (int rc = grow(&b.colliders, n_col)) return rc;
        b.cap_colliders = n_col;
    }
    const size_t np = e->np, padded = ((np + 1 + 4095) / 4096) * 4096;
    if (!b.gen_cnt) {
        int rc;
        if ((rc = grow(&b.gen_cnt, padded)) || (rc = grow(&b.gen_sums, padded / 4096 + 8))) return rc;
    }
    H2D(e, b.colliders, cols, n_col * sizeof(Collider));
    HIP_TRY(hipMemsetAsync(b.gen_cnt, 0, padded * 4, s));
    hipLaunchKernelGGL(k_ct_gen_count, dim3(e->g_np), dim3(256), 0, s, p, (const int*)e->d_pids_api, (int)n_col,
                       (const Collider*)b.colliders, b.gen_cnt);
    if (device_exclusive_scan(s, b.gen_cnt, np + 1, b.gen_sums)) return fail(MPM_ERR_HIP, "contact scan failed");
    // the one number the host needs: how many pairs (and, for the key width of the solve's sort -- see
    // update_contact -- how many blocks are active); gen_sums[0..1] is free again after the scan
    int two[2] = {0, 0};
    hipLaunchKernelGGL(k_pair_of_ints, dim3(1), dim3(1), 0, s, b.gen_sums, (const int*)(b.gen_cnt + np), (const unsigned*)&p.ctl->n_active);
    D2H(e, two, b.gen_sums, 8);
    const int total = two[0];
    b.n_active_hint = (unsigned)two[1];
    if (total <= 0) return 0;
    if (int rc = ensure_contact_capacity(e, (size_t)total)) return rc;
    ContactDev c{};
    c.n = total;
    c.slot = b.slot; c.body = b.body; c.dist = b.dist; c.normal = b.normal; c.pos = b.pos;
    c.rigid_v = b.rigid_v; c.p_WB = b.p_WB; c.vel = b.vel;
    hipLaunchKernelGGL(k_ct_gen_write, dim3(e->g_np), dim3(256), 0, s, p, (const int*)e->d_pids_api, (int)n_col,
                       (const Collider*)b.colliders, (const int*)b.gen_cnt, total, b.api_idx, c);
    HIP_TRY(hipGetLastError());
    b.n = (size_t)total;
    if (n_out) *n_out = (size_t)total;
    return 0;
}

static int download_contacts(mpm_engine* e, uint32_t* particle, uint32_t* body, float* dist, float* normal, float* pos,
                             float* rigid_v, float* p_WB) {
    const ContactBuffers& b = e->cb;
    const size_t n = b.n;
    if (n == 0) return 0;
    if (particle) D2H(e, particle, b.api_idx, n * 4);
    if (body) D2H(e, body, b.body, n * 4);
    if (dist) D2H(e, dist, b.dist, n * 4);
    if (normal) D2H(e, normal, b.normal, n * 12);
    if (pos) D2H(e, pos, b.pos, n * 12);
    if (rigid_v) D2H(e, rigid_v, b.rigid_v, n * 12);
    if (p_WB) D2H(e, p_WB, b.p_WB, n * 12);
    return 0;
}

static ContactDev make_contact_dev(mpm_engine* e, float dt, float mu, float k, float d, int max_iters) {
    ContactBuffers& b = e->cb;
    ContactDev c{};
    c.n = (int)b.n;
    c.max_iters = max_iters;
    c.dt = dt; c.mu = mu; c.k = k; c.d = d;
    c.epsv = e->mat.epsv;
    c.relax = 0.3f;   // jacobi_relax_coeff, cuda_mpm_solver.cu:239
    if (e->ct_relax > 0.f) c.relax = e->ct_relax;   // (tests: overshoot on purpose, so that the backtracking has work)
    c.tol = 1e-4f;    // kTol, cuda_mpm_solver.cu:236
    c.slot = b.slot; c.body = b.body; c.dist = b.dist; c.normal = b.normal; c.pos = b.pos;
    c.rigid_v = b.rigid_v; c.p_WB = b.p_WB; c.vel = b.vel; c.vel0 = b.vel0;
    c.key = b.key; c.order = b.order;
    c.cnode = b.cnode; c.cfx = b.cfx; c.cmass = b.cmass; c.cphi0 = b.cphi0; c.cR = b.cR; c.cv0 = b.cv0;
    c.crv = b.crv; c.cvel = b.cvel; c.seg_part = b.seg_part;
    c.run = b.run; c.node_flag = b.node_flag; c.node_list = b.node_list; c.flag_bits = b.flag_bits; c.node_runs = b.node_runs;
    c.cap_nodes = (int)b.cap_cells; c.gD = b.gD; c.hg = b.hg;
    c.part = b.part; c.part_dir = b.part_dir; c.st = b.st; c.it_log = b.it_log;
    c.body_tau = b.body_tau; c.body_f = b.body_f; c.n_bodies = (int)b.n_bodies;
    return c;
}

// ---- partitioned domain: transports of the distributed solve ----------------------------------
static size_t zone_buffer_bytes(size_t cap, int nv) { return (((4 + cap) * 4 + 15) / 16) * 16 + cap * 64 * nv * 16; }

// contact fields of the zone blocks to both neighbours and back: pack -> transport -> add
static int zone_exchange3(mpm_engine* e, float4* field) {
    ContactBuffers& b = e->cb;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    const mpm_engine::Chain& ch = e->chain;
    const size_t cap = ch.comm ? ch.cap : e->dist_zone_cap;
    const size_t bytes = zone_buffer_bytes(cap, 3);
    if (b.zone_bytes < bytes) {
        for (void*& q : b.zone_buf) {
            if (q) HIP_TRY(hipFree(q));
            q = nullptr;
            HIP_TRY(hipMalloc(&q, bytes));
            HIP_TRY(hipMemsetAsync(q, 0, bytes, s));
        }
        b.zone_bytes = bytes;
        b.zone_cap = cap;
    }
    const Dist& d = p.dist;
    const int Z = d.zone_cells / 4;
    // [0] = towards / from the left neighbour, [1] = right
    ZoneX zs{}, zr{};
    int n = 0, side[2];
    if (d.has_left) { zs.lo[n] = d.own_lo / 4 - Z; zs.hi[n] = d.own_lo / 4 + Z - 1; zs.buf[n] = (uint32_t*)b.zone_buf[0]; zr.buf[n] = (uint32_t*)b.zone_buf[2]; side[n++] = 0; }
    if (d.has_right) { zs.lo[n] = d.own_hi / 4 - Z; zs.hi[n] = d.own_hi / 4 + Z - 1; zs.buf[n] = (uint32_t*)b.zone_buf[1]; zr.buf[n] = (uint32_t*)b.zone_buf[3]; side[n++] = 1; }
    if (n == 0) return 0;
    (void)side;
    for (int k = 0; k < n; ++k) HIP_TRY(hipMemsetAsync(zs.buf[k], 0, 16, s));
    hipLaunchKernelGGL(k_zone_pack<3>, dim3(e->g_grid, n), dim3(256), 0, s, p, zs, (unsigned)cap, (const float4*)field);
    if (ch.comm) {
        const rccl_rt::Api* a = rccl_rt::api();
        RCCL_TRY(a->group_start());
        int rc_g = 0;
        if (!rc_g && ch.left >= 0) rc_g = a->send(b.zone_buf[0], bytes, 0, ch.left, ch.comm, s);
        if (!rc_g && ch.right >= 0) rc_g = a->send(b.zone_buf[1], bytes, 0, ch.right, ch.comm, s);
        if (!rc_g && ch.right >= 0) rc_g = a->recv(b.zone_buf[3], bytes, 0, ch.right, ch.comm, s);
        if (!rc_g && ch.left >= 0) rc_g = a->recv(b.zone_buf[2], bytes, 0, ch.left, ch.comm, s);
        const int rc_e = a->group_end();
        RCCL_TRY(rc_g);
        RCCL_TRY(rc_e);
    } else {
        REQUIRE(e->dist_exchange, "partitioned domain: no transport for the contact solve (mpm_chain_init or mpm_dist_set_transport)");
        HIP_TRY(hipStreamSynchronize(s));
        if (e->dist_exchange(e->dist_user, b.zone_buf[0], b.zone_buf[1], b.zone_buf[2], b.zone_buf[3], bytes))
            return fail(MPM_ERR_HIP, "the exchange callback of the distributed contact solve failed");
    }
    hipLaunchKernelGGL(k_zone_add<3>, dim3(64, n), dim3(256), 0, s, p, zr, (unsigned)cap, field);
    return 0;
}

// sum of n (<= 32) doubles at dev over all ranks, in place
static int dist_allreduce(mpm_engine* e, double* dev, size_t n) {
    const mpm_engine::Chain& ch = e->chain;
    if (ch.comm && rccl_rt::api()->all_reduce) {
        RCCL_TRY(rccl_rt::api()->all_reduce(dev, dev, n, 8 /* ncclFloat64 */, 0 /* ncclSum */, ch.comm, e->stream));
        return 0;
    }
    REQUIRE(e->dist_allreduce, "partitioned domain: no all-reduce for the contact solve (mpm_chain_init or mpm_dist_set_transport)");
    double host[32];
    HIP_TRY(hipMemcpyAsync(host, dev, n * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->dist_allreduce(e->dist_user, host, n)) return fail(MPM_ERR_HIP, "the all-reduce callback of the distributed contact solve failed");
    HIP_TRY(hipMemcpyAsync(dev, host, n * 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// The batched Newton loops: `pattern(i)` enqueues the launches of one iteration (kernels of an iteration
// that starts after convergence return at once).  The solver state is read back after every batch, into
// pinned memory, and the NEXT batch is enqueued before the host waits for that read-back: the device
// never idles while the host finds out whether the solve has finished (a stream synchronisation plus
// relaunch cost ~40 us per batch); the price is one batch of idle launches at the end.
// (batch sizes: MPM_CT_BATCH="first,next" overrides the defaults, for measurements)
static int ct_batch(const mpm_engine* e, int which, int dflt) { return e->ct_batch[which] > 0 ? e->ct_batch[which] : dflt; }

template <class Pattern>
static int run_batches(mpm_engine* e, Pattern&& pattern, int first_batch, int batch, int max_iters, ContactState* out,
                       int max_patterns = 1 << 30) {
    ContactBuffers& b = e->cb;
    hipStream_t s = e->stream;
    for (int i = 0; i < 2; ++i) {
        if (!b.h_st[i]) HIP_TRY(hipHostMalloc((void**)&b.h_st[i], sizeof(ContactState), hipHostMallocDefault));
        if (!b.h_ev[i]) HIP_TRY(hipEventCreateWithFlags(&b.h_ev[i], hipEventDisableTiming));
    }
    int launched = 0, slot = 0;
    auto enqueue = [&](int count) -> int {
        for (int q = 0; q < count; ++q) pattern(launched + q);
        launched += count;
        HIP_TRY(hipMemcpyAsync(b.h_st[slot], b.st, sizeof(ContactState), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(b.h_ev[slot], s));
        slot ^= 1;
        return 0;
    };
    if (int rc = enqueue(first_batch)) return rc;
    while (true) {
        if (int rc = enqueue(batch)) return rc;            // speculative: idle if the previous batch finished
        HIP_TRY(hipEventSynchronize(b.h_ev[slot]));        // (slot now names the older of the two)
        const ContactState& st = *b.h_st[slot];
        if (st.done || st.iters >= max_iters || launched > max_patterns) break;
    }
    HIP_TRY(hipEventSynchronize(b.h_ev[slot ^ 1]));        // the speculative batch has drained
    *out = *b.h_st[slot ^ 1];
    return 0;
}

// The four kernels of one backtracking Newton iteration, each launched `reps` times back to back on the state the last
// mpm_update_contact left (contacts sorted, per-cell runs, node list: all still valid), timed with one pair of events
// per kernel: duration per launch including the hand-over to the next launch, the same notion as a rocprofv3 kernel
// average.  The kernels run whatever the solver state says (ContactDev::force); velocities are not changed (no lazy
// update, no apply), the solver's scalars are scratch afterwards.
static int profile_contact_iteration(mpm_engine* e, int reps, float* kernel_ms) {
    ContactBuffers& b = e->cb;
    REQUIRE(b.n > 0 && b.last_iters > 0, "mpm_profile_contact_iteration needs a finished mpm_update_contact on this state");
    const mpm_contact_stats_t& lc = e->last_contact;
    (void)lc;
    ContactDev c = make_contact_dev(e, e->last_contact_dt, e->last_contact_mu, e->last_contact_k, e->last_contact_d, 1 << 30);
    c.force = 1;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    const size_t n = b.n;
    const int n_con_wg = (int)std::min<size_t>((n + CT_WG / 4 - 1) / (CT_WG / 4), CT_ROWS_CON);
    const int n_grid_wg = CT_ROWS, n_dir_wg = CT_DIR_WG;
    const unsigned n_tile_wg = (unsigned)std::min<size_t>((n + CT_TILE - 1) / CT_TILE, CT_TILE_WG);
    hipEvent_t ev[5];
    for (auto& x : ev) HIP_TRY(hipEventCreate(&x));
    auto phase = [&](int k) {
        for (int r = 0; r < reps; ++r) {
            if (k == 0) hipLaunchKernelGGL(k_ct_tile, dim3(n_tile_wg), dim3(256), 0, s, p, c, 0, 0);
            if (k == 1) hipLaunchKernelGGL(k_ct_node_dir<0>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, c, 0);
            if (k == 2) hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 0, 0.f);
            if (k == 3) hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 0, 0);
        }
    };
    phase(0);   // warm-up
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipEventRecord(ev[k], s));
        phase(k);
    }
    HIP_TRY(hipEventRecord(ev[4], s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 4; ++k) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        kernel_ms[k] = ms / (float)reps;
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    HIP_TRY(hipMemsetAsync(b.st, 0, sizeof(ContactState), s));
    return 0;
}

static int update_contact(mpm_engine* e, int frame, int substep, float dt, float mu, float stiffness, float damping,
                          int dump, int exact, int max_iters, int* iters_out, float* residual_out) {
    ContactBuffers& b = e->cb;
    const size_t n = b.n;
    if (max_iters <= 0) max_iters = 2000;  // cuda_mpm_solver.cu:234
    if (b.n_bodies == 0) {
        // the reference requires ReallocateExternelBodies first; size the accumulators to the ids in use
        return fail(MPM_ERR_INVALID, "call mpm_reallocate_external_bodies before mpm_update_contact");
    }
    ContactDev c = make_contact_dev(e, dt, mu, stiffness, damping, max_iters);
    e->last_contact_dt = dt; e->last_contact_mu = mu; e->last_contact_k = stiffness; e->last_contact_d = damping;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    const unsigned gc = (unsigned)((n + 255) / 256);
    // The pairs name particles by the caller's slot; the engine's internal slots change with every
    // re-sort (which the device decides by itself, e.g. inside a RebuildMapping between
    // CopyContactPairs and this call): k_ct_keys resolves them again, from the kept caller indices.
    // grid-stride contact part of k_ct_ls: 4 lanes per contact
    const int n_con_wg = (int)std::min<size_t>((n + CT_WG / 4 - 1) / (CT_WG / 4), CT_ROWS_CON);
    const int n_grid_wg = CT_ROWS;                               // grid-stride node part
    const int n_dir_wg = CT_DIR_WG;
    TraceRange tr_all("mpm:UpdateContact");
    // ---- set-up: contacts in base-cell order, per-cell runs, nodes that see contacts ---------
    std::unique_ptr<TraceRange> tr_phase(new TraceRange("mpm:UpdateContact set-up (sort, per-cell runs, node list)"));
    auto trace_phase = [&](const char* name) {   // (phases end where the next begins; early returns close them too)
        tr_phase.reset();
        tr_phase.reset(new TraceRange(name));
    };
    HIP_TRY(hipMemsetAsync(b.st, 0, sizeof(ContactState), s));
    hipLaunchKernelGGL(k_ct_keys, dim3(1024), dim3(256), 0, s, p, c, (const uint32_t*)b.api_idx, (const int*)e->d_pids_api, b.slot);
    {
        // Keys are compact cells (active block * 64 + cell).  The number of active blocks is known from
        // the hand-over of the pairs unless a re-sort may have run since (launch_rebuild clears the hint):
        // with config 3's 1805 blocks that is 18 key bits, two passes of 9, instead of 22 bits for the
        // table capacity, three passes of 8.
        const size_t blocks = b.n_active_hint ? std::min<size_t>(b.n_active_hint, p.capA) : p.capA;
        int bits = 1;
        while (((size_t)1 << bits) < blocks * 64) ++bits;
        // CT_NO_CELL has all those bits set and more: it sorts behind every real cell as long as
        // one more bit takes part
        if (radix_sort_pairs(s, b.key, b.order, b.key2, b.order2, b.sort_hist, n, std::min(bits + 1, 31)))
            return fail(MPM_ERR_HIP, "contact sort failed");
    }
    hipLaunchKernelGGL(k_ct_prepare, dim3(gc), dim3(256), 0, s, p, c);
    // (MPM_CT_FORCE_DIST=1, tests: a partitioned engine of ONE rank takes the distributed paths too -- split direction
    // kernels, sums through the all-reduce -- so that they can run on a box with one GPU)
    const bool dist = p.dist.on && (p.dist.world > 1 || e->ct_force_dist);
    if (dist) {
        // a node in a zone may be reached by the neighbour's contacts only: both ranks need it listed
        hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, c, 0);
        if (int rc = zone_exchange3(e, b.hg)) return rc;
        hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, c, 1);
    }
    hipLaunchKernelGGL(k_ct_flag_bits, dim3(256), dim3(256), 0, s, p, c);
    hipLaunchKernelGGL(k_ct_node_list, dim3(1), dim3(1024), 0, s, p, c);
    hipLaunchKernelGGL(k_ct_node_runs, dim3(1024), dim3(256), 0, s, p, c);
    // pre-contact velocity at the contact points (cuda_mpm_solver.cu:267-272)
    hipLaunchKernelGGL(k_ct_gather_vel, dim3(gc), dim3(256), 0, s, p, c, b.vel0);

    trace_phase("mpm:UpdateContact Newton iterations");
    std::vector<float> s_res, s_energy;
    std::vector<int> s_ls;
    float s_alpha_last = 0.f, s_E0_last = 0.f;
    ContactState st{};
    int iters = 0;
    float residual = 1e10f;
    const unsigned n_tile_wg = (unsigned)std::min<size_t>((n + CT_TILE - 1) / CT_TILE, CT_TILE_WG);
    // contact gradients/Hessians and their per-cell sums, then (H, G) and the direction per node
    auto launch_dir = [&](int first, int lazy) {
        hipLaunchKernelGGL(k_ct_tile, dim3(n_tile_wg), dim3(256), 0, s, p, c, first, lazy);
        hipLaunchKernelGGL(k_ct_node_dir<0>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, c, lazy);
    };
    auto newton_direction = [&](int first) -> int {
        if (!dist) {
            launch_dir(first, 0);
            return 0;
        }
        hipLaunchKernelGGL(k_ct_tile, dim3(n_tile_wg), dim3(256), 0, s, p, c, first, 0);
        hipLaunchKernelGGL(k_ct_node_dir<1>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, c, 0);
        if (int rc = zone_exchange3(e, b.hg)) return rc;
        hipLaunchKernelGGL(k_ct_node_dir<2>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, c, 0);
        return 0;
    };
    if (!exact && dist) {
        // one iteration at a time: every iteration has two exchanges with the other ranks
        while (true) {
            if (int rc = newton_direction(iters == 0)) return rc;
            hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 0, 0.f);
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 0, 1);
            if (int rc = dist_allreduce(e, b.st->red, 32)) return rc;
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 0, 2);
            hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c);
            iters += 1;
            HIP_TRY(hipMemcpyAsync(&st, b.st, sizeof(ContactState), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (st.done || st.iters >= max_iters) break;   // (the same on every rank: decided from the global sums)
        }
        iters = st.iters;
        residual = st.residual;
    } else if (!exact) {
        // device-resident loop: iterations are launched in batches, kernels of an iteration
        // that starts after convergence return immediately
        // (as many as the previous solve needed plus one, then a few at a time: iterations that start
        // after convergence are idle launches of ~2 us each)
        // Four launches per iteration: contacts -> sums per cell, nodes -> direction, energies of the
        // candidate steps, decision.  The accepted step reaches the grid velocity at the start of the
        // next iteration (k_ct_node_dir, lazy; k_ct_tile reads v - alpha D meanwhile) and, for the last
        // iteration, in the k_ct_apply after the loop.  MPM_CT_EAGER=1 keeps the separate k_ct_apply.
        const bool eager = e->ct_eager;   // (per handle, read at mpm_create)
        auto pattern = [&](int index) {
            launch_dir(index == 0, eager ? 0 : 1);
            hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 0, 0.f);
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 0, 0);
            if (eager) hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 0);
        };
        if (int rc = run_batches(e, pattern, ct_batch(e, 0, 2), ct_batch(e, 1, 2), max_iters, &st)) return rc;
        if (!eager) hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 2);
        iters = st.iters;
        residual = st.residual;
    } else if (!dist || (e->chain.comm && rccl_rt::api()->all_reduce)) {
        // Exact line search, device resident: a fixed pattern of launches per Newton iteration --
        // direction, PROBES x (energies at st->alpha_probe, one-thread state machine = root finder of
        // mpm_rootfind.h), apply, close -- whose kernels skip themselves according to the state: a search
        // that needs more probes continues in the next pattern (direction skipped), patterns after
        // convergence are idle.  One read-back per batch instead of one per probe.
        // Partitioned domain with the native chain: the same pattern; the sums of a probe go through ncclAllReduce ON
        // THE ENGINE'S STREAM between the two halves of the decision (k_ct_decide phase 1: this rank's sums; phase 2:
        // the state machine, on the global sums -- identical on every rank, so all ranks walk the same states and
        // every collective of the pattern is entered by all of them), the direction through the zone exchange.  No
        // host loop per probe: the root finder is inherently sequential (every probe is placed by the one before), so
        // one all-reduce per probe is the floor; what goes is the host round trip around each of them.
        const int PROBES = 6;
        int rc_pattern = 0;
        auto pattern = [&](int index) {
            if (!dist) {
                launch_dir(index == 0, 0);
            } else if (int rc = newton_direction(index == 0)) {
                rc_pattern = rc;
            }
            for (int k = 0; k < PROBES; ++k) {
                hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 2, 0.f);
                if (!dist) {
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 0);
                } else {
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 1);
                    if (int rc = dist_allreduce(e, b.st->red, 32)) rc_pattern = rc;
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 2);
                }
            }
            hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 1);
            hipLaunchKernelGGL(k_ct_exact_finish, dim3(1), dim3(64), 0, s, c);
        };
        // (a search that never terminates cannot happen -- the root finder stops after 200 evaluations, which in
        // float it often needs: |dx| < 3e-9 is out of reach -- but the launches are bounded too: every Newton
        // iteration may take 200 / PROBES + 1 patterns)
        if (int rc = run_batches(e, pattern, ct_batch(e, 0, 1), ct_batch(e, 1, 1), max_iters, &st,
                                 (200 / PROBES + 2) * max_iters + 64))
            return rc;
        if (rc_pattern) return rc_pattern;
        iters = st.iters;
        residual = st.residual;
        s_alpha_last = st.alpha;
        s_E0_last = st.E0;
        if (iters > 0) {
            std::vector<float> log((size_t)3 * std::min(iters, CT_LOG));
            HIP_TRY(hipMemcpyAsync(log.data(), b.it_log, log.size() * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (int i = 0; i < std::min(iters, CT_LOG); ++i) {
                s_res.push_back(log[i * 3]);
                s_ls.push_back((int)log[i * 3 + 1]);
                s_energy.push_back(log[i * 3 + 2]);
            }
        }
    } else {
        // exact line search on a partitioned domain whose transport is a pair of host callbacks
        // (mpm_dist_set_transport): every all-reduce is a host round trip anyway, so the search is driven from the
        // host, probe by probe, exactly like cuda_mpm_solver.cu:383-471 (itself a clone of Drake's
        // DoNewtonWithBisectionFallback)
        auto probe = [&](float alpha, std::tuple<float, float, float>* out) -> int {
            hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 1, alpha);
            if (dist) {
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 1, 1);
                if (int rc2 = dist_allreduce(e, b.st->red, 32)) return rc2;
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 1, 2);
            } else {
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 1);
            }
            HIP_TRY(hipMemcpyAsync(&st, b.st, sizeof(ContactState), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            *out = std::make_tuple((float)st.scal[0], (float)st.scal[1], (float)st.scal[2]);
            return 0;
        };
        const float f_tol = 1e-8f, x_tol = f_tol * c.relax;   // cuda_mpm_solver.cu:383-385
        while (residual > c.tol && iters < max_iters) {
            int rc;
            if ((rc = newton_direction(iters == 0))) return rc;
            std::tuple<float, float, float> f_lo, f_hi, f_root;
            if ((rc = probe(0.f, &f_lo)) || (rc = probe(1.f, &f_hi))) return rc;
            s_E0_last = std::get<0>(f_lo);   // E(alpha = 0)
            float x_lo = 0.f;
            if (std::get<1>(f_lo) < 0.f && std::get<1>(f_hi) < 0.f) {   // :395-398
                x_lo = 1.f;
                f_lo = f_hi;
            }
            // Newton with bisection fallback on dE/dalpha, starting at the upper end (:399), with the
            // clone's three deviations from Drake's routine (mpm_rootfind.h)
            RootFinder<float> rf;
            rf.start(x_lo, std::get<1>(f_lo), 1.f, std::get<1>(f_hi), 1.f, x_tol, f_tol, 200,
                     RF_SIGN3 | RF_NO_ENDS | RF_STEP_LAST);
            float energy = 0.f;
            while (rf.status == 0) {
                if ((rc = probe(rf.root, &f_root))) return rc;
                energy = std::get<0>(f_root);
                rf.feed(std::get<1>(f_root), std::get<2>(f_root));
            }
            const float alpha = rf.root;
            const int ls = rf.evals;
            s_alpha_last = alpha;
            HIP_TRY(hipMemcpyAsync(&b.st->alpha, &alpha, 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c);
            residual = std::sqrt(st.norm_dir_sq) / st.dofs;
            iters += 1;
            s_res.push_back(residual);
            s_energy.push_back(energy);
            s_ls.push_back(ls);
        }
    }
    // contact velocities after the solve and the reaction on the rigid bodies
    trace_phase("mpm:UpdateContact impulses");
    hipLaunchKernelGGL(k_ct_impulse, dim3(std::min(gc, 256u)), dim3(256), 0, s, p, c);
    HIP_TRY(hipGetLastError());
    b.last_iters = iters;
    if (e->ct_debug) fprintf(stderr, "contact solve: n %zu nodes %d items %d iters %d\n", n, st.n_nodes, 0, iters);
    if (iters_out) *iters_out = iters;
    if (residual_out) *residual_out = residual;
    {
        // what the reference prints / dumps per substep (cuda_mpm_solver.cu:577-612), kept for mpm_get_contact_stats
        mpm_contact_stats_t& cs = e->last_contact;
        cs.iterations = iters;
        cs.contacts = (uint32_t)n;
        cs.nodes = (uint32_t)st.n_nodes;
        cs.residual = residual;
        cs.norm_dir_sq = st.norm_dir_sq;
        cs.dofs = st.dofs;
        if (exact) {
            cs.line_search_evals = 0;
            for (int v : s_ls) cs.line_search_evals += v;
            cs.alpha = s_alpha_last;
            cs.energy = s_energy.empty() ? 0.f : s_energy.back();
            cs.E0 = s_E0_last;
        } else {
            cs.line_search_evals = st.ls_total;
            cs.alpha = st.alpha;
            cs.energy = st.energy;
            cs.E0 = st.E0;
        }
    }
    if (dump) {
        // per-substep statistics, same fields as cuda_mpm_solver.cu:587-612
        const std::string fn = e->dump_dir + "/jacobi_iter_" + std::to_string(max_iters) + "_frame_" +
                               std::to_string(frame) + "_substep_" + std::to_string(substep) + ".json";
        std::ofstream f(fn);
        if (!f) return fail(MPM_ERR_INVALID, "cannot write " + fn);
        f << "[\n";
        if (exact) {
            for (size_t i = 0; i < s_res.size(); ++i)
                f << "  {\n      \"residual\": " << s_res[i] << ",\n      \"line_search_cnt\": " << s_ls[i]
                  << ",\n      \"energy\": " << s_energy[i] << "\n  }" << (i + 1 < s_res.size() ? "," : "") << "\n";
        } else {
            f << "  {\n      \"iterations\": " << iters << ",\n      \"residual\": " << residual
              << ",\n      \"line_search_cnt\": " << (iters ? (double)st.ls_total / iters : 0.0)
              << ",\n      \"energy\": " << st.energy << "\n  }\n";
        }
        f << "]\n";
    }
    return 0;
}
