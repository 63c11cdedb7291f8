// Host orchestration of the contact solve (CopyContactPairs / UpdateContact,
// cuda_mpm_solver.cu:193-621).
#pragma once
#include <chrono>
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
        const size_t pad = cap + CT_TILE;   // (k_ct_tile / k_ct_ls read a whole tile of keys before they know the count)
        int rc;
        if ((rc = grow(&b.api_idx, cap)) || (rc = grow(&b.slot, cap)) || (rc = grow(&b.body, cap)) ||
            (rc = grow(&b.dist, cap)) ||
            (rc = grow(&b.normal, 3 * cap)) || (rc = grow(&b.pos, 3 * cap)) || (rc = grow(&b.rigid_v, 3 * cap)) ||
            (rc = grow(&b.p_WB, 3 * cap)) || (rc = grow(&b.vel, 3 * cap)) || (rc = grow(&b.vel0, 3 * cap)) ||
            (rc = grow(&b.key, pad)) || (rc = grow(&b.order, pad)) || (rc = grow(&b.key2, pad)) ||
            (rc = grow(&b.order2, pad)) || (rc = grow(&b.cnode, 27 * cap)) || (rc = grow(&b.cfx, 3 * cap)) ||
            (rc = grow(&b.cmass, cap)) || (rc = grow(&b.cphi0, cap)) || (rc = grow(&b.cR, 9 * cap)) ||
            (rc = grow(&b.cv0, 3 * cap)) || (rc = grow(&b.crv, 3 * cap)) || (rc = grow(&b.cvel, 3 * cap)) ||
            (rc = grow(&b.seg_part, cap * CT_SEG_F)) || (rc = grow(&b.prev_key, cap)) || (rc = grow(&b.prev_api, cap)) ||
            (rc = grow(&b.prev_body, cap)))
            return rc;
        // (keys beyond the count are read speculatively and then ignored: they only have to be there)
        for (uint32_t* q : {b.key, b.key2}) HIP_TRY(hipMemsetAsync(q, 0xFF, pad * 4, e->stream));
        b.cap = cap;
        b.last_unchanged = false;   // (prev_* are new: the next solve compares against nothing)
        if (b.st) {
            const int none = -1;
            HIP_TRY(hipMemcpyAsync(&b.st->prev_n, &none, 4, hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
        }
    }
    {
        // (sized for every launch bound n <= capacity: the count may live on the device, and a bound below 2^18 pairs
        // makes MORE tiles than the capacity itself when that lies above -- smaller tiles --: round 6, a latent overrun)
        const size_t want = sort_hist_ints_upto(b.cap);
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
            (rc = grow(&b.st, 1)) || (rc = grow(&b.it_log, (size_t)CT_LOG_F * CT_LOG)))
            return rc;
        HIP_TRY(hipMemsetAsync(b.st, 0, sizeof(ContactState), e->stream));
        {
            const int none = -1;   // (no previous solve)
            HIP_TRY(hipMemcpyAsync(&b.st->prev_n, &none, 4, hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
        }
        b.cap_cells = cells;
        b.last_unchanged = false;
    }
    if (!b.h_mbox) {
        // the mailbox the host polls (ContactMailbox): pinned, mapped, coherent -- the device's stores land without a copy
        HIP_TRY(hipHostMalloc((void**)&b.h_mbox, sizeof(ContactMailbox), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(b.h_mbox, 0, sizeof(ContactMailbox));
        HIP_TRY(hipHostGetDevicePointer((void**)&b.d_mbox, b.h_mbox, 0));
    }
    return 0;
}

// GpuMpmState::ReallocateContacts (cuda_mpm_model.cu:267-317) + the uploads of CopyContactPairs
static int copy_contacts(mpm_engine* e, size_t n, const uint32_t* particle, const uint32_t* body, const float* dist,
                         const float* normal, const float* pos, const float* rigid_v, const float* p_WB) {
    ContactBuffers& b = e->cb;
    b.n = n;
    b.dev_counted = false;
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
    HIP_TRY(hipStreamSynchronize(e->stream));  // the host arrays may be released by the caller
    return 0;
}

// The mailbox the kernels publish to (ContactMailbox): wait until publication `target` (or a later one) is there and
// return a consistent set of its four words.  Spins on host memory; every few thousand spins it asks the stream whether
// it has drained without the publication arriving (a kernel that faulted, a miscounted launch): an error, not a hang.
struct MailboxState {
    unsigned seq = 0;
    int done = 0, iters = 0;
    float residual = 0.f;
    unsigned count = 0, nodes = 0, n_active = 0;
    float quiet_left = 0.f;
    bool unchanged = false;
};
static int wait_mailbox_impl(mpm_engine* e, unsigned target, MailboxState* out);
static int wait_mailbox(mpm_engine* e, unsigned target, MailboxState* out) {
    if (!e->ct_debug) return wait_mailbox_impl(e, target, out);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = wait_mailbox_impl(e, target, out);
    e->ct_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
static int wait_mailbox_impl(mpm_engine* e, unsigned target, MailboxState* out) {
    ContactBuffers& b = e->cb;
    const volatile unsigned long long* w = b.h_mbox->w;
    unsigned spins = 0;
    bool drained = false;
    while (true) {
        const unsigned long long w0 = __atomic_load_n(&w[0], __ATOMIC_ACQUIRE);
        const unsigned seq = (unsigned)(w0 >> 32);
        if ((int)(seq - target) >= 0) {
            const unsigned long long w1 = __atomic_load_n(&w[1], __ATOMIC_ACQUIRE), w2 = __atomic_load_n(&w[2], __ATOMIC_ACQUIRE),
                                     w3 = __atomic_load_n(&w[3], __ATOMIC_ACQUIRE), w4 = __atomic_load_n(&w[4], __ATOMIC_ACQUIRE),
                                     w5 = __atomic_load_n(&w[5], __ATOMIC_ACQUIRE);
            if ((unsigned)(w1 >> 32) == seq && (unsigned)(w2 >> 32) == seq && (unsigned)(w3 >> 32) == seq &&
                (unsigned)(w4 >> 32) == seq && (unsigned)(w5 >> 32) == seq && __atomic_load_n(&w[0], __ATOMIC_ACQUIRE) == w0) {
                out->n_active = (unsigned)w4;
                const unsigned qb = (unsigned)w5;
                std::memcpy(&out->quiet_left, &qb, 4);
                out->seq = seq;
                out->done = (int)((w0 >> 24) & 0xFF);
                out->iters = (int)(w0 & 0xFFFFFF);
                const unsigned rb = (unsigned)w1;
                std::memcpy(&out->residual, &rb, 4);
                out->count = (unsigned)w2;
                out->nodes = (unsigned)w3 & 0x7FFFFFFFu;
                out->unchanged = ((unsigned)w3 >> 31) != 0;
                return 0;
            }
            continue;   // (a later publication is landing: its words arrive one by one)
        }
        if (drained) return fail(MPM_ERR_INTERNAL, "contact solve: the stream drained without the expected publication in the mailbox");
        if ((++spins & 0x3FFF) == 0) {
            const hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) drained = true;   // (one more look at the mailbox: the last store may still be landing)
            else if (q != hipErrorNotReady) return fail(MPM_ERR_HIP, std::string("contact solve: ") + hipGetErrorString(q));
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

// The pair count of the buffers, read back if the host does not know it yet (pairs made on the device without a
// read-back: mpm_generate_contact_pairs with n_contacts_out == NULL).  A synchronisation point.
static int generate_contacts_launch(mpm_engine* e);
static int resolve_contact_count(mpm_engine* e) {
    ContactBuffers& b = e->cb;
    for (int attempt = 0; b.dev_counted && attempt < 8; ++attempt) {
        int three[3] = {0, 0, 0};   // n, n_wanted, gen_fault
        D2H(e, three, &b.st->n, 12);
        if (!three[2]) {
            // (a count off the device: checked before anything is sized from it or indexed with it)
            if (three[0] < 0 || (size_t)three[0] > b.cap) {
                b.n = 0;
                b.dev_counted = false;
                return fail(MPM_ERR_CAPACITY, "contact pairs: the count on the device (" + std::to_string(three[0]) +
                                                  ") is outside the capacity of the pair buffers (" + std::to_string(b.cap) + ")");
            }
            b.n = (size_t)three[0];
            b.dev_counted = false;
            return 0;
        }
        // more pairs than the buffers hold: grow them and make the pairs again (nothing has used them yet)
        e->ct_counters[3] += 1;
        if (int rc = ensure_contact_capacity(e, (size_t)three[1])) return rc;
        if (int rc = generate_contacts_launch(e)) return rc;
    }
    REQUIRE(!b.dev_counted, "contact pairs: the buffers keep overflowing");
    return 0;
}

// Device-side CalcMpmContactPairs + CopyContactPairs for analytic colliders (include/mpm_hip.h).  Nothing here waits for
// the device: the colliders travel as a kernel argument, the pairs are counted, placed and LEFT COUNTED on the device
// (ContactState::n); mpm_update_contact's launches have fixed grids and read the count there.  Two launches: count per slot
// with the scan inside 4096-blocks, write (every workgroup adds up the block totals it needs; workgroup 0 leaves the count).
// the colliders of the last generate_contacts() as the kernels take them
static ColliderTable collider_table(const ContactBuffers& b) {
    const size_t n_col = b.last_colliders.size();
    ColliderTable tab{};
    tab.n = (int)n_col;
    if (n_col <= (size_t)CT_COLLIDER_ARGS) {
        for (size_t j = 0; j < n_col; ++j) tab.c[j] = b.last_colliders[j];
    } else {
        tab.dev = b.colliders;   // (uploaded by generate_contacts when they changed)
    }
    return tab;
}
// see k_ct_watch (mpm_run_coupled_substeps); `p` carries the gate of the substep in front of it
static void launch_contact_watch(mpm_engine* e, const DP& p, unsigned seq) {
    hipLaunchKernelGGL(k_ct_watch, dim3(1024), dim3(256), 0, e->stream, p, collider_table(e->cb), seq);
}
static int generate_contacts_launch(mpm_engine* e) {
    ContactBuffers& b = e->cb;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    const ColliderTable tab = collider_table(b);
    const size_t np = e->np, padded = ((np + 1 + 4095) / 4096) * 4096;
    const int nb = (int)(padded / 4096);
    hipLaunchKernelGGL(k_ct_gen_count_scan, dim3(nb), dim3(1024), 0, s, p, (const int*)e->d_pids_api, tab, b.gen_cnt, b.gen_sums);
    ContactDev c{};
    c.n = -1;
    c.st = b.st;
    c.slot = b.slot; c.body = b.body; c.dist = b.dist; c.normal = b.normal; c.pos = b.pos;
    c.rigid_v = b.rigid_v; c.p_WB = b.p_WB; c.vel = b.vel;
    b.gen_stamp += 1;   // (the solve that follows names this generation: k_ct_keys refuses a count another one left)
    hipLaunchKernelGGL(k_ct_gen_write, dim3(e->g_np), dim3(256), 0, s, p, (const int*)e->d_pids_api, tab, (const int*)b.gen_cnt,
                       (const int*)b.gen_sums, nb, (int)std::min<size_t>(b.cap, 0x7FFFFFFF), b.api_idx, c, b.gen_stamp);
    HIP_TRY(hipGetLastError());
    b.dev_counted = true;
    b.n = 0;
    // (the block tables are what they are now: the keys of the solve's sort are as wide as the active blocks need.
    // The host's last reading of that count, mpm_sync / mpm_get_stats, may be older than a re-sort: unknown then)
    return 0;
}

static int generate_contacts(mpm_engine* e, size_t n_col, const mpm_collider_t* cols, size_t* n_out) {
    GrowPoison gp(e);
    TraceRange tr("mpm:contact pairs (device)");
    static_assert(sizeof(Collider) == sizeof(mpm_collider_t), "collider layouts differ");
    ContactBuffers& b = e->cb;
    b.n = 0;
    b.dev_counted = false;
    if (n_out) *n_out = 0;
    if (n_col == 0) return 0;
    for (size_t j = 0; j < n_col; ++j) {
        REQUIRE(cols[j].kind >= 0 && cols[j].kind <= 3, "unknown collider kind");
        REQUIRE(cols[j].body < std::max<size_t>(b.n_bodies, 1), "collider body index out of range");
    }
    const Collider* in = reinterpret_cast<const Collider*>(cols);
    const bool same = b.last_colliders.size() == n_col && std::memcmp(b.last_colliders.data(), in, n_col * sizeof(Collider)) == 0;
    if (!same) b.last_colliders.assign(in, in + n_col);
    if (n_col > (size_t)CT_COLLIDER_ARGS) {
        if (n_col > b.cap_colliders) {
            if (int rc = grow(&b.colliders, n_col)) return rc;
            b.cap_colliders = n_col;
            H2D(e, b.colliders, cols, n_col * sizeof(Collider));
        } else if (!same) {
            H2D(e, b.colliders, cols, n_col * sizeof(Collider));
        }
    }
    const size_t np = e->np, padded = ((np + 1 + 4095) / 4096) * 4096;
    if (!b.gen_cnt) {
        int rc;
        if ((rc = grow(&b.gen_cnt, padded)) || (rc = grow(&b.gen_sums, padded / 4096 + 8))) return rc;
        HIP_TRY(hipMemsetAsync(b.gen_sums, 0, (padded / 4096 + 8) * 4, e->stream));
    }
    // a first capacity (the buffers grow when a scene needs more: resolve_contact_count / update_contact)
    if (int rc = ensure_contact_capacity(e, std::max<size_t>(b.cap, e->ct_initial_capacity ? e->ct_initial_capacity
                                                                                            : std::max<size_t>(4096, np / 16))))
        return rc;
    if (int rc = generate_contacts_launch(e)) return rc;
    if (n_out) {   // the caller wants the count now: a synchronisation point
        if (int rc = resolve_contact_count(e)) return rc;
        *n_out = b.n;
    }
    return 0;
}

static int download_contacts(mpm_engine* e, uint32_t* particle, uint32_t* body, float* dist, float* normal, float* pos,
                             float* rigid_v, float* p_WB) {
    if (int rc = resolve_contact_count(e)) return rc;
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

// `sorted`: the arrays the solve's kernels read after the sort (the sorted keys / order may sit in the second pair of
// buffers: radix_sort_pairs does not copy them back); otherwise the pair the set-up fills
static ContactDev make_contact_dev(mpm_engine* e, float dt, float mu, float k, float d, int max_iters, bool sorted = true) {
    ContactBuffers& b = e->cb;
    ContactDev c{};
    c.n = b.dev_counted ? -1 : (int)b.n;
    c.stride = (int)b.cap;
    c.max_iters = max_iters;
    c.dt = dt; c.mu = mu; c.k = k; c.d = d;
    c.epsv = e->mat.epsv;
    c.relax = 0.3f;   // jacobi_relax_coeff, cuda_mpm_solver.cu:239
    if (e->ct_relax > 0.f) c.relax = e->ct_relax;   // (tests: overshoot on purpose, so that the backtracking has work)
    c.tol = 1e-4f;    // kTol, cuda_mpm_solver.cu:236
    c.slot = b.slot; c.body = b.body; c.dist = b.dist; c.normal = b.normal; c.pos = b.pos;
    c.rigid_v = b.rigid_v; c.p_WB = b.p_WB; c.vel = b.vel; c.vel0 = b.vel0;
    const bool alt = sorted && b.sorted_in_alt;
    c.key = alt ? b.key2 : b.key;
    c.order = alt ? b.order2 : b.order;
    c.cnode = b.cnode; c.cfx = b.cfx; c.cmass = b.cmass; c.cphi0 = b.cphi0; c.cR = b.cR; c.cv0 = b.cv0;
    c.crv = b.crv; c.cvel = b.cvel; c.seg_part = b.seg_part;
    c.run = b.run; c.node_flag = b.node_flag; c.node_list = b.node_list; c.flag_bits = b.flag_bits; c.node_runs = b.node_runs;
    c.cap_nodes = (int)b.cap_cells; c.gD = b.gD; c.hg = b.hg;
    c.part = b.part; c.part_dir = b.part_dir; c.st = b.st; c.it_log = b.it_log;
    c.mbox = b.d_mbox;
    c.ctl = e->dp.ctl;
    c.prev_key = b.prev_key; c.prev_api = b.prev_api; c.prev_body = b.prev_body;
    c.body_acc = b.body_acc; c.n_bodies = (int)b.n_bodies;
    c.imp_fix = e->dp.fix_p;
    b.imp_unfix = e->dp.unfix_p;
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

// The batched Newton loops: `pattern(i)` enqueues the launches of one iteration (kernels of an iteration that starts
// after convergence return at once) and returns how many of them PUBLISH the solver state in the host-mapped mailbox
// (k_ct_decide, k_ct_exact_finish: ContactMailbox).  The host never copies the state back: it polls the mailbox for
// the last publication of the batch before the one it has just enqueued, so the device never idles while the host
// finds out whether the solve has finished (round 4 read the state back with a blit kernel per batch, ~4 us of the
// engine's stream each, 8 - 20 per solve); the price is one batch of idle launches at the end.
// (batch sizes: MPM_CT_BATCH="first,next" overrides the defaults, for measurements)
static int ct_batch(const mpm_engine* e, int which, int dflt) { return e->ct_batch[which] > 0 ? e->ct_batch[which] : dflt; }

// speculate = false: one pattern at a time, each waited for before the next is enqueued -- every rank of a partitioned
// domain then enqueues EXACTLY the same number of patterns (what it sees after pattern k does not depend on how fast its
// host polls), which patterns that contain a COLLECTIVE need: with speculation two ranks may see "finished" one
// publication apart and enqueue different numbers of ncclAllReduce calls (round 6: found by reading, the path had only
// ever run on a ring of one).
template <class Pattern>
static int run_batches(mpm_engine* e, Pattern&& pattern, int first_batch, int batch, int max_iters, MailboxState* out,
                       int max_patterns = 1 << 30, bool speculate = true) {
    ContactBuffers& b = e->cb;
    int launched = 0;
    auto enqueue = [&](int count) {
        for (int q = 0; q < count; ++q) b.published += (unsigned)pattern(launched + q);
        launched += count;
    };
    if (!speculate) {
        while (true) {
            enqueue(1);
            if (int rc = wait_mailbox(e, b.published, out)) return rc;
            if (out->done || out->iters >= max_iters || launched > max_patterns) return 0;
        }
    }
    enqueue(first_batch);
    while (true) {
        const unsigned older = b.published;                // the last publication of what is enqueued so far
        enqueue(batch);                                    // speculative: idle if the previous batch finished
        if (int rc = wait_mailbox(e, older, out)) return rc;
        if (out->done || out->iters >= max_iters || launched > max_patterns) break;
    }
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
    if (int rc = resolve_contact_count(e)) return rc;
    ContactDev c = make_contact_dev(e, e->last_contact_dt, e->last_contact_mu, e->last_contact_k, e->last_contact_d, 1 << 30);
    c.force = 1;
    c.mbox = nullptr;   // (nothing here is waited for through the mailbox)
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    const size_t n = b.n;
    const int n_con_wg = CT_ROWS_CON;   // (as the solve launches it)
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
    HIP_TRY(hipMemsetAsync(b.st, 0, offsetof(ContactState, n), s));
    return 0;
}

// What a finished (or refused) solve left for the host: the mailbox's last words
struct SolveOutcome {
    MailboxState mb;
    ContactState st;          // (only the paths that copy the state back fill it: partitioned domain)
    bool have_state = false;
};

static int solve_once(mpm_engine* e, float dt, float mu, float stiffness, float damping, int exact, int max_iters, bool full_setup,
                      SolveOutcome* oc, std::vector<float>* s_res, std::vector<int>* s_ls, std::vector<float>* s_energy,
                      float* s_alpha_last, float* s_E0_last);
static int contact_stats_from_device(mpm_engine* e);
static int team_solve(const std::vector<mpm_engine*>& L, float dt, float mu, float stiffness, float damping, int exact, int max_iters,
                      const std::function<void(size_t)>& before_impulse, std::vector<SolveOutcome>* ocs);

static int update_contact(mpm_engine* e, int frame, int substep, float dt, float mu, float stiffness, float damping,
                          int dump, int exact, int max_iters, int* iters_out, float* residual_out) {
    ContactBuffers& b = e->cb;
    if (max_iters <= 0) max_iters = 2000;  // cuda_mpm_solver.cu:234
    if (b.n_bodies == 0) {
        // the reference requires ReallocateExternelBodies first; size the accumulators to the ids in use
        return fail(MPM_ERR_INVALID, "call mpm_reallocate_external_bodies before mpm_update_contact");
    }
    if (e->dp.dist.on && e->team.on) {
        // partitioned domain with the TEAM transport: device resident, every rank takes part whatever its own pair count
        REQUIRE(!dump, "the JSON statistics dump is not available on a partitioned domain");
        std::vector<SolveOutcome> ocs;
        std::function<void(size_t)> hook;
        if (e->ct_before_impulse) hook = [e](size_t) { e->ct_before_impulse(); };
        if (int rc = team_solve({e}, dt, mu, stiffness, damping, exact, max_iters, hook, &ocs)) return rc;
        if (iters_out) *iters_out = ocs[0].mb.iters;
        if (residual_out) *residual_out = ocs[0].mb.residual;
        return 0;
    }
    e->last_contact_dt = dt; e->last_contact_mu = mu; e->last_contact_k = stiffness; e->last_contact_d = damping;
    const bool dist = e->dp.dist.on && (e->dp.dist.world > 1 || e->ct_force_dist);
    // (the distributed paths are host-driven per iteration anyway: they work with a count the host knows)
    if (dist)
        if (int rc = resolve_contact_count(e)) return rc;
    if (!b.dev_counted && b.n == 0) return 0;   // cuda_mpm_solver.cu:216-217
    TraceRange tr_all("mpm:UpdateContact");
    SolveOutcome oc;
    std::vector<float> s_res, s_energy;
    std::vector<int> s_ls;
    float s_alpha_last = 0.f, s_E0_last = 0.f;
    // A solve may REFUSE to run, in which case none of its kernels has touched the grid and it is repeated:
    //   CT_DONE_FAULT  pairs counted on the device did not fit the buffers: grow them, make the pairs again;
    //   CT_DONE_STALE  the set-up was enqueued on a guess that the device found wrong (the previous solve's sorted order
    //                  reused for a pair list that has changed; sort keys narrower than the active blocks need): full set-up.
    bool full_setup = false;
    for (int attempt = 0;; ++attempt) {
        REQUIRE(attempt < 8, "contact solve: the set-up keeps being refused");
        s_res.clear(); s_energy.clear(); s_ls.clear();
        if (int rc = solve_once(e, dt, mu, stiffness, damping, exact, max_iters, full_setup, &oc, &s_res, &s_ls, &s_energy,
                                &s_alpha_last, &s_E0_last))
            return rc;
        if (oc.mb.done == CT_DONE_CORRUPT) {
            // the count on the device is not one the solve may index with; nothing ran, nothing is repeated: the pairs
            // have to be made again (mpm_generate_contact_pairs / mpm_copy_contact_pairs) before the next solve
            const long long raw = (long long)(int)oc.mb.count;
            b.dev_counted = false;
            b.n = 0;
            b.last_unchanged = false;
            if (raw < 0 || (size_t)raw > b.cap)
                return fail(MPM_ERR_CAPACITY, "contact solve: the pair count on the device (" + std::to_string(raw) +
                                                  ") is outside the capacity that sized the per-pair buffers (" + std::to_string(b.cap) +
                                                  "): nothing was indexed with it, nothing was solved; make the pairs again");
            return fail(MPM_ERR_INTERNAL, "contact solve: the pair count on the device was not written by the pair generation this "
                                          "solve was enqueued for (a stale count): nothing was solved; make the pairs again");
        }
        if (oc.mb.done == CT_DONE_FAULT) {
            e->ct_counters[3] += 1;
            if (int rc = ensure_contact_capacity(e, (size_t)oc.mb.count)) return rc;
            if (int rc = generate_contacts_launch(e)) return rc;
            full_setup = true;
            continue;
        }
        if (oc.mb.done == CT_DONE_GATED) {
            // (mpm_run_coupled_substeps enqueued the substep without its re-sort launches and a re-sort was pending: nothing
            // of the substep has run; the caller repeats all of it)
            e->last_contact_gated = true;
            if (iters_out) *iters_out = 0;
            if (residual_out) *residual_out = 0.f;
            return 0;
        }
        if (oc.mb.done == CT_DONE_STALE) {
            e->ct_counters[2] += 1;
            b.n_active_hint = 0;
            full_setup = true;
            continue;
        }
        break;
    }
    const int iters = oc.mb.iters;
    const float residual = oc.mb.residual;
    e->ct_quiet_left = oc.mb.quiet_left;
    e->ct_counters[0] += 1;
    e->ct_counters[1] += e->last_contact_reused ? 1 : 0;
    if (b.dev_counted) {   // the count has arrived with the solve's first publication
        b.n = oc.mb.count;
        b.dev_counted = false;
    }
    b.last_iters = iters;
    b.last_unchanged = oc.mb.unchanged;
    if (e->ct_debug) fprintf(stderr, "contact solve: n %zu nodes %u iters %d unchanged %d\n", b.n, oc.mb.nodes, iters, (int)oc.mb.unchanged);
    if (iters_out) *iters_out = iters;
    if (residual_out) *residual_out = residual;
    {
        // what the reference prints / dumps per substep (cuda_mpm_solver.cu:577-612), kept for mpm_get_contact_stats: the
        // mailbox carries the iteration count, the residual, the contact and node counts; the other numbers stay on the
        // device until somebody asks (contact_stats_from_device: a synchronisation point of its own)
        mpm_contact_stats_t& cs = e->last_contact;
        cs = mpm_contact_stats_t{};
        cs.iterations = iters;
        cs.contacts = (uint32_t)b.n;
        cs.nodes = oc.mb.nodes;
        cs.residual = residual;
        e->last_contact_on_device = !oc.have_state;
        e->last_contact_exact = exact != 0;
        if (oc.have_state) {
            const ContactState& st = oc.st;
            cs.norm_dir_sq = st.norm_dir_sq;
            cs.dofs = st.dofs;
            cs.line_search_evals = st.ls_total;
            cs.alpha = st.alpha;
            cs.energy = st.energy;
            cs.E0 = st.E0;
            if (exact && !s_ls.empty()) {   // (host-driven exact search: its own bookkeeping)
                cs.line_search_evals = 0;
                for (int v : s_ls) cs.line_search_evals += v;
                cs.alpha = s_alpha_last;
                cs.energy = s_energy.empty() ? 0.f : s_energy.back();
                cs.E0 = s_E0_last;
            }
        }
    }
    if (dump) {
        // per-substep statistics, same fields as cuda_mpm_solver.cu:587-612
        if (int rc = contact_stats_from_device(e)) return rc;
        const mpm_contact_stats_t& cs = e->last_contact;
        if (exact && s_res.empty() && iters > 0) {
            std::vector<float> log((size_t)CT_LOG_F * std::min(iters, CT_LOG));
            D2H(e, log.data(), b.it_log, log.size() * 4);
            for (int i = 0; i < std::min(iters, CT_LOG); ++i) {
                s_res.push_back(log[(size_t)i * CT_LOG_F]);
                s_ls.push_back((int)log[(size_t)i * CT_LOG_F + 1]);
                s_energy.push_back(log[(size_t)i * CT_LOG_F + 2]);
            }
        }
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
              << ",\n      \"line_search_cnt\": " << (iters ? (double)cs.line_search_evals / iters : 0.0)
              << ",\n      \"energy\": " << cs.energy << "\n  }\n";
        }
        f << "]\n";
    }
    return 0;
}

// the numbers of the last solve that the mailbox does not carry (step, energies, |Dir|^2, DoFs, line-search count)
static int contact_stats_from_device(mpm_engine* e) {
    if (!e->last_contact_on_device) return 0;
    ContactBuffers& b = e->cb;
    ContactState st;
    D2H(e, &st, b.st, sizeof(ContactState));
    mpm_contact_stats_t& cs = e->last_contact;
    cs.norm_dir_sq = st.norm_dir_sq;
    cs.dofs = st.dofs;
    cs.line_search_evals = st.ls_total;
    cs.alpha = st.alpha;
    cs.energy = st.energy;
    cs.E0 = st.E0;
    e->last_contact_on_device = false;
    return 0;
}

static int solve_once(mpm_engine* e, float dt, float mu, float stiffness, float damping, int exact, int max_iters, bool full_setup,
                      SolveOutcome* oc, std::vector<float>* s_res_p, std::vector<int>* s_ls_p, std::vector<float>* s_energy_p,
                      float* s_alpha_last_p, float* s_E0_last_p) {
    ContactBuffers& b = e->cb;
    std::vector<float>&s_res = *s_res_p, &s_energy = *s_energy_p;
    std::vector<int>& s_ls = *s_ls_p;
    float &s_alpha_last = *s_alpha_last_p, &s_E0_last = *s_E0_last_p;
    *oc = SolveOutcome();
    e->last_contact_reused = false;
    e->last_contact_gated = false;
    const DP& p = e->dp;
    hipStream_t s = e->stream;
    // what sizes the grids: the count, or -- counted on the device -- a bound (every kernel strides over its grid and
    // reads the count itself): the last solve's count with some room, at most the buffers' capacity
    size_t n = b.n;
    if (b.dev_counted) n = b.n_hint && !full_setup ? std::min<size_t>(b.cap, b.n_hint + b.n_hint / 8 + 256) : b.cap;
    n = std::max<size_t>(n, 1);
    const unsigned gc = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    // The pairs name particles by the caller's slot; the engine's internal slots change with every
    // re-sort (which the device decides by itself, e.g. inside a RebuildMapping between
    // CopyContactPairs and this call): k_ct_keys resolves them again, from the kept caller indices.
    // grid-stride contact part of k_ct_ls: 4 lanes per contact
    // (always all CT_ROWS_CON rows: tile t goes to row t mod CT_ROWS_CON whatever the count or the buffers' capacity, so
    // the energy sums -- and with them every `E1 <= E0` decision -- do not depend on what sized the launch; workgroups
    // without a tile write a row of zeros)
    const int n_con_wg = CT_ROWS_CON;
    const int n_grid_wg = CT_ROWS;                               // grid-stride node part
    const int n_dir_wg = CT_DIR_WG;
    // (MPM_CT_FORCE_DIST=1, tests: a partitioned engine of ONE rank takes the distributed paths too -- split direction
    // kernels, sums through the all-reduce -- so that they can run on a box with one GPU)
    const bool dist = p.dist.on && (p.dist.world > 1 || e->ct_force_dist);
    // ---- set-up: contacts in base-cell order, per-cell runs, nodes that see contacts ---------
    std::unique_ptr<TraceRange> tr_phase(new TraceRange("mpm:UpdateContact set-up (sort, per-cell runs, node list)"));
    auto trace_phase = [&](const char* name) {   // (phases end where the next begins; early returns close them too)
        tr_phase.reset();
        tr_phase.reset(new TraceRange(name));
    };
    b.solves += 1;
    // Settled scenes: when the last solve found its pair list equal to its predecessor's, this one
    // is enqueued on the guess that it is again -- no sort, no per-cell runs, no node list: the last full set-up's stand
    // -- and k_ct_keys checks the guess on the device, entry by entry (same particle, body and base cell, same count, same
    // block tables).  A wrong guess refuses the solve (CT_DONE_STALE), the caller repeats it with the full set-up.
    const bool reuse = !full_setup && b.last_unchanged && !dist && !e->ct_no_reuse;
    // Keys are compact cells (active block * 64 + cell).  How many blocks are active the host knows from the last
    // solve's publication (it changes with re-sorts only); with config 3's 1805 blocks that is 18 key bits, two passes
    // of 9, instead of 22 bits for the table capacity, three passes of 8.  Checked on the device as well: keys wider
    // than the guess refuse the solve (CT_DONE_STALE).
    const size_t blocks = b.n_active_hint && !full_setup ? std::min<size_t>((size_t)b.n_active_hint * 2, p.capA) : p.capA;
    int bits = 1;
    while (((size_t)1 << bits) < blocks * 64) ++bits;
    {
        ContactDev c_in = make_contact_dev(e, dt, mu, stiffness, damping, max_iters, /* sorted = */ false);
        hipLaunchKernelGGL(k_ct_keys, dim3(1024), dim3(256), 0, s, p, c_in, (const uint32_t*)b.api_idx, (const int*)e->d_pids_api, b.slot,
                           b.published, b.solves, bits, (int)std::min<size_t>(n, 0x7FFFFFFF), reuse ? 1 : 0, full_setup ? 1 : 0, b.gen_stamp, 0);
    }
    if (!reuse) {
        // CT_NO_CELL has all those bits set and more: it sorts behind every real cell as long as
        // one more bit takes part
        // (a sort's tiles are fixed chunks, not a stride over the grid: with the count on the device its launch covers
        // the guess `n` -- the last solve's count with an eighth of room --, and k_ct_keys refuses the solve when the
        // count has outgrown it: a sheet of the stack landing doubles the count from one substep to the next, once)
        bool in_alt = false;
        if (radix_sort_pairs(s, b.key, b.order, b.key2, b.order2, b.sort_hist, n, std::min(bits + 1, 31), &in_alt,
                             b.dev_counted ? &b.st->n : nullptr))
            return fail(MPM_ERR_HIP, "contact sort failed");
        b.sorted_in_alt = in_alt;
    }
    ContactDev c = make_contact_dev(e, dt, mu, stiffness, damping, max_iters);
    if (reuse) {
        hipLaunchKernelGGL(k_ct_prepare<true>, dim3(gc), dim3(256), 0, s, p, c);
    } else {
        hipLaunchKernelGGL(k_ct_prepare<false>, dim3(gc), dim3(256), 0, s, p, c);
        if (dist) {
            // a node in a zone may be reached by the neighbour's contacts only: both ranks need it listed
            hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, c, 0);
            if (int rc = zone_exchange3(e, b.hg)) return rc;
            hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, c, 1);
        }
        hipLaunchKernelGGL(k_ct_flag_bits, dim3(256), dim3(256), 0, s, p, c);
        hipLaunchKernelGGL(k_ct_node_list, dim3((unsigned)((p.capA + 255) / 256)), dim3(256), 0, s, p, c);
        hipLaunchKernelGGL(k_ct_node_runs, dim3(1024), dim3(256), 0, s, p, c);
    }

    trace_phase("mpm:UpdateContact Newton iterations");
    ContactState st{};
    int iters = 0;
    float residual = 1e10f;
    const unsigned n_tile_wg = (unsigned)std::max<size_t>(1, std::min<size_t>((n + CT_TILE - 1) / CT_TILE, CT_TILE_WG));
    // the iterations are driven by the host, step by step, with the state copied back (partitioned domain without the
    // device-resident exact search): nothing is published, nothing is polled
    const bool host_driven = dist && !(exact && e->chain.comm && rccl_rt::api()->all_reduce);
    if (host_driven) c.mbox = nullptr;
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
        // Four launches per iteration: contacts -> sums per cell, nodes -> direction, energies of the
        // candidate steps, decision.  The accepted step reaches the grid velocity at the start of the
        // next iteration (k_ct_node_dir, lazy; k_ct_tile reads v - alpha D meanwhile) and, for the last
        // iteration, in the k_ct_apply after the loop.  MPM_CT_EAGER=1 keeps the separate k_ct_apply.
        // (the first iteration has no step to catch up with -- and, with a reused set-up, directions of the previous
        // solve in place: it never reads them)
        const bool eager = e->ct_eager;   // (per handle, read at mpm_create)
        auto pattern = [&](int index) -> int {
            launch_dir(index == 0, eager || index == 0 ? 0 : 1);
            hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 0, 0.f);
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 0, 0);
            if (eager) hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 0);
            return 1;
        };
        // two iterations first, then one at a time: an iteration is ~34 us of device time, the host needs ~20 to see a
        // publication and enqueue the next four launches, so one iteration in flight behind the one it waits for keeps
        // the device busy, and what is enqueued in vain when the solve converges is one iteration's idle launches.
        // (Round 5 first enqueued as many iterations as the LAST solve had taken: where the counts fall from solve to
        // solve -- 64, 26, 19, 17, 12 ... after an impact -- that was 5-10 idle patterns of 8 us per solve; measured
        // over four runs each, scratch/ct_batch_ab.sh: the time of a coupled substep beyond its iterations 0.32 -> 0.30
        // ms in the window of the impact, unchanged once settled.)
        const int first = ct_batch(e, 0, 2);
        if (int rc = run_batches(e, pattern, first, ct_batch(e, 1, 1), max_iters, &oc->mb)) return rc;
        if (!eager) hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 2);
        iters = oc->mb.iters;
        residual = oc->mb.residual;
    } else if (!host_driven) {
        // Exact line search, device resident: a fixed pattern of launches per Newton iteration --
        // direction, PROBES x (energies at st->alpha_probe, one-thread state machine = root finder of
        // mpm_rootfind.h), apply, close -- whose kernels skip themselves according to the state: a search
        // that needs more probes continues in the next pattern (direction skipped), patterns after
        // convergence are idle.  No read-back at all: the host polls the mailbox.
        // Partitioned domain with the native chain: the same pattern; the sums of a probe go through ncclAllReduce ON
        // THE ENGINE'S STREAM between the two halves of the decision (k_ct_decide phase 1: this rank's sums; phase 2:
        // the state machine, on the global sums -- identical on every rank, so all ranks walk the same states and
        // every collective of the pattern is entered by all of them), the direction through the zone exchange.  No
        // host loop per probe: the root finder is inherently sequential (every probe is placed by the one before), so
        // one all-reduce per probe is the floor; what goes is the host round trip around each of them.
        const int PROBES = 6;
        int rc_pattern = 0;
        auto pattern = [&](int index) -> int {
            int published = 0;
            if (!dist) {
                launch_dir(index == 0, 0);
            } else if (int rc = newton_direction(index == 0)) {
                rc_pattern = rc;
            }
            for (int k = 0; k < PROBES; ++k) {
                hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 2, 0.f);
                if (!dist) {
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 0);
                    published += 1;
                } else {
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 1);
                    if (int rc = dist_allreduce(e, b.st->red, 32)) rc_pattern = rc;
                    hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 2, 2);
                    published += 2;
                }
            }
            hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, s, p, c, 1);
            hipLaunchKernelGGL(k_ct_exact_finish, dim3(1), dim3(64), 0, s, c);
            return published + 1;
        };
        // (a search that never terminates cannot happen -- the root finder stops after 200 evaluations, which in
        // float it often needs: |dx| < 3e-9 is out of reach -- but the launches are bounded too: every Newton
        // iteration may take 200 / PROBES + 1 patterns)
        // (patterns with ncclAllReduce in them: every rank must enqueue the same number -- no speculation beyond one rank)
        const bool collective = dist && p.dist.world > 1;
        if (int rc = run_batches(e, pattern, ct_batch(e, 0, 1), ct_batch(e, 1, 1), max_iters, &oc->mb,
                                 (200 / PROBES + 2) * max_iters + 64, !collective))
            return rc;
        if (rc_pattern) return rc_pattern;
        iters = oc->mb.iters;
        residual = oc->mb.residual;
    } else {
        // exact line search on a partitioned domain whose transport is a pair of host callbacks
        // (mpm_dist_set_transport): every all-reduce is a host round trip anyway, so the search is driven from the
        // host, probe by probe, exactly like cuda_mpm_solver.cu:383-471 (itself a clone of Drake's
        // DoNewtonWithBisectionFallback)
        auto probe = [&](float alpha, std::tuple<float, float, float>* out) -> int {
            hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, s, p, c, n_con_wg, 1, alpha);
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 1, 1);
            if (int rc2 = dist_allreduce(e, b.st->red, 32)) return rc2;
            hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, s, c, n_dir_wg, n_con_wg, n_grid_wg, 1, 2);
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
    // (mpm_run_coupled_substeps puts GridToParticle in front of this kernel: the host has just learnt that the solve is
    // over and the device's queue is short -- a long kernel first gives the host time to enqueue what follows; the
    // impulses read the grid velocities and the contact arrays, which GridToParticle does not touch)
    // (not behind a solve that refused itself: the caller repeats that one, and GridToParticle belongs behind the repeat)
    if (e->ct_before_impulse && (host_driven || (oc->mb.done != CT_DONE_FAULT && oc->mb.done != CT_DONE_STALE && oc->mb.done != CT_DONE_CORRUPT)))
        e->ct_before_impulse();
    hipLaunchKernelGGL(k_ct_impulse, dim3(std::min(gc, 256u)), dim3(256), 0, s, p, c);
    HIP_TRY(hipGetLastError());
    if (host_driven) {
        // (these paths know the state: hand it over like a publication)
        oc->st = st;
        oc->have_state = true;
        oc->mb.done = st.done ? st.done : 1;
        oc->mb.iters = iters;
        oc->mb.residual = residual;
        oc->mb.count = (unsigned)b.n;
        oc->mb.nodes = (unsigned)st.n_nodes;
        oc->mb.unchanged = false;
        if (oc->mb.done == 2) oc->mb.done = 1;
    } else {
        e->last_contact_reused = reuse && oc->mb.done != CT_DONE_STALE;
        b.n_hint = oc->mb.count;
        // (how many blocks were active: from the control block as this solve's kernels saw it -- a word of the mailbox)
        b.n_active_hint = oc->mb.n_active;
    }
    return 0;
}


// ---- TEAM solve: the distributed contact solve, device resident (mpm_team.h) ------------------------------------------
static TeamDev team_dev(const mpm_engine* e) {
    TeamDev t{};
    const mpm_engine::Team& tm = e->team;
    const Dist& d = e->dp.dist;
    t.on = tm.on ? 1 : 0;
    t.rank = tm.rank; t.world = tm.world;
    t.left = d.has_left ? tm.rank - 1 : -1;
    t.right = d.has_right ? tm.rank + 1 : -1;
    t.ts = tm.ts;
    for (int r = 0; r < TEAM_MAX; ++r) t.peer[r] = static_cast<char*>(tm.peer[r]);
    t.zone_cap = (unsigned)tm.zone_cap;
    t.zone_bytes = tm.zone_bytes;
    t.timeout_ticks = (unsigned long long)((double)tm.timeout_s * 1e8);
    const int Z = d.zone_cells / 4;
    t.lo[0] = d.own_lo / 4 - Z; t.hi[0] = d.own_lo / 4 + Z - 1;
    t.lo[1] = d.own_hi / 4 - Z; t.hi[1] = d.own_hi / 4 + Z - 1;
    return t;
}

// One attempt of the solve on every LOCAL rank of the team (`L`: one engine in a process-per-GPU run, all of them in an
// in-process world), enqueued PHASE BY PHASE across the local ranks: with one rank that is simply the rank's own order;
// with several ranks on ONE stream every signal is enqueued before the wait that needs it, so the waits return at once
// and the same kernels serve both set-ups.  The host waits for nothing but the mailbox of the first local rank (all ranks
// take the same decisions from the same global sums, in the same pattern).
// `before_impulse(i)`: launched for rank i between the solve's last update and its impulses (GridToParticle), unless the
// solve refused itself.
static int team_solve_once(const std::vector<mpm_engine*>& L, float dt, float mu, float stiffness, float damping, int exact, int max_iters,
                           bool full_setup, std::vector<SolveOutcome>* ocs, const std::function<void(size_t)>& before_impulse) {
    struct Rank {
        mpm_engine* e;
        ContactDev c;
        TeamDev t;
        size_t n;
        unsigned gc, n_tile_wg;
    };
    std::vector<Rank> R(L.size());
    ocs->assign(L.size(), SolveOutcome());
    const int n_con_wg = CT_ROWS_CON, n_grid_wg = CT_ROWS, n_dir_wg = CT_DIR_WG;
    // ---- S1: keys, state reset, this rank's status on its way to every rank ---------------------------------------------
    for (size_t i = 0; i < L.size(); ++i) {
        mpm_engine* e = L[i];
        ContactBuffers& b = e->cb;
        Rank& r = R[i];
        r.e = e;
        r.t = team_dev(e);
        e->last_contact_reused = false;
        e->last_contact_gated = false;
        size_t n = b.n;
        if (b.dev_counted) n = b.n_hint && !full_setup ? std::min<size_t>(b.cap, b.n_hint + b.n_hint / 8 + 256) : b.cap;
        n = std::max<size_t>(n, 1);
        r.n = n;
        r.gc = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
        r.n_tile_wg = (unsigned)std::max<size_t>(1, std::min<size_t>((n + CT_TILE - 1) / CT_TILE, CT_TILE_WG));
        b.solves += 1;
        const DP& p = e->dp;
        const size_t blocks = b.n_active_hint && !full_setup ? std::min<size_t>((size_t)b.n_active_hint * 2, p.capA) : p.capA;
        int bits = 1;
        while (((size_t)1 << bits) < blocks * 64) ++bits;
        ContactDev c_in = make_contact_dev(e, dt, mu, stiffness, damping, max_iters, /* sorted = */ false);
        hipLaunchKernelGGL(k_ct_keys, dim3(1024), dim3(256), 0, e->stream, p, c_in, (const uint32_t*)b.api_idx, (const int*)e->d_pids_api,
                           b.slot, b.published, b.solves, bits, (int)std::min<size_t>(n, 0x7FFFFFFF), 0, full_setup ? 1 : 0, b.gen_stamp, 1);
        hipLaunchKernelGGL(k_team_status<0>, dim3(1), dim3(64), 0, e->stream, c_in, r.t, e->dp.ctl);
        // (the sort below runs whatever the status says: it only moves keys inside the buffers)
        bool in_alt = false;
        if (radix_sort_pairs(e->stream, b.key, b.order, b.key2, b.order2, b.sort_hist, n, std::min(bits + 1, 31), &in_alt,
                             b.dev_counted ? &b.st->n : nullptr))
            return fail(MPM_ERR_HIP, "contact sort failed");
        b.sorted_in_alt = in_alt;
        r.c = make_contact_dev(e, dt, mu, stiffness, damping, max_iters);
    }
    // ---- S2: the status of all ranks, constants in sorted order, node flags on their way to the neighbours ---------------
    for (Rank& r : R) {
        mpm_engine* e = r.e;
        hipStream_t s = e->stream;
        const DP& p = e->dp;
        hipLaunchKernelGGL(k_team_status<1>, dim3(1), dim3(64), 0, s, r.c, r.t, e->dp.ctl);
        hipLaunchKernelGGL(k_ct_prepare<false>, dim3(r.gc), dim3(256), 0, s, p, r.c);
        // a node in a zone may be reached by the neighbour's contacts only: both ranks need it listed
        hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, r.c, 0);
        hipLaunchKernelGGL(k_team_zone_pack<3>, dim3(e->g_grid, 2), dim3(256), 0, s, p, r.c, r.t, (const float4*)e->cb.hg, 0);
        hipLaunchKernelGGL(k_team_zone_signal, dim3(1), dim3(64), 0, s, r.c, r.t, 0);
    }
    // ---- S3: the neighbours' flags, node list ---------------------------------------------------------------------------
    for (Rank& r : R) {
        mpm_engine* e = r.e;
        hipStream_t s = e->stream;
        const DP& p = e->dp;
        hipLaunchKernelGGL(k_team_zone_wait, dim3(1), dim3(64), 0, s, r.c, r.t, 0, e->dp.ctl);
        hipLaunchKernelGGL(k_team_zone_add<3>, dim3(64, 2), dim3(256), 0, s, p, r.c, r.t, e->cb.hg, 0);
        hipLaunchKernelGGL(k_ct_flags_to_field, dim3(512), dim3(256), 0, s, p, r.c, 1);
        hipLaunchKernelGGL(k_ct_flag_bits, dim3(256), dim3(256), 0, s, p, r.c);
        hipLaunchKernelGGL(k_ct_node_list, dim3((unsigned)((p.capA + 255) / 256)), dim3(256), 0, s, p, r.c);
        hipLaunchKernelGGL(k_ct_node_runs, dim3(1024), dim3(256), 0, s, p, r.c);
    }
    // ---- Newton iterations ---------------------------------------------------------------------------------------------
    // direction: contacts -> per-cell sums -> this rank's (H, G) per node | zone exchange | solve per node
    auto direction_out = [&](Rank& r, int first, int lazy) {
        mpm_engine* e = r.e;
        hipStream_t s = e->stream;
        const DP& p = e->dp;
        hipLaunchKernelGGL(k_ct_tile, dim3(r.n_tile_wg), dim3(256), 0, s, p, r.c, first, lazy);
        hipLaunchKernelGGL(k_ct_node_dir<1>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, r.c, 0);
        hipLaunchKernelGGL(k_team_zone_pack<3>, dim3(e->g_grid, 2), dim3(256), 0, s, p, r.c, r.t, (const float4*)e->cb.hg, 1);
        hipLaunchKernelGGL(k_team_zone_signal, dim3(1), dim3(64), 0, s, r.c, r.t, 1);
    };
    auto direction_in = [&](Rank& r, int lazy) {
        mpm_engine* e = r.e;
        hipStream_t s = e->stream;
        const DP& p = e->dp;
        hipLaunchKernelGGL(k_team_zone_wait, dim3(1), dim3(64), 0, s, r.c, r.t, 1, e->dp.ctl);
        hipLaunchKernelGGL(k_team_zone_add<3>, dim3(64, 2), dim3(256), 0, s, p, r.c, r.t, e->cb.hg, 1);
        hipLaunchKernelGGL(k_ct_node_dir<2>, dim3(n_dir_wg), dim3(CT_WG), 0, s, p, r.c, lazy);
    };
    const int PROBES = 6;
    auto pattern = [&](int index) -> int {
        if (!exact) {
            // (lazy update as on one GPU: the accepted step reaches the grid in the next iteration's k_ct_node_dir<2>)
            const int lazy = index == 0 ? 0 : 1;
            for (Rank& r : R) direction_out(r, index == 0, lazy);
            for (Rank& r : R) {
                direction_in(r, lazy);
                hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, r.e->stream, r.e->dp, r.c, n_con_wg, 0, 0.f);
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, r.e->stream, r.c, n_dir_wg, n_con_wg, n_grid_wg, 0, 4, r.t);
            }
            for (Rank& r : R)
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, r.e->stream, r.c, n_dir_wg, n_con_wg, n_grid_wg, 0, 3, r.t);
            return 2;
        }
        // exact line search: direction, PROBES x (energies at st->alpha_probe | all ranks' sums | state machine), apply, close
        for (Rank& r : R) direction_out(r, index == 0, 0);
        for (Rank& r : R) direction_in(r, 0);
        for (int k = 0; k < PROBES; ++k) {
            for (Rank& r : R) {
                hipLaunchKernelGGL(k_ct_ls, dim3(n_con_wg + n_grid_wg), dim3(CT_WG), 0, r.e->stream, r.e->dp, r.c, n_con_wg, 2, 0.f);
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, r.e->stream, r.c, n_dir_wg, n_con_wg, n_grid_wg, 2, 4, r.t);
            }
            for (Rank& r : R)
                hipLaunchKernelGGL(k_ct_decide, dim3(1), dim3(1024), 0, r.e->stream, r.c, n_dir_wg, n_con_wg, n_grid_wg, 2, 3, r.t);
        }
        for (Rank& r : R) {
            hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, r.e->stream, r.e->dp, r.c, 1);
            hipLaunchKernelGGL(k_ct_exact_finish, dim3(1), dim3(64), 0, r.e->stream, r.c);
        }
        return 2 * PROBES + 1;
    };
    {
        const int max_patterns = exact ? (200 / PROBES + 2) * max_iters + 64 : 1 << 30;
        int launched = 0;
        std::vector<unsigned> older(L.size());
        auto enqueue = [&](int count) {
            for (int q = 0; q < count; ++q) {
                const int pub = pattern(launched + q);
                for (mpm_engine* e : L) e->cb.published += (unsigned)pub;
            }
            launched += count;
        };
        enqueue(exact ? ct_batch(L[0], 0, 1) : ct_batch(L[0], 0, 2));
        while (true) {
            for (size_t i = 0; i < L.size(); ++i) older[i] = L[i]->cb.published;
            enqueue(ct_batch(L[0], 1, 1));   // speculative: idle (exchanges included) if the previous batch finished
            MailboxState& mb = (*ocs)[0].mb;
            if (int rc = wait_mailbox(L[0], older[0], &mb)) return rc;
            if (mb.done || mb.iters >= max_iters || launched > max_patterns) break;
        }
        // (the other local ranks' words of THE SAME publication -- the one the first rank's poll happened to see, which may
        // be later than the one it waited for --: the same decisions, their own counts)
        const unsigned ahead = (*ocs)[0].mb.seq - older[0];
        for (size_t i = 1; i < L.size(); ++i) {
            if (int rc = wait_mailbox(L[i], older[i] + ahead, &(*ocs)[i].mb)) return rc;
            if ((*ocs)[i].mb.done != (*ocs)[0].mb.done && (*ocs)[i].mb.seq == older[i] + ahead)
                return fail(MPM_ERR_INTERNAL, "team solve: two ranks of one process disagree about the state of the solve");
        }
    }
    // ---- the last step, GridToParticle, impulses -----------------------------------------------------------------------
    for (size_t i = 0; i < L.size(); ++i) {
        Rank& r = R[i];
        mpm_engine* e = r.e;
        const int done = (*ocs)[i].mb.done;
        if (!exact) hipLaunchKernelGGL(k_ct_apply, dim3(CT_ROWS), dim3(CT_WG), 0, e->stream, e->dp, r.c, 2);
        if (before_impulse && done != CT_DONE_FAULT && done != CT_DONE_STALE && done != CT_DONE_CORRUPT) before_impulse(i);
        hipLaunchKernelGGL(k_ct_impulse, dim3(std::min(r.gc, 256u)), dim3(256), 0, e->stream, e->dp, r.c);
        HIP_TRY(hipGetLastError());
        e->cb.n_hint = (*ocs)[i].mb.count;
        e->cb.n_active_hint = (*ocs)[i].mb.n_active;
    }
    return 0;
}

// The solve of all local ranks, refusals repeated (as update_contact does for one engine): the ranks agree on a refusal on
// the device (k_team_status), so every rank's host sees the same code and all repeat together.
static int team_solve(const std::vector<mpm_engine*>& L, float dt, float mu, float stiffness, float damping, int exact, int max_iters,
                      const std::function<void(size_t)>& before_impulse, std::vector<SolveOutcome>* ocs) {
    if (max_iters <= 0) max_iters = 2000;  // cuda_mpm_solver.cu:234
    for (mpm_engine* e : L) {
        REQUIRE(e->team.on && e->dp.dist.on, "team solve: mpm_dist_init and mpm_team_connect first");
        REQUIRE(e->cb.n_bodies > 0, "call mpm_reallocate_external_bodies before the contact solve");
        GrowPoison gp(e);
        // (a rank without a single pair so far takes part all the same: its buffers must exist)
        if (int rc = ensure_contact_capacity(e, std::max<size_t>(e->cb.cap, e->ct_initial_capacity ? e->ct_initial_capacity
                                                                                                    : std::max<size_t>(4096, e->np / 16))))
            return rc;
        e->last_contact_dt = dt; e->last_contact_mu = mu; e->last_contact_k = stiffness; e->last_contact_d = damping;
    }
    bool full_setup = false;
    for (int attempt = 0;; ++attempt) {
        REQUIRE(attempt < 8, "contact solve: the set-up keeps being refused");
        if (int rc = team_solve_once(L, dt, mu, stiffness, damping, exact, max_iters, full_setup, ocs, before_impulse)) return rc;
        const int done = (*ocs)[0].mb.done;
        if (done == CT_DONE_CORRUPT) {
            for (mpm_engine* e : L) {
                e->cb.dev_counted = false;
                e->cb.n = 0;
            }
            return fail(MPM_ERR_CAPACITY, "contact solve: a rank's pair count on the device is not one the solve may index with "
                                          "(outside the capacity of its buffers, or stale): nothing was solved on any rank; make the pairs again");
        }
        if (done == CT_DONE_FAULT) {
            for (size_t i = 0; i < L.size(); ++i) {
                mpm_engine* e = L[i];
                GrowPoison gp(e);
                e->ct_counters[3] += 1;
                if (int rc = ensure_contact_capacity(e, std::max<size_t>((size_t)(*ocs)[i].mb.count, e->cb.cap))) return rc;
                if (e->cb.dev_counted)   // (pairs made on the device: again, into the grown buffers; nothing has used them yet)
                    if (int rc = generate_contacts_launch(e)) return rc;
            }
            full_setup = true;
            continue;
        }
        if (done == CT_DONE_STALE) {
            for (mpm_engine* e : L) {
                e->ct_counters[2] += 1;
                e->cb.n_active_hint = 0;
            }
            full_setup = true;
            continue;
        }
        break;
    }
    for (size_t i = 0; i < L.size(); ++i) {
        mpm_engine* e = L[i];
        ContactBuffers& b = e->cb;
        const MailboxState& mb = (*ocs)[i].mb;
        e->ct_quiet_left = mb.quiet_left;
        e->ct_counters[0] += 1;
        if (b.dev_counted) {
            b.n = mb.count;
            b.dev_counted = false;
        }
        b.last_iters = mb.iters;
        b.last_unchanged = false;
        mpm_contact_stats_t& cs = e->last_contact;
        cs = mpm_contact_stats_t{};
        cs.iterations = mb.iters;
        cs.contacts = (uint32_t)b.n;
        cs.nodes = mb.nodes;
        cs.residual = mb.residual;
        e->last_contact_on_device = true;
        e->last_contact_exact = exact != 0;
    }
    return 0;
}
