// Device buffers and kernels of the grid-space contact solve (UpdateContact,
// cuda_mpm_solver.cu:214-621): Jacobi-Newton on the grid velocities of the
// SAP-style cost  E(v) = sum_nodes 1/2 m |v - v*|^2 + sum_contacts m_p l(v_p).
//
// Differences in organisation from the reference (same arithmetic):
//   - the contact -> node scatter of Hessians/gradients (12 float atomics x 27
//     nodes per contact per iteration, cuda_mpm_kernels.cuh:1184-1212) is two
//     fixed-order sums.  Contact positions are fixed during the solve, so once per
//     UpdateContact the contacts are sorted by the cell of their stencil base
//     (stable radix sort, mpm_sort.h).  Contacts of one cell share their 27 stencil
//     nodes: k_ct_tile (a workgroup per 64 sorted contacts) reads those nodes once
//     into LDS, evaluates the contacts and sums, per cell, what they add to each of
//     the 27 nodes; k_ct_node_dir adds the sums of a node's 27 base cells, found
//     through the per-cell runs.  No per-(node, contact) adjacency in memory;
//   - all per-contact work arrays live in that sorted order (coalesced);
//   - the backtracking line search evaluates the step lengths 1, 1/2, 1/4, 1/8 in
//     one pass (the other 24, down to 2^-27, in a second pass when none of these is
//     accepted) and picks the first acceptable one on the device, so an iteration
//     needs no host round trip (the reference syncs >= 3 times);
//   - the accepted step is added to the grid by the next iteration's k_ct_node_dir
//     (the kernel that replaces the direction), not by a launch of its own;
//   - global scalars are reduced per workgroup and then in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "mpm_device.h"
#include "mpm_sort.h"
#include "mpm_rootfind.h"
#include "mpm_team_dev.h"

namespace mpm {

constexpr int LS_CAND = 28;          // alpha = 2^-j, j = 0..27 (alpha < 1e-8 is accepted as is)
constexpr int LS_SHALLOW = 4;        // candidates of the first pass; the other 24 only when none of these is accepted
constexpr int CT_PART = LS_CAND + 4; // partial record: E1[28], E0, norm_dir, dofs, pad
constexpr int CT_WG = 256;           // threads per workgroup of the contact kernels
constexpr int CT_DIR_WG = 1024;      // workgroups of k_ct_node_dir (16 lanes per node)
constexpr int CT_TILE = 64;          // sorted contacts handled by one workgroup of k_ct_tile
constexpr int CT_TILE_WG = 2048;     // workgroups of k_ct_tile at most
static_assert(CT_TILE * 4 == 256, "k_ct_tile: 4 lanes per contact, one 16-byte piece of the record each");
constexpr int CT_STAGE = 16;         // segments whose stencil nodes are staged in LDS at a time
constexpr int CT_SEG_F = 27 * 12;    // floats of one segment's sums: (H[9], G[3]) per stencil node
constexpr int CT_ROWS = 64;          // partial-sum records of the node part of k_ct_ls (= its workgroups)
constexpr int CT_ROWS_CON = 1024;    // ... of the contact part (64 contacts per workgroup and pass)
constexpr uint32_t CT_NO_CELL = 0x7FFFFFFFu;  // sort key of a contact whose base cell is outside the active grid

struct ContactState {       // device-resident solver state
    int done;               // 0 running, 2 finish after this update, 1 finished
    int iters;
    int ls_total;           // accumulated line-search evaluations (statistics)
    int n_nodes;            // nodes that see at least one contact
    float alpha;            // step accepted in the current iteration
    float residual;         // sqrt(sum |Dir|^2) / DoFs
    float energy;
    float E0;
    float norm_dir_sq;      // sum |Dir|^2 (before relaxation) of the current iteration
    float dofs;
    float pad2[2];
    double scal[4];         // exact line search: E, dE, d2E at the probed alpha
    double red[32];         // partitioned domain: this rank's sums (energies[29], |Dir|^2, DoFs) on their way
                            // through the all-reduce, then the global ones
    // exact line search kept on the device (cuda_mpm_solver.cu:383-471): 0 = new Newton iteration, the
    // probe at alpha = 0 is pending; 1 = probe at alpha = 1 pending; 2 = root finder running;
    // 3 = step decided, to be applied.  Backtracking search: 4 = none of the first LS_SHALLOW candidates
    // was accepted, the next pass of k_ct_ls / k_ct_decide tries the rest (direction kernels skip)
    int ls_phase;
    int ls_evals;           // evaluations of the root finder in this Newton iteration
    float alpha_probe;      // where k_ct_ls evaluates next
    float f_lo[3];          // (E, dE, d2E) at alpha = 0
    RootFinder<float> rf;
    // ---- what the HOST would otherwise have to read back before it can enqueue the solve (round 5) -------------------
    // Pairs made on the device (mpm_generate_contact_pairs) are COUNTED on the device: the count stays here, every
    // kernel of the solve reads it (ct_count), every launch that depends on it has a fixed grid.  More pairs than the
    // buffers hold: gen_fault -- the solve's kernels skip themselves (done = CT_DONE_FAULT), the host learns it from the
    // first mailbox word it waits for anyway, grows the buffers and repeats pair generation + solve (nothing has
    // touched the grid).  None of these fields is reset by the solve's set-up.
    int n;                  // contacts the solve works on (<= capacity)
    int n_wanted;           // pairs the scene has (mpm_generate_contact_pairs)
    int gen_fault;          // n_wanted > capacity
    unsigned n_stamp;       // number of the pair generation that wrote the three words above (k_ct_gen_write's argument): the
                            // solve is enqueued for ONE generation and refuses a count that another one left (k_ct_keys)
    unsigned seq;           // publications in the mailbox so far (k_ct_decide, k_ct_exact_finish); the host re-bases it
                            // with every solve (k_ct_keys)
    // settled scenes: the pair list of this solve equals the previous solve's entry by entry (same particle, same body,
    // same base cell, same block tables): changed_solve != the solve's number (kernel argument of k_ct_keys)
    unsigned changed_solve;
    unsigned solve_no;      // this solve's number
    unsigned n_active;      // active blocks as this solve's set-up saw them (the host sizes the next solve's sort keys from it)
    int prev_n;             // contacts of the previous solve
    unsigned setup_rebuilds;  // Ctl::rebuilds when the per-cell tables were last built
};
constexpr int CT_DONE_FAULT = 4;     // ContactState::done: the pair buffers overflowed, nothing was solved
constexpr int CT_DONE_STALE = 5;     // ... a speculated set-up (previous solve's order reused) did not match, nothing was solved
constexpr int CT_DONE_GATED = 6;     // ... the substep was enqueued without its re-sort launches and found a re-sort pending: all of
                                     // it skipped itself (DP::gated), the host runs it again with the re-sort in front
constexpr int CT_DONE_CORRUPT = 7;   // ... the pair count on the device is not one this solve may index with: negative, larger than
                                     // the capacity that sized the per-pair buffers (MPM_ERR_CAPACITY), or left by another pair
                                     // generation than the one the solve was enqueued for (MPM_ERR_INTERNAL).  Nothing was solved,
                                     // nothing was indexed with it, and the host does NOT repeat the call: it reports the error.
                                     // (Round 5's memory access fault -- DESIGN.md section 3.3 -- was a launch that indexed a
                                     // per-pair array from its workgroup number alone; the invariant since: EVERY index into a
                                     // per-pair array is formed from ct_count(), which is clamped to ContactDev::stride.)

// Mailbox in host-mapped pinned memory: what the host polls instead of copying the state back after every batch of
// iterations (a blit kernel of ~4 us on the engine's stream per read-back, 8 - 20 per solve).  Four
// words, each written by ONE naturally aligned 8-byte store (untorn) and each carrying the publication number in its
// upper half, so that the host can tell a consistent set without any fence on the device:
//   w[0] = seq << 32 | done << 24 | iterations        w[1] = seq << 32 | bits of the residual (float)
//   w[2] = seq << 32 | contacts (the wanted count when done = CT_DONE_FAULT)
//   w[3] = seq << 32 | unchanged << 31 | nodes that see contacts          w[4] = seq << 32 | active blocks
//   w[5] = seq << 32 | bits of the quiet time left (float seconds; 0 when a re-sort is pending: Ctl::quiet_time)
struct ContactMailbox {
    unsigned long long w[8];
};

constexpr int CT_LOG = 2048;   // Newton iterations whose decisions are logged (mpm_download_contact_log, the JSON dump)
constexpr int CT_LOG_F = 8;    // floats per logged iteration

struct ContactDev {
    int n;                  // contacts; < 0: counted on the device, read ContactState::n (ct_count)
    int stride;             // distance between the planes of the per-contact SoA arrays below (the buffers' capacity)
    int max_iters;
    int force;              // mpm_profile_contact_iteration: the iteration kernels run whatever the solver state says
    float dt, mu, k, d, epsv, relax, tol;
    // as handed over by CopyContactPairs (contact order of the caller)
    const uint32_t* slot;   // internal particle slot
    const uint32_t* body;
    const float *dist, *normal, *pos, *rigid_v, *p_WB;
    float *vel, *vel0;      // contact_vel / contact_vel0, caller's order
    // sorted by base cell (position j <-> caller's contact order[j])
    uint32_t* key;          // [n] compact base cell (active slot * 64 + cell) or CT_NO_CELL
    uint32_t* order;        // [n]
    int* cnode;             // [27][n] compact grid index of the stencil nodes or -1
    float* cfx;             // [3][n]
    float* cmass;           // [n]
    float* cphi0;           // [n]
    float* cR;              // [9][n] world -> contact frame
    float* cv0;             // [3][n] lagged relative velocity, contact frame
    float* crv;             // [3][n] rigid velocity at the contact point
    float* cvel;            // [3][n] contact velocity (sorted order)
    int2* run;              // [cells] (begin, end) of the contacts whose base cell this is
    int* node_flag;         // [cells] 1 if some contact's stencil reaches the node
    unsigned long long* flag_bits;   // [active blocks] node_flag as one bit per cell
    int* node_list;         // [<= cells] those nodes
    int2* node_runs;        // [27][cap_nodes] per listed node and stencil offset: contact run of the base cell there
    float* seg_part;        // [n][27][12], filled at the first contact of every segment (= the contacts of one
                            // cell inside one tile of CT_TILE sorted contacts): what the segment adds to (H, G) of
                            // the 27 nodes of its stencil
    int cap_nodes;
    float4* gD;             // [cells] search direction (relaxed)
    float4* hg;             // [cells][3] partitioned domain: contact Hessian (9) and gradient (3) sums per node,
                            // this rank's contacts first, the neighbours' added by the zone exchange
    double* part;           // [CT_ROWS_CON + CT_ROWS][CT_PART] line-search partial sums (contacts, then cells)
    double* part_dir;       // [CT_DIR_WG][2] (|Dir|^2, DoFs) per workgroup of k_ct_node_dir
    ContactState* st;
    ContactMailbox* mbox;   // host-mapped (device address)
    const Ctl* ctl;         // the engine's control block (the publication carries the quiet time left)
    // the previous solve's pair list in the caller's order, for the "unchanged" test of k_ct_keys
    uint32_t *prev_key, *prev_api, *prev_body;
    float* it_log;          // [CT_LOG][CT_LOG_F] per Newton iteration: residual, line-search evaluations, E(alpha), alpha,
                            // E(0), sum |Dir|^2, DoFs (both searches; mpm_download_contact_log)
    long long* body_acc;    // [n_bodies][6] (tau, f) impulse sums in 64-bit fixed point (k_ct_impulse)
    double imp_fix;         // its scale (DP::fix_p: the momentum scale of ParticleToGrid's tiles)
    int n_bodies;
};

struct Collider {        // mirrors mpm_collider_t (include/mpm_hip.h)
    int kind;
    uint32_t body;
    float p[3], R[9], dims[3], v[3], w[3];
};

struct ContactBuffers {
    size_t n = 0, cap = 0;
    size_t n_bodies = 0, cap_bodies = 0;
    size_t cap_cells = 0;
    // contact SoA (MpmParticleContactPairs, cpu_mpm_model.h:73-114)
    uint32_t* slot = nullptr;
    uint32_t* body = nullptr;
    float* dist = nullptr;
    float* normal = nullptr;
    float* pos = nullptr;
    float* rigid_v = nullptr;
    float* p_WB = nullptr;
    float* vel = nullptr;       // contact_vel
    float* vel0 = nullptr;      // contact_vel0
    uint32_t* api_idx = nullptr;  // particle_in_contact_index (caller's slot), kept for downloads
    Collider* colliders = nullptr;
    size_t cap_colliders = 0;
    int *gen_cnt = nullptr, *gen_sums = nullptr;
    uint32_t *key = nullptr, *order = nullptr, *key2 = nullptr, *order2 = nullptr;
    int* sort_hist = nullptr;
    size_t cap_hist = 0;
    int* cnode = nullptr;
    float *cfx = nullptr, *cmass = nullptr, *cphi0 = nullptr, *cR = nullptr, *cv0 = nullptr, *crv = nullptr,
          *cvel = nullptr;
    int2* run = nullptr;
    int* node_flag = nullptr;
    unsigned long long* flag_bits = nullptr;
    int* node_list = nullptr;
    int2* node_runs = nullptr;
    float* seg_part = nullptr;
    float4* gD = nullptr;
    float4* hg = nullptr;
    void* zone_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // send left / right, receive left / right
    size_t zone_cap = 0, zone_bytes = 0;
    double* part = nullptr;
    double* part_dir = nullptr;
    ContactState* st = nullptr;
    float* it_log = nullptr;
    int last_iters = 0;         // Newton iterations of the previous solve (sizes the first batch of launches)
    unsigned n_active_hint = 0; // active blocks as the last solve saw them (0: unknown); sizes the sort keys, verified on the device
    unsigned n_hint = 0;        // contacts of the last solve (sizes the grids of a solve whose count is on the device)
    ContactState* h_st[2] = {nullptr, nullptr};   // pinned read-back slots of the batched loops
    hipEvent_t h_ev[2] = {nullptr, nullptr};
    ContactMailbox* h_mbox = nullptr;   // host-mapped pinned mailbox (host address) ...
    ContactMailbox* d_mbox = nullptr;   // ... and its device address
    unsigned published = 0;             // publications enqueued so far (the device's ContactState::seq follows it)
    unsigned solves = 0;                // mpm_update_contact calls so far (numbers the "unchanged" test)
    bool dev_counted = false;           // the pairs in the buffers were counted on the device and the host has not read the count
    unsigned gen_stamp = 0;             // pair generations launched so far (ContactState::n_stamp follows it)
    uint32_t *prev_key = nullptr, *prev_api = nullptr, *prev_body = nullptr;
    // the colliders of the last mpm_generate_contact_pairs (a buffer overflow repeats the generation)
    std::vector<Collider> last_colliders;
    bool sorted_in_alt = false;         // the sorted (key, order) of the last set-up sit in key2 / order2
    bool last_unchanged = false;        // the last solve found its pair list equal to its predecessor's
    // F_Bq_W_tau / F_Bq_W_f of the reference (cuda_mpm_model.cuh: float arrays fed by float atomics, whose sums depend on
    // the order of arrival): accumulated in 64-bit fixed point -- integer sums, exact and independent of the order --
    // and converted when ExternelBodyForceToHost downloads them.  [body][tau xyz, f xyz]
    long long* body_acc = nullptr;
    double imp_unfix = 0.0;     // 1 / scale of what has been accumulated since the last reset

    void release() {
        void* ptrs[] = {api_idx, colliders, gen_cnt, gen_sums, slot, body, dist, normal, pos, rigid_v, p_WB, vel, vel0, key, order, key2, order2, sort_hist,
                        cnode, cfx, cmass, cphi0, cR, cv0, crv, cvel, run, node_flag, flag_bits, node_list, node_runs, seg_part, gD, hg,
                        zone_buf[0], zone_buf[1], zone_buf[2], zone_buf[3], part, part_dir, st, it_log, body_acc,
                        prev_key, prev_api, prev_body};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        for (int i = 0; i < 2; ++i) {
            if (h_st[i]) (void)hipHostFree(h_st[i]);
            if (h_ev[i]) (void)hipEventDestroy(h_ev[i]);
        }
        if (h_mbox) (void)hipHostFree(h_mbox);
        *this = ContactBuffers();
    }
    int resize_bodies(size_t nb, hipStream_t s) {
        if (nb > cap_bodies) {
            if (body_acc) (void)hipFree(body_acc);
            body_acc = nullptr;
            if (hipMalloc((void**)&body_acc, nb * 6 * sizeof(long long)) != hipSuccess) return -2;
            cap_bodies = nb;
        }
        n_bodies = nb;
        // reset at the beginning of each time step (cuda_mpm_model.cu:334-337)
        if (cap_bodies && hipMemsetAsync(body_acc, 0, cap_bodies * 6 * sizeof(long long), s) != hipSuccess) return -2;
        return 0;
    }
};

// ---- CopyContactPairs ----------------------------------------------------------

// API slot -> internal slot of the particle each contact refers to
__global__ __launch_bounds__(256) void k_ct_slots(int n, const uint32_t* api_slot, const int* pids_api,
                                                  const int* imap, uint32_t* out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = (uint32_t)imap[pids_api[api_slot[k]]];
}

// initialize_contact_velocities (cuda_mpm_kernels.cuh:926-938)
__global__ __launch_bounds__(256) void k_ct_init_vel(DP p, ContactDev c) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n) return;
    const PSet& S = p.set[p.ctl->cur];
    const float4 v = S.q[1][c.slot[k]];
    c.vel[k * 3] = v.x; c.vel[k * 3 + 1] = v.y; c.vel[k * 3 + 2] = v.z;
}

// ---- device-side CalcMpmContactPairs (deformable_driver.h:120-194) ---------------------------
// signed distance of the world point x to collider c and its unit gradient in the world
MPM_DEV float collider_sdf(const Collider& c, const float* x, float* grad) {
    const float d[3] = {x[0] - c.p[0], x[1] - c.p[1], x[2] - c.p[2]};
    // body frame: xb = R^T d
    const float xb[3] = {c.R[0] * d[0] + c.R[3] * d[1] + c.R[6] * d[2], c.R[1] * d[0] + c.R[4] * d[1] + c.R[7] * d[2],
                         c.R[2] * d[0] + c.R[5] * d[1] + c.R[8] * d[2]};
    float gb[3] = {0.f, 0.f, 1.f}, phi;
    if (c.kind == 0) {            // half-space z_B <= 0
        phi = xb[2];
    } else if (c.kind == 1) {     // sphere
        const float len = sqrtf(xb[0] * xb[0] + xb[1] * xb[1] + xb[2] * xb[2]);
        phi = len - c.dims[0];
        if (len > 0.f) { gb[0] = xb[0] / len; gb[1] = xb[1] / len; gb[2] = xb[2] / len; }
    } else if (c.kind == 2) {     // box with half extents dims
        const float q[3] = {fabsf(xb[0]) - c.dims[0], fabsf(xb[1]) - c.dims[1], fabsf(xb[2]) - c.dims[2]};
        const float sg[3] = {xb[0] < 0.f ? -1.f : 1.f, xb[1] < 0.f ? -1.f : 1.f, xb[2] < 0.f ? -1.f : 1.f};
        const float m = fmaxf(q[0], fmaxf(q[1], q[2]));
        if (m <= 0.f) {           // inside: distance to the nearest face
            const int a = (q[0] >= q[1] && q[0] >= q[2]) ? 0 : (q[1] >= q[2] ? 1 : 2);
            phi = m;
            gb[0] = a == 0 ? sg[0] : 0.f; gb[1] = a == 1 ? sg[1] : 0.f; gb[2] = a == 2 ? sg[2] : 0.f;
        } else {
            const float o[3] = {fmaxf(q[0], 0.f), fmaxf(q[1], 0.f), fmaxf(q[2], 0.f)};
            phi = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
            gb[0] = sg[0] * o[0] / phi; gb[1] = sg[1] * o[1] / phi; gb[2] = sg[2] * o[2] / phi;
        }
    } else {                      // capsule along z_B
        const float zc = fminf(fmaxf(xb[2], -c.dims[1]), c.dims[1]);
        const float r[3] = {xb[0], xb[1], xb[2] - zc};
        const float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        phi = len - c.dims[0];
        if (len > 0.f) { gb[0] = r[0] / len; gb[1] = r[1] / len; gb[2] = r[2] / len; }
    }
    mulv3(c.R, gb, grad);         // world = R gb
    return phi;
}

// The colliders of a call travel as a kernel ARGUMENT while there are at most CT_COLLIDER_ARGS of them (no upload, no
// synchronisation: an upload of 92 bytes per collider costs a blit launch and, with the host buffer on the stack, a wait
// for the stream); beyond that through a device array that the host refreshes only when the colliders have changed.
constexpr int CT_COLLIDER_ARGS = 16;
struct ColliderTable {
    int n;
    const Collider* dev;   // non-null: n colliders there
    Collider c[CT_COLLIDER_ARGS];
};
MPM_DEV const Collider& collider_of(const ColliderTable& t, int j) { return t.dev ? t.dev[j] : t.c[j]; }

// this solve's contact count: the host's when it knows it, else the device's (mpm_generate_contact_pairs without a
// read-back)
// ALWAYS inside the capacity that sized the per-pair arrays (ContactDev::stride), whatever the word on the device holds:
// a count that is negative or larger is a corrupt one -- k_ct_keys refuses the solve for it (CT_DONE_CORRUPT) --, but no
// kernel may index with it before, after or instead of that refusal.
MPM_DEV int ct_count_raw(const ContactDev& c) { return c.n >= 0 ? c.n : c.st->n; }
MPM_DEV int ct_count(const ContactDev& c) { return min(max(ct_count_raw(c), 0), c.stride); }

// P1: number of penetrated colliders per particle slot (the caller's slot order), 0 in the padding of the scan's storage.
// With the scan inside 4096-blocks in ONE launch (round 5: a kernel boundary and k_scan_blocks' 5 us less per coupled
// substep): a workgroup of 1024 threads owns a 4096-block, a thread four consecutive slots -- their three dependent loads
// (API slot -> original id -> slot -> position) are four independent chains --, then the exclusive scan of the block's
// counts in place and the block's total (what a count kernel + k_scan_blocks left until then).
__global__ __launch_bounds__(1024) void k_ct_gen_count_scan(DP p, const int* pids_api, ColliderTable cols, int* cnt, int* sums) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * 4096 + tid * 4;
    const PSet& S = p.set[p.ctl->cur];
    int slot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = base + k;
        slot[k] = s < p.NpG ? p.imap[pids_api[s]] : -1;
    }
    float4 q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = slot[k] >= 0 ? S.q[0][slot[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
    int n[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x[3] = {q[k].x, q[k].y, q[k].z};
        n[k] = 0;
        // (partitioned domain: only the particles this rank owns make contacts here)
        for (int j = 0; j < cols.n && q[k].w > 0.f; ++j) {
            float g[3];
            n[k] += collider_sdf(collider_of(cols, j), x, g) < 0.f ? 1 : 0;
        }
    }
    const int sum = n[0] + n[1] + n[2] + n[3];
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int k = 0; k < w; ++k) run += s_w[k];
    if (tid == 1023) sums[blockIdx.x] = run + sum;
    int4 o;
    o.x = run; run += n[0];
    o.y = run; run += n[1];
    o.z = run; run += n[2];
    o.w = run;
    *reinterpret_cast<int4*>(cnt + base) = o;
}

// "Would the next substep have contact pairs?" -- asked right after GridToParticle, answered for mpm_run_coupled_substeps,
// which then enqueues that substep WITHOUT pair generation and contact solve (and without waiting for anything).  Exact: the
// pairs of a substep are the (particle, collider) with phi < 0 at the positions the substep starts from -- vertices where
// GridToParticle has just put them, faces at the centroid of their corners (CalcFemStateAndForce's first act,
// cuda_mpm_kernels.cuh:203-207) -- so "no particle with phi < margin" means "no pairs", and the margin (a thousandth of a
// cell) only absorbs the last bit of the centroid.  A hit is a plain store of the launch's number (any wave that sees one).
__global__ __launch_bounds__(256) void k_ct_watch(DP p, ColliderTable cols, unsigned seq) {
    const Ctl* c = p.ctl;
    if (p.gated && c->skip_this) return;   // (the substep in front of this launch skipped itself: nothing has moved)
    const PSet& S = p.set[c->cur];
    const int nf = c->nfa, total = nf + c->nva;
    const float margin = 1e-3f * p.dx;
    bool hit = false;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int slot = active_slot(p, idx, nf);
        float x[3];
        if (slot < p.Nf) {
            const float4 f3 = S.fq[3][slot];
            const float4 xa = S.q[0][__float_as_int(f3.y)], xb = S.q[0][__float_as_int(f3.z)], xc = S.q[0][__float_as_int(f3.w)];
            x[0] = (xa.x + xb.x + xc.x) * (1.f / 3.f);
            x[1] = (xa.y + xb.y + xc.y) * (1.f / 3.f);
            x[2] = (xa.z + xb.z + xc.z) * (1.f / 3.f);
        } else {
            const float4 q = S.q[0][slot];
            x[0] = q.x; x[1] = q.y; x[2] = q.z;
        }
        for (int j = 0; j < cols.n; ++j) {
            float g[3];
            hit |= collider_sdf(collider_of(cols, j), x, g) < margin;
        }
    }
    if (__ballot(hit) && (threadIdx.x & 63) == 0) p.ctl->watch_hit = seq;
}

// P2: the pairs, at the scanned offsets (offset inside its 4096-block + the pairs of the blocks before it): ascending
// (slot, collider).  Every workgroup adds up the block totals it needs itself (a few hundred ints: rounds 1 - 4 scanned
// them with a single-workgroup kernel in between); workgroup 0 leaves the pair count where the solve reads it --
// ContactState::n / n_wanted / gen_fault -- instead of sending it to the host.
__global__ __launch_bounds__(256) void k_ct_gen_write(DP p, const int* pids_api, ColliderTable cols, const int* offs,
                                                      const int* sums, int nb, int cap, uint32_t* api_idx, ContactDev c,
                                                      unsigned stamp) {
    __shared__ int s_part[4];
    const int first = (int)(blockIdx.x * 256u) >> 12;   // 4096-block of this workgroup's first slot; a workgroup of 256 slots
                                                         // never straddles two (4096 is a multiple of 256)
    // pairs in the blocks before `first`, and in all blocks
    int before = 0, total = 0;
    for (int t = threadIdx.x; t < nb; t += 256) {
        const int v = sums[t];
        total += v;
        before += t < first ? v : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += __shfl_xor(before, d);
        total += __shfl_xor(total, d);
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = before;
    __syncthreads();
    before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = total;
    __syncthreads();
    total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ContactState* st = c.st;
        st->n_wanted = total;
        st->n = min(total, cap);
        st->gen_fault = total > cap ? 1 : 0;
        st->n_stamp = stamp;
    }
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.NpG) return;
    int at = offs[s] + before;
    // (the slot after the last one of a 4096-block starts the next block: its offset there is 0, the block's total is
    // what this slot's pairs end at)
    const int end = ((s + 1) & 4095) ? offs[s + 1] + before : before + sums[first];
    if (end == at) return;
    const PSet& S = p.set[p.ctl->cur];
    const uint32_t slot = (uint32_t)p.imap[pids_api[s]];
    const float4 q = S.q[0][slot], vq = S.q[1][slot];
    const float x[3] = {q.x, q.y, q.z};
    for (int j = 0; j < cols.n; ++j) {
        const Collider& cl = collider_of(cols, j);
        float g[3];
        const float phi = collider_sdf(cl, x, g);
        if (!(phi < 0.f) || at >= cap) continue;
        api_idx[at] = (uint32_t)s;
        const_cast<uint32_t*>(c.slot)[at] = slot;
        const_cast<uint32_t*>(c.body)[at] = cl.body;
        const_cast<float*>(c.dist)[at] = phi;
        const float r[3] = {x[0] - cl.p[0], x[1] - cl.p[1], x[2] - cl.p[2]};
        const float rv[3] = {cl.v[0] + cl.w[1] * r[2] - cl.w[2] * r[1], cl.v[1] + cl.w[2] * r[0] - cl.w[0] * r[2],
                             cl.v[2] + cl.w[0] * r[1] - cl.w[1] * r[0]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const_cast<float*>(c.normal)[at * 3 + d] = -g[d];
            const_cast<float*>(c.pos)[at * 3 + d] = x[d];
            const_cast<float*>(c.rigid_v)[at * 3 + d] = rv[d];
            const_cast<float*>(c.p_WB)[at * 3 + d] = cl.p[d];
        }
        // initialize_contact_velocities (cuda_mpm_kernels.cuh:926-938)
        c.vel[at * 3] = vq.x; c.vel[at * 3 + 1] = vq.y; c.vel[at * 3 + 2] = vq.z;
        ++at;
    }
}

// ---- set-up (once per UpdateContact) -----------------------------------------

MPM_DEV int compact_cell(const DP& p, uint32_t x, uint32_t y, uint32_t z) {
    const int a = p.lut_act[block_id(x >> 2, y >> 2, z >> 2)];
    return a < 0 ? -1 : a * 64 + (int)(((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u));
}

MPM_DEV void contact_base(const DP& p, const float* pos, uint32_t* b) {
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
#pragma unroll
    for (int d = 0; d < 3; ++d) b[d] = min(base_cell(pos[d], p.dxinv), hi);
}

// S1: sort key = compact index of the stencil's base cell; clears the per-cell tables
// (also re-resolves the pairs' particles: caller's slot -> engine id -> current internal slot, k_ct_slots' job; resets
// the solver state -- what a hipMemsetAsync did --; and compares the pair list with the previous solve's, entry by
// entry: ContactState::changed_solve)
// key_bits, count_bound: the host sized the sort for keys below 2^key_bits and at most count_bound pairs (from the
// previous solve's publication); reuse: the
// host enqueued the set-up that reuses the previous solve's sorted order, this kernel only verifies (k_ct_prepare<true>
// acts on the verdict); force_changed: this is the repetition of a refused solve (its first attempt has already overwritten
// the previous list with this one: "unchanged" would be a tautology)
__global__ __launch_bounds__(256) void k_ct_keys(DP p, ContactDev c, const uint32_t* api_slot, const int* pids_api, uint32_t* slot_out,
                                                 unsigned seq_base, unsigned solve_no, int key_bits, int count_bound, int reuse,
                                                 int force_changed, unsigned gen_stamp, int team) {
    ContactState* st = c.st;
    const int n_raw = ct_count_raw(c);
    const int n = ct_count(c);   // (clamped: nothing below indexes with the raw word)
    // the count must be one this solve may use: inside the capacity, and -- counted on the device -- written by the pair
    // generation the host enqueued this solve for
    const bool corrupt = n_raw != n || (c.n < 0 && st->n_stamp != gen_stamp);
    const unsigned n_active = p.ctl->n_active;
    if (blockIdx.x == 0) {
        // every word in front of ContactState::n starts a solve as zero, except `done`: finished at once when there is
        // no contact (cuda_mpm_solver.cu:216-217); refused when the pair buffers overflowed or the sort keys are wider
        // than the host assumed (the host repeats the call)
        constexpr int NW = (int)(offsetof(ContactState, n) / 4);
        static_assert(NW <= 256 && offsetof(ContactState, done) == 0, "state reset: one word per thread of one workgroup");
        const bool fault = c.n < 0 && st->gen_fault;
        // (count_bound: the sort's launch covers that many pairs)
        const bool narrow = !reuse && ((key_bits < 31 && ((unsigned long long)n_active * 64ull > (1ull << key_bits))) || n > count_bound);
        // (p.gated: the substep went without its re-sort launches; k_grid, in front of this kernel, has left its verdict)
        const bool gated = p.gated && p.ctl->skip_this;
        // (team: a rank without contacts takes part in the solve all the same -- its zone nodes see the neighbour's contacts,
        // its rows count in every sum --; "nobody has a contact" is decided by all ranks together, k_team_status)
        const int done0 = corrupt ? CT_DONE_CORRUPT
                                  : (gated ? CT_DONE_GATED : (fault ? CT_DONE_FAULT : (narrow ? CT_DONE_STALE : (n == 0 && !team ? 1 : 0))));
        // (a reused set-up keeps its node list, and with it the count of listed nodes)
        constexpr int W_NODES = (int)(offsetof(ContactState, n_nodes) / 4);
        if ((int)threadIdx.x < NW) {
            int v = 0;
            if (threadIdx.x == 0) v = done0;
            else if (reuse && (int)threadIdx.x == W_NODES) v = st->n_nodes;
            reinterpret_cast<int*>(st)[threadIdx.x] = v;
        }
        if (threadIdx.x == 0) {
            st->seq = seq_base;
            st->solve_no = solve_no;
            st->n_active = n_active;
            if (c.n >= 0) st->n = n;
            if (corrupt) st->n_wanted = n_raw;   // (what the publication reports: ct_publish)
            const unsigned rebuilds = (unsigned)p.ctl->rebuilds;
            if (force_changed || n != st->prev_n || rebuilds != st->setup_rebuilds) st->changed_solve = solve_no;
            st->prev_n = n;
            st->setup_rebuilds = rebuilds;
        }
    }
    const int gs = gridDim.x * 256, i0 = blockIdx.x * 256 + threadIdx.x;
    if (!reuse) {
        const int ncell = (int)n_active * 64;
        for (int g = i0; g < ncell; g += gs) {
            c.run[g] = make_int2(0, 0);
            c.node_flag[g] = 0;
            c.gD[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    bool differs = false;
    if (corrupt) return;   // (every later kernel of the solve returns on ContactState::done)
    bool bad_api = false;
    for (int k = i0; k < n; k += gs) {
        uint32_t b[3];
        contact_base(p, c.pos + (size_t)k * 3, b);
        const int cc = compact_cell(p, b[0], b[1], b[2]);
        const uint32_t key = cc < 0 ? CT_NO_CELL : (uint32_t)cc;
        uint32_t api = api_slot[k];
        const uint32_t body = c.body[k];
        // (an entry that names no particle -- only a pair list that was never written can hold one: uploaded lists are
        // checked on the host, generated ones are slots by construction -- must not become an index)
        if (api >= (uint32_t)p.NpG) {
            bad_api = true;
            api = 0u;
        }
        if (!reuse) {   // (reused set-up: the sorted keys and the order of the last full set-up stay where they are)
            c.key[k] = key;
            c.order[k] = (uint32_t)k;
        }
        slot_out[k] = (uint32_t)p.imap[pids_api[api]];
        differs |= c.prev_key[k] != key || c.prev_api[k] != api || c.prev_body[k] != body;
        c.prev_key[k] = key;
        c.prev_api[k] = api;
        c.prev_body[k] = body;
    }
    // (a plain store of the same value by every wave that saw a difference: thousands of same-address atomics would
    // serialise at ~50 ns each)
    if (__ballot(differs) && (threadIdx.x & 63) == 0) st->changed_solve = solve_no;
    if (__ballot(bad_api) && (threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
}

MPM_DEV float stencil_weight(const float* wx, const float* wy, const float* wz, int n) {
    return wx[n / 9] * wy[(n / 3) % 3] * wz[n % 3];
}

// grid_to_particle_kernel<CONTACT_TRANSFER=true> (cuda_mpm_kernels.cuh:860-866, 891-894):
// velocity at a contact point (stencil nodes g[27], -1 = not on the active grid) from nodes with m > 1e-7
MPM_DEV void gather_from_nodes(const DP& p, const int* g, float fx, float fy, float fz, float* v) {
    float wx[3], wy[3], wz[3];
    bspline3(fx, wx);
    bspline3(fy, wy);
    bspline3(fz, wz);
    v[0] = v[1] = v[2] = 0.f;
    // branch-free: a skipped node costs a load of cell 0 and a zero weight, but the 27 node loads are all in flight
    // together (a branch per node serialises them: one L2 round trip each)
    float4 q[27];
#pragma unroll
    for (int n = 0; n < 27; ++n) q[n] = p.gv[max(g[n], 0)];
#pragma unroll
    for (int n = 0; n < 27; ++n) {
        const float w = (g[n] >= 0 && q[n].w > 1e-7f) ? stencil_weight(wx, wy, wz, n) : 0.f;
        v[0] += w * q[n].x;
        v[1] += w * q[n].y;
        v[2] += w * q[n].z;
    }
}

// S2 (after the sort): everything that stays fixed during the solve, in sorted order
// (stencil, fx, mass, contact frame, lagged velocity: cuda_mpm_kernels.cuh:1107-1139), and the pre-contact velocity at
// the contact points (contact_vel0, cuda_mpm_solver.cu:267-272: the caller's order).
// REUSE: the pair list equals the previous solve's (ContactState::changed_solve, verified by k_ct_keys in front of this
// launch): the sorted order, the per-cell runs, the stencil nodes and the node list of that solve stand; only what
// moves with the particles is refreshed.
template <bool REUSE>
__global__ __launch_bounds__(256) void k_ct_prepare(DP p, ContactDev c) {
    const int n = ct_count(c);
    const int N = c.stride;
    if (REUSE && c.st->changed_solve == c.st->solve_no) {
        // the speculation failed: nothing of this solve may run (the host repeats it with the full set-up)
        if (blockIdx.x == 0 && threadIdx.x == 0) c.st->done = CT_DONE_STALE;
        return;
    }
    // (a count the solve may not index with: k_ct_keys has written no keys and no order for it -- what the sort left in
    // `order` is whatever the buffers held.  Round 6: this kernel ran on regardless and formed addresses from it)
    if (c.st->done == CT_DONE_CORRUPT) return;
    const PSet& S = p.set[p.ctl->cur];
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const int k = min((int)(c.order[j] & 0x7FFFFFFFu), N - 1);   // (an index into the per-pair arrays: never outside them)
    const uint32_t key = c.key[j];
    // runs of equal keys
    if (!REUSE && key != CT_NO_CELL) {
        if (j == 0 || c.key[j - 1] != key) c.run[key].x = j;
        if (j == n - 1 || c.key[j + 1] != key) c.run[key].y = j + 1;
    }
    uint32_t b[3];
    contact_base(p, c.pos + (size_t)k * 3, b);
    float fx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        fx[d] = c.pos[k * 3 + d] * p.dxinv - (float)b[d];
        c.cfx[d * N + j] = fx[d];
    }
    const uint32_t slot = c.slot[k];
    c.cmass[j] = fabsf(S.q[0][slot].w) * p.M.density;
    // The 27 stencil cells lie in at most 2 x 2 x 2 blocks: 8 block look-ups (Morton spread + table) instead
    // of 27.  (A contact whose base cell lies outside the active grid cannot be found by the nodes' run
    // lookup: it is left out of the solve altogether; contact points at particle positions, which is what
    // the driver produces, never are.)
    int gn[27];
    if (REUSE) {
#pragma unroll
        for (int t = 0; t < 27; ++t) gn[t] = c.cnode[t * N + j];
    } else {
        const uint32_t bb[3] = {b[0] >> 2, b[1] >> 2, b[2] >> 2};
        uint32_t sx[2], sy[2], sz[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t last = (uint32_t)p.nb - 1u;   // (a neighbour beyond the grid is never selected below)
            sx[h] = spread3(min(bb[0] + h, last)) * 4u;
            sy[h] = spread3(min(bb[1] + h, last)) * 2u;
            sz[h] = spread3(min(bb[2] + h, last));
        }
        int act[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) act[q] = p.lut_act[sx[q >> 2] + sy[(q >> 1) & 1] + sz[q & 1]];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const uint32_t x = b[0] + i, y = b[1] + jj, z = b[2] + l;
                    const int hx = (int)((x >> 2) - bb[0]), hy = (int)((y >> 2) - bb[1]), hz = (int)((z >> 2) - bb[2]);
                    const int q = hx * 4 + hy * 2 + hz;
                    int a = act[0];
#pragma unroll
                    for (int t = 1; t < 8; ++t) a = q == t ? act[t] : a;
                    const int g = (key == CT_NO_CELL || a < 0) ? -1 : a * 64 + (int)(((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u));
                    gn[i * 9 + jj * 3 + l] = g;
                    c.cnode[(i * 9 + jj * 3 + l) * N + j] = g;
                    if (g >= 0) c.node_flag[g] = 1;
                }
    }
    const float nh[3] = {-c.normal[k * 3], -c.normal[k * 3 + 1], -c.normal[k * 3 + 2]};
    float R[9];
    frame_from_normal(nh, R);
    const float4 pv = S.q[1][slot];
    const float rv[3] = {c.rigid_v[k * 3], c.rigid_v[k * 3 + 1], c.rigid_v[k * 3 + 2]};
    const float v0r[3] = {pv.x - rv[0], pv.y - rv[1], pv.z - rv[2]};
    float v0[3];
    mulv3(R, v0r, v0);
#pragma unroll
    for (int t = 0; t < 9; ++t) c.cR[t * N + j] = R[t];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        c.cv0[t * N + j] = v0[t];
        c.crv[t * N + j] = rv[t];
        c.cvel[t * N + j] = c.vel[k * 3 + t];   // contact_vel of CopyContactPairs (first iteration)
    }
    c.cphi0[j] = -c.dist[k];
    // pre-contact velocity at the contact point (what k_ct_gather_vel did as a launch of its own)
    float vg[3];
    gather_from_nodes(p, gn, fx[0], fx[1], fx[2], vg);
    c.vel0[k * 3] = vg[0];
    c.vel0[k * 3 + 1] = vg[1];
    c.vel0[k * 3 + 2] = vg[2];
    }
}

// S3: the nodes that see contacts, in ascending order.  Two steps: the flags of a block (64 ints) become one
// 64-bit word (a ballot per wave, all CUs), then k_ct_node_list turns the words into the list.
__global__ __launch_bounds__(256) void k_ct_flag_bits(DP p, ContactDev c) {
    const int words = (int)p.ctl->n_active;   // one word per active block
    const int lane = threadIdx.x & 63;
    for (int w = (blockIdx.x * 256 + threadIdx.x) >> 6; w < words; w += (gridDim.x * 256) >> 6) {
        const unsigned long long m = __ballot(c.node_flag[(size_t)w * 64 + lane] != 0);
        if (lane == 0) c.flag_bits[w] = m;
    }
}
// The list of the nodes that see contacts, ascending (block word, bit), from the words of k_ct_flag_bits.  A workgroup
// owns 256 words, a wave 64 of them (lane = word); how many nodes the words in front of the workgroup's hold it counts
// itself (all words' popcounts: a few coalesced rows), so no scan kernel and no single workgroup: one workgroup walking
// all words, a word per wave and step, took 19 us for config 3's 1805 words (64 steps per wave and chunk, most of them
// over empty words -- the nodes that see contacts sit in a tenth of the blocks, the floor's).  Here a wave steps
// through its NON-EMPTY words only; lane l owns bit l of the word being written.
__global__ __launch_bounds__(256) void k_ct_node_list(DP p, ContactDev c) {
    __shared__ int s_red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int words = (int)p.ctl->n_active;
    const int start = blockIdx.x * 256;
    if (start >= words && blockIdx.x != 0) return;
    int before = 0, total = 0;
    for (int w0 = tid; w0 < words; w0 += 8 * 256) {   // (eight rows in flight per step)
        unsigned long long x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = w0 + q * 256 < words ? c.flag_bits[w0 + q * 256] : 0ull;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = (int)__popcll(x[q]);
            total += n;
            before += w0 + q * 256 < start ? n : 0;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        before += __shfl_xor(before, d);
        total += __shfl_xor(total, d);
    }
    if (lane == 0) { s_red[0][wv] = before; s_red[1][wv] = total; }
    const int w = start + tid;
    const unsigned long long m = w < words ? c.flag_bits[w] : 0ull;
    const int cnt = (int)__popcll(m);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    __shared__ int s_wt[4];
    if (lane == 63) s_wt[wv] = inc;
    __syncthreads();
    before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    total = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    if (blockIdx.x == 0 && tid == 0) c.st->n_nodes = total;
    int at = before + inc - cnt;
    for (int k = 0; k < wv; ++k) at += s_wt[k];
    unsigned long long nz = __ballot(m != 0ull);
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (nz) {
        const int j = __builtin_ctzll(nz);
        nz &= nz - 1ull;
        const unsigned long long mj = __shfl(m, j);
        const int aj = __shfl(at, j);
        if ((mj >> lane) & 1ull) c.node_list[aj + (int)__popcll(mj & lt)] = (start + wv * 64 + j) * 64 + lane;
    }
}

__global__ __launch_bounds__(256) void k_ct_node_runs(DP p, ContactDev c) {
    const int n_nodes = c.st->n_nodes;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_nodes * 27; t += gridDim.x * 256) {
        const int q = t / 27, o = t % 27;
        const int g = c.node_list[q];
        const int a = g >> 6, cell = g & 63;
        int bx, by, bz;
        block_coords(p.act_block[a], bx, by, bz);
        const int x = bx * 4 + (cell >> 4) - o / 9, y = by * 4 + ((cell >> 2) & 3) - (o / 3) % 3,
                  z = bz * 4 + (cell & 3) - o % 3;
        int2 r = make_int2(0, 0);
        if (x >= 0 && y >= 0 && z >= 0) {
            const int cc = compact_cell(p, (uint32_t)x, (uint32_t)y, (uint32_t)z);
            if (cc >= 0) r = c.run[cc];
        }
        c.node_runs[(size_t)o * c.cap_nodes + q] = r;
    }
}

// velocity at the sorted contact j from its stencil nodes
MPM_DEV void gather_contact_velocity(const DP& p, const ContactDev& c, int j, float* v) {
    int g[27];
#pragma unroll
    for (int n = 0; n < 27; ++n) g[n] = c.cnode[n * c.stride + j];
    gather_from_nodes(p, g, c.cfx[j], c.cfx[c.stride + j], c.cfx[2 * c.stride + j], v);
}

// ---- partitioned domain: node ownership and the exchange of per-node fields in the zones -------
// A node belongs to the rank whose slab holds it; nodes of zone blocks exist on both ranks (same
// values), so global sums count them on their owner only.
MPM_DEV bool node_owned(const DP& p, int g) {
    if (!p.dist.on) return true;
    int bx, by, bz;
    block_coords(p.act_block[g >> 6], bx, by, bz);
    const int gx = bx * 4 + ((g & 63) >> 4);
    return gx >= p.dist.own_lo && gx < p.dist.own_hi;
}

// Buffer: [0] block count, [4 ..) block ids (cap), then cap * 64 * NV float4.  Same shape as the
// halo buffers of the grid sums (mpm_step.h) with NV vectors per cell.
MPM_DEV size_t zone_data_offset(unsigned cap) { return ((size_t)(4 + cap) * 4 + 15) / 16; }
struct ZoneX {
    int lo[2], hi[2];
    uint32_t* buf[2];
};
template <int NV>
__global__ __launch_bounds__(256) void k_zone_pack(DP p, ZoneX z, unsigned cap, const float4* field) {
    const int k = blockIdx.y;
    uint32_t* buf = z.buf[k];
    const unsigned n_active = p.ctl->n_active;
    float4* data = reinterpret_cast<float4*>(buf) + zone_data_offset(cap);
    for (unsigned a = blockIdx.x * 4 + (threadIdx.x >> 6); a < n_active; a += gridDim.x * 4) {
        int bx, by, bz;
        block_coords(p.act_block[a], bx, by, bz);
        if (bx < z.lo[k] || bx > z.hi[k]) continue;   // wave-uniform
        unsigned slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(&buf[0], 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= cap) {
            if ((threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
            continue;
        }
        if ((threadIdx.x & 63) == 0) buf[4 + slot] = p.act_block[a];
        const size_t cell = (size_t)a * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int v = 0; v < NV; ++v) data[((size_t)slot * 64 + (threadIdx.x & 63)) * NV + v] = field[cell * NV + v];
    }
}
template <int NV>
__global__ __launch_bounds__(256) void k_zone_add(DP p, ZoneX z, unsigned cap, float4* field) {
    const uint32_t* buf = z.buf[blockIdx.y];
    const unsigned n = min(buf[0], cap);
    const float4* data = reinterpret_cast<const float4*>(buf) + zone_data_offset(cap);
    for (unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const uint32_t id = buf[4 + e];
        if (id >= p.nblocks) continue;
        const int a = p.lut_act[id];
        if (a < 0) continue;   // nothing of this rank reaches that block
        const size_t cell = (size_t)a * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 r = data[((size_t)e * 64 + (threadIdx.x & 63)) * NV + v];
            float4 q = field[cell * NV + v];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
            field[cell * NV + v] = q;
        }
    }
}
// node_flag (int) <-> the first vector of hg, for the one-off exchange of "this node sees a contact"
__global__ __launch_bounds__(256) void k_ct_flags_to_field(DP p, ContactDev c, int back) {
    const int ncell = (int)p.ctl->n_active * 64;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ncell; g += gridDim.x * 256) {
        if (!back) {
            c.hg[(size_t)g * 3] = make_float4(c.node_flag[g] ? 1.f : 0.f, 0.f, 0.f, 0.f);
            c.hg[(size_t)g * 3 + 1] = c.hg[(size_t)g * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            c.node_flag[g] = c.hg[(size_t)g * 3].x > 0.f ? 1 : 0;
        }
    }
}

// ---- one Newton iteration ------------------------------------------------------

template <int count>
MPM_DEV void wg_reduce_store(double* vals, double* out) {
    // vals: per-thread values; reduces over the workgroup (CT_WG threads) into out[count].  Fully
    // unrolled: `vals` has to stay in registers
    __shared__ double s_red[CT_WG / 64][CT_PART];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < count; ++q) {
        double v = vals[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
        if (lane == 0) s_red[w][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < count) {
        double v = 0;
        for (int q = 0; q < CT_WG / 64; ++q) v += s_red[q][threadIdx.x];
        out[threadIdx.x] = v;
    }
    __syncthreads();
}

// Segments of a tile of sorted contacts: a segment = the contacts of one cell inside the tile.  Wave 0
// fills s_seg[0..nseg] (first contact of every segment, then cnt) and s_cseg[contact] (its segment).
// (`key`: this lane's contact key, loaded by the caller -- tile_key -- so that the load can be in flight early)
MPM_DEV uint32_t tile_key(const ContactDev& c, int lo, int cnt) {
    return threadIdx.x < 64 ? c.key[lo + min((int)threadIdx.x, cnt - 1)] : 0u;
}
// The FIRST tile's keys are requested together with the solver state (one round trip instead of two before the kernel
// knows what to do).  With the contact count on the device that is before the count is known: every lane of wave 0
// reads its own key speculatively (the key arrays are padded by a tile, and a workgroup's first tile lies inside the
// capacity that sized the grid), and once the count is there the lanes past the end take the last contact's key.
// (lo < stride: a launch may have more workgroups than the buffers have tiles -- k_ct_ls always has CT_ROWS_CON --, and
// theirs would lie outside the arrays; the arrays are padded by one tile beyond `stride` entries)
MPM_DEV uint32_t tile_key_early(const ContactDev& c, int lo) {
    return threadIdx.x < 64 && lo < c.stride ? c.key[lo + (int)threadIdx.x] : 0u;
}
MPM_DEV uint32_t tile_key_settle(uint32_t key, int cnt) {
    return threadIdx.x < 64 ? (uint32_t)__shfl((int)key, min((int)threadIdx.x, max(cnt, 1) - 1)) : 0u;
}
MPM_DEV void tile_segments(uint32_t key, int cnt, int* s_seg, int* s_cseg, int* s_nseg) {
    const int tid = threadIdx.x;
    if (tid >= 64) return;
    const uint32_t prev = __shfl_up(key, 1);
    const bool head = tid < cnt && (tid == 0 || key != prev);
    const unsigned long long m = __ballot(head);
    const int before = __popcll(m & ((1ull << tid) - 1ull));   // heads in front of this contact
    if (head) s_seg[before] = tid;
    s_cseg[tid] = before - (head ? 0 : 1);
    if (tid == 0) {
        *s_nseg = __popcll(m);
        s_seg[__popcll(m)] = cnt;
    }
}

// C1 + G1a: one workgroup per tile of CT_TILE sorted contacts.
//  0. Contacts of one cell share their 27 stencil nodes: the nodes of every segment are read once, into
//     LDS (per contact, the 62k x 27 gathers of config 3 kept the L1 tag lookup busy for ~9 us).
//  1. The 4 lanes of a contact gather its velocity from there (9 nodes each; the fourth lane stores the
//     contact's nine 1-D weights meanwhile), all four evaluate the contact's gradient and Hessian
//     (cuda_mpm_kernels.cuh:1107-1154, 1276-1473) in the world frame, and three of them store one
//     16-byte piece of the record (mass-weighted symmetric H[6], G[3]) -- in LDS only.
//  2. Per segment, the sums  H_n = sum w_n^2 H,  G_n = sum w_n G  over its contacts are formed by thread
//     (segment, node, 1 of S interleaved subsets), the subsets added in fixed order, and written to
//     seg_part at the index of the segment's first contact.  k_ct_node_dir adds up the segments of a
//     node's 27 base cells.
// (Per node instead, every record was read 27 times through L2 and only nodes x 16 lanes were busy:
// 31-47 us per Newton iteration on config 3, 62k contacts in 4225 cells.)
// lazy: the step of the previous Newton iteration has not been added to the grid yet (k_ct_node_dir does
// that, after this kernel): velocities are read as v - alpha D.
__global__ __launch_bounds__(256) void k_ct_tile(DP p, ContactDev c, int first, int lazy) {
    // (the first tile's keys are requested together with the solver state: one round trip instead of two before the
    // kernel knows what to do -- the state was written by a single workgroup of the previous kernel and is a miss in
    // seven of the eight L2s)
    uint32_t key_next = tile_key_early(c, (int)blockIdx.x * CT_TILE);
    const int st_done = c.st->done, st_phase = c.st->ls_phase;
    const float st_alpha = c.st->alpha;
    const int cn = ct_count(c), N = c.stride;
    if (!c.force && (st_done || st_phase != 0)) return;   // (line search of the previous direction still running)
    const int n_tiles = (cn + CT_TILE - 1) / CT_TILE;
    key_next = tile_key_settle(key_next, min(CT_TILE, cn - (int)blockIdx.x * CT_TILE));
#if MPM_DIAG
    const unsigned long long tt0 = __builtin_readcyclecounter();
    auto tstamp = [&](int k) {
        if (blockIdx.x == 100 && threadIdx.x == 0 && (diag_flags(p) & 2048)) atomicAdd(&p.dbgbuf[k], (unsigned long long)__builtin_readcyclecounter() - tt0);
    };
#else
    auto tstamp = [&](int) {};
#endif
    __shared__ float s_part[8][CT_SEG_F];
    __shared__ __attribute__((aligned(16))) float4 s_rec[CT_TILE * 3];
    __shared__ __attribute__((aligned(16))) float4 s_nv[CT_STAGE * 27];
    __shared__ float s_w[CT_TILE][9];
    __shared__ int s_seg[CT_TILE + 1];
    __shared__ int s_cseg[CT_TILE];
    __shared__ int s_nseg;
    const float al = lazy ? st_alpha : 0.f;
    const int tid = threadIdx.x, lc = tid >> 2, part = tid & 3;
    const ContactParams cp = {c.dt, c.mu, c.k, c.d, c.epsv};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int lo = tile * CT_TILE, cnt = min(CT_TILE, cn - lo);
        const uint32_t key = key_next;
        if (tile + (int)gridDim.x < n_tiles) {
            const int lo2 = (tile + (int)gridDim.x) * CT_TILE;
            key_next = tile_key(c, lo2, min(CT_TILE, cn - lo2));
        }
        const int jc = min(lc, cnt - 1);   // (lanes past the end repeat the last contact and store nothing)
        const int j = lo + jc;
        // this contact's constants: requested here, used after the stencil nodes have been staged (the loads ride on the
        // staging's round trips instead of following them)
        float R[9], v0[3], crv[3];
#pragma unroll
        for (int t = 0; t < 9; ++t) R[t] = c.cR[t * N + j];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            v0[t] = c.cv0[t * N + j];
            crv[t] = c.crv[t * N + j];
        }
        const float phi0_j = c.cphi0[j], mass = c.cmass[j];
        tile_segments(key, cnt, s_seg, s_cseg, &s_nseg);
        __syncthreads();
        tstamp(0);
        const int nseg = s_nseg;
        float wx[3], wy[3], wz[3];
        bspline3(c.cfx[j], wx);
        bspline3(c.cfx[N + j], wy);
        bspline3(c.cfx[2 * N + j], wz);
        if (part == 3 && lc < cnt) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                s_w[lc][t] = wx[t];
                s_w[lc][3 + t] = wy[t];
                s_w[lc][6 + t] = wz[t];
            }
        }
        float v[3] = {0.f, 0.f, 0.f};
        if (first) {
            v[0] = c.cvel[j]; v[1] = c.cvel[N + j]; v[2] = c.cvel[2 * N + j];
        } else {
            // grid_to_particle_kernel<CONTACT_TRANSFER=true> (cuda_mpm_kernels.cuh:860-866, 891-894):
            // velocity at the contact point from nodes with m > 1e-7
            const int myseg = s_cseg[jc];
            const float wxi = part == 0 ? wx[0] : (part == 1 ? wx[1] : wx[2]);
            for (int s0 = 0; s0 < nseg; s0 += CT_STAGE) {
                const int ns = min(CT_STAGE, nseg - s0);
                for (int t = tid; t < ns * 27; t += 256) {
                    const int sg = t / 27, n = t - sg * 27;
                    const int g = c.cnode[(size_t)n * N + lo + s_seg[s0 + sg]];
                    const float4 q = p.gv[max(g, 0)], D = c.gD[max(g, 0)];
                    const bool ok = g >= 0 && q.w > 1e-7f;
                    s_nv[t] = ok ? make_float4(fmaf(-al, D.x, q.x), fmaf(-al, D.y, q.y), fmaf(-al, D.z, q.z), 1.f)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __syncthreads();
                if (part < 3 && myseg >= s0 && myseg < s0 + ns) {
                    const float4* nv = s_nv + (myseg - s0) * 27 + part * 9;
#pragma unroll
                    for (int m = 0; m < 9; ++m) {
                        const float4 q = nv[m];
                        const float w = q.w != 0.f ? wxi * wy[m / 3] * wz[m % 3] : 0.f;
                        v[0] += w * q.x; v[1] += w * q.y; v[2] += w * q.z;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                v[t] += __shfl_xor(v[t], 1);
                v[t] += __shfl_xor(v[t], 2);
            }
        }
        tstamp(1);
        const float vr[3] = {v[0] - crv[0], v[1] - crv[1], v[2] - crv[2]};
        float vl[3];
        mulv3(R, vr, vl);
        float CH[9], CG[3];
        contact_grad_hess(cp, phi0_j, v0, vl, CH, CG);
        // world frame: R^T G and R^T H R, with the zeros of the contact-frame Hessian (its tangential 2 x 2
        // block and the normal entry are all there is) left out of the products
        float tm[9];   // R^T H
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            tm[i * 3] = R[i] * CH[0] + R[3 + i] * CH[3];
            tm[i * 3 + 1] = R[i] * CH[1] + R[3 + i] * CH[4];
            tm[i * 3 + 2] = R[6 + i] * CH[8];
        }
        float hs[6];   // xx xy xz yy yz zz
        {
            const int ii[6] = {0, 0, 0, 1, 1, 2}, jj2[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
            for (int t = 0; t < 6; ++t)
                hs[t] = mass * (tm[ii[t] * 3] * R[jj2[t]] + tm[ii[t] * 3 + 1] * R[3 + jj2[t]] + tm[ii[t] * 3 + 2] * R[6 + jj2[t]]);
        }
        float WG[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) WG[i] = mass * (R[i] * CG[0] + R[3 + i] * CG[1] + R[6 + i] * CG[2]);
        float4 piece;
        if (part == 0) piece = make_float4(hs[0], hs[1], hs[2], hs[3]);
        else if (part == 1) piece = make_float4(hs[4], hs[5], WG[0], WG[1]);
        else piece = make_float4(WG[2], 0.f, 0.f, 0.f);
        if (lc < cnt && part < 3) s_rec[lc * 3 + part] = piece;
        __syncthreads();
        tstamp(2);
        // ---- sums per segment: S subsets per segment, 8 / S segments at a time ----------------
        int S = 8, lgS = 3;
        while (S > 1 && nseg * S > 8) {
            S >>= 1;
            --lgS;
        }
        const int per_round = 8 >> lgS;
        const int slot = tid >> 5, o = tid & 31;
        const int cl = slot >> lgS, sub = slot & (S - 1);
        // a contact with base cell b reaches node b + (i, j, l) with weight N_i N_j N_l
        const int wi = o / 9, wj = 3 + (o / 3) % 3, wl = 6 + o % 3;
        for (int s0 = 0; s0 < nseg; s0 += per_round) {
            const int sg = s0 + cl;
            if (o < 27 && sg < nseg) {
                float H[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, G[3] = {0.f, 0.f, 0.f};
                const int kb = s_seg[sg + 1];
                for (int k = s_seg[sg] + sub; k < kb; k += S) {
                    const float4 r0 = s_rec[k * 3], r1 = s_rec[k * 3 + 1];
                    const float gz = s_rec[k * 3 + 2].x;
                    const float w = s_w[k][wi] * s_w[k][wj] * s_w[k][wl];
                    const float w2 = w * w;
                    H[0] += w2 * r0.x; H[1] += w2 * r0.y; H[2] += w2 * r0.z; H[3] += w2 * r0.w;
                    H[4] += w2 * r1.x; H[5] += w2 * r1.y;
                    G[0] += w * r1.z; G[1] += w * r1.w; G[2] += w * gz;
                }
                float* d = s_part[slot] + o * 12;
                d[0] = H[0]; d[1] = H[1]; d[2] = H[2];
                d[3] = H[1]; d[4] = H[3]; d[5] = H[4];
                d[6] = H[2]; d[7] = H[4]; d[8] = H[5];
                d[9] = G[0]; d[10] = G[1]; d[11] = G[2];
            }
            __syncthreads();
            for (int e = tid; e < per_round * CT_SEG_F; e += 256) {
                const int c2 = e / CT_SEG_F, ee = e - c2 * CT_SEG_F;
                if (s0 + c2 >= nseg) break;
                float a = s_part[c2 << lgS][ee];
                for (int q = 1; q < S; ++q) a += s_part[(c2 << lgS) + q][ee];
                c.seg_part[(size_t)(lo + s_seg[s0 + c2]) * CT_SEG_F + ee] = a;
            }
            __syncthreads();
        }
        tstamp(3);
        if (blockIdx.x == 100 && threadIdx.x == 0 && (diag_flags(p) & 2048)) atomicAdd(&p.dbgbuf[15], 1ull);
    }
}

// G1b: per node that sees contacts, add up the per-cell sums of k_ct_tile, add the inertia
// term and solve for the Newton direction (cuda_mpm_kernels.cuh:1217-1274).  16 lanes per node:
// the node's contacts are the runs of its 27 neighbour base cells; lane s adds the segment sums of the
// stencil offsets s and s + 16, the 16 partial sums are folded in a fixed butterfly.
// MODE 0: gather and solve in one pass.  Partitioned domain: MODE 1 gathers this rank's contacts into
// c.hg (every cell of it is written: zeros where no contact reaches), the zone exchange adds the
// neighbours' sums, MODE 2 solves from c.hg; |Dir|^2 and the DoF count only include owned nodes.
template <int MODE>
__global__ __launch_bounds__(CT_WG) void k_ct_node_dir(DP p, ContactDev c, int lazy = 0) {
    if (!c.force && (c.st->done || c.st->ls_phase != 0)) return;   // k_ct_decide does not read the records of a finished solve
    double acc[2] = {0, 0};
    const int sub = threadIdx.x & 15;
    // lazy: the accepted step of the previous iteration is still to be added to the grid velocity
    // (k_ct_apply's job otherwise); this kernel is the one that replaces D, so it does that first
    const float al_prev = lazy ? c.st->alpha : 0.f;
    {
        const int n_nodes = c.st->n_nodes;
        const int stride = (gridDim.x * CT_WG) >> 4;
        for (int q = (blockIdx.x * CT_WG + threadIdx.x) >> 4; q < ((n_nodes + 3) & ~3); q += stride) {
            // (a wave holds 4 nodes; all 64 lanes stay in the loop for the shuffles below)
            const bool live = q < n_nodes;
            const int g = live ? c.node_list[q] : 0;
            float H[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, G[3] = {0.f, 0.f, 0.f};
            float4 gq = p.gv[g];
            if (lazy) {
                const float4 Do = c.gD[g];
                gq.x = fmaf(-al_prev, Do.x, gq.x);
                gq.y = fmaf(-al_prev, Do.y, gq.y);
                gq.z = fmaf(-al_prev, Do.z, gq.z);
            }
            if (MODE == 2) {
                if (live && sub == 0) {
                    const float4 h0 = c.hg[(size_t)g * 3], h1 = c.hg[(size_t)g * 3 + 1], h2 = c.hg[(size_t)g * 3 + 2];
                    H[0] = h0.x; H[1] = h0.y; H[2] = h0.z; H[3] = h0.w; H[4] = h1.x; H[5] = h1.y; H[6] = h1.z; H[7] = h1.w;
                    H[8] = h2.x; G[0] = h2.y; G[1] = h2.z; G[2] = h2.w;
                }
            } else if (live && gq.w > 0.f) {
                // lane s adds up the segments of stencil offsets s and s + 16 (27 offsets over 16 lanes):
                // contacts whose base cell is node - (i, j, l) reach this node with weight N_i N_j N_l,
                // already applied by k_ct_tile
                const int o1 = min(sub + 16, 26);
                const int2 ra = c.node_runs[(size_t)sub * c.cap_nodes + q];
                int2 rb = c.node_runs[(size_t)o1 * c.cap_nodes + q];
                if (sub + 16 >= 27) rb = make_int2(0, 0);
                // (a run is cut where it crosses a tile boundary; each piece has its sums at its first contact)
                for (int k = ra.x; k < ra.y; k = (k | (CT_TILE - 1)) + 1) {
                    const float4* part = reinterpret_cast<const float4*>(c.seg_part + ((size_t)k * 27 + sub) * 12);
                    const float4 a0 = part[0], a1 = part[1], a2 = part[2];
                    H[0] += a0.x; H[1] += a0.y; H[2] += a0.z; H[3] += a0.w;
                    H[4] += a1.x; H[5] += a1.y; H[6] += a1.z; H[7] += a1.w;
                    H[8] += a2.x; G[0] += a2.y; G[1] += a2.z; G[2] += a2.w;
                }
                for (int k = rb.x; k < rb.y; k = (k | (CT_TILE - 1)) + 1) {
                    const float4* part = reinterpret_cast<const float4*>(c.seg_part + ((size_t)k * 27 + o1) * 12);
                    const float4 a0 = part[0], a1 = part[1], a2 = part[2];
                    H[0] += a0.x; H[1] += a0.y; H[2] += a0.z; H[3] += a0.w;
                    H[4] += a1.x; H[5] += a1.y; H[6] += a1.z; H[7] += a1.w;
                    H[8] += a2.x; G[0] += a2.y; G[1] += a2.z; G[2] += a2.w;
                }
            }
            // fold the 16 lanes of the node (xor butterfly inside a row of 16)
            if (MODE != 2) {
#pragma unroll
                for (int d = 8; d >= 1; d >>= 1) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) H[t] += __shfl_xor(H[t], d, 16);
#pragma unroll
                    for (int t = 0; t < 3; ++t) G[t] += __shfl_xor(G[t], d, 16);
                }
            }
            if (!live || sub != 0) continue;
            if (MODE == 1) {
                c.hg[(size_t)g * 3] = make_float4(H[0], H[1], H[2], H[3]);
                c.hg[(size_t)g * 3 + 1] = make_float4(H[4], H[5], H[6], H[7]);
                c.hg[(size_t)g * 3 + 2] = make_float4(H[8], G[0], G[1], G[2]);
                continue;
            }
            float4 D = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gq.w > 0.f) {
                float hn = 0.f, gn = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) hn += H[t] * H[t];
#pragma unroll
                for (int t = 0; t < 3; ++t) gn += G[t] * G[t];
                if ((double)sqrtf(hn) > 1e-7 || (double)sqrtf(gn) > 1e-7) {
                    const float4 vs = p.gvs[g];
                    H[0] -= gq.w; H[4] -= gq.w; H[8] -= gq.w;
                    G[0] -= gq.w * (gq.x - vs.x);
                    G[1] -= gq.w * (gq.y - vs.y);
                    G[2] -= gq.w * (gq.z - vs.z);
                    float Hi[9], d[3];
                    inv33(H, Hi);
                    mulv3(Hi, G, d);
                    if (node_owned(p, g)) {
                        acc[0] += (double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        acc[1] += 1.0;
                    }
                    D = make_float4(d[0] * c.relax, d[1] * c.relax, d[2] * c.relax, 1.f);
                }
            }
            c.gD[g] = D;
            if (lazy && gq.w > 0.f) p.gv[g] = gq;
        }
    }
    wg_reduce_store<2>(acc, c.part_dir + (size_t)blockIdx.x * 2);
}

// C2 + G2: line-search energies for every candidate step: contact part
// (cuda_mpm_kernels.cuh:1276-1473 with global_line_search = true) in workgroups [0, n_con_wg),
// inertia part (cuda_mpm_kernels.cuh:1536-1589) in the rest
// exact: 0 = backtracking (all candidate steps at once); 1 = (E, dE, d2E) at the step given by the
// host; 2 = the same at st->alpha_probe, the device-resident search (skipped once the step is decided)
__global__ __launch_bounds__(CT_WG) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ct_ls(DP p, ContactDev c, int n_con_wg, int exact,
                                                                                             float alpha_probe) {
    // (as in k_ct_tile: the first tile's keys are requested together with the solver state)
    uint32_t key_next = (int)blockIdx.x < n_con_wg ? tile_key_early(c, (int)blockIdx.x * CT_TILE) : 0u;
    const int st_done = c.st->done, st_phase = c.st->ls_phase;
    const float st_probe = c.st->alpha_probe;
    const int cn = ct_count(c), N = c.stride;
    if (st_done && !c.force) return;   // k_ct_decide does not read the records of a finished solve
    const int n_tiles = (cn + CT_TILE - 1) / CT_TILE;
    if ((int)blockIdx.x < n_con_wg) key_next = tile_key_settle(key_next, min(CT_TILE, cn - (int)blockIdx.x * CT_TILE));
    if (exact == 2) {
        if (st_phase == 3) return;
        alpha_probe = st_probe;
    }
    const bool deep = !exact && st_phase == 4;
    if ((int)blockIdx.x < n_con_wg) {
        // A tile of CT_TILE sorted contacts at a time, 4 lanes per contact: the stencil nodes of every
        // segment are staged in LDS once (see k_ct_tile), each lane adds up 9 of the 27 and evaluates its
        // share of the candidate steps
        __shared__ double s_red[CT_WG / 64][CT_PART];
        __shared__ __attribute__((aligned(16))) float4 s_nv[CT_STAGE * 27];
        __shared__ __attribute__((aligned(16))) float4 s_nD[CT_STAGE * 27];
        __shared__ int s_seg[CT_TILE + 1];
        __shared__ int s_cseg[CT_TILE];
        __shared__ int s_nseg;
        const int tid = threadIdx.x, lc = tid >> 2;
        const int part = threadIdx.x & 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const ContactParams cp = {c.dt, c.mu, c.k, c.d, c.epsv};
        double acc[6] = {0, 0, 0, 0, 0, 0}, e0 = 0;
        for (int tile = blockIdx.x; tile < n_tiles; tile += n_con_wg) {
            const int lo = tile * CT_TILE, cnt = min(CT_TILE, cn - lo);
            const uint32_t key = key_next;
            if (tile + n_con_wg < n_tiles) {
                const int lo2 = (tile + n_con_wg) * CT_TILE;
                key_next = tile_key(c, lo2, min(CT_TILE, cn - lo2));
            }
            __syncthreads();   // the previous tile's tables are no longer read
            tile_segments(key, cnt, s_seg, s_cseg, &s_nseg);
            __syncthreads();
            const int nseg = s_nseg;
            const int jc = min(lc, cnt - 1), j = lo + jc;
            const bool counted = lc < cnt;   // (lanes past the end repeat the last contact and add nothing)
            float wx[3], wy[3], wz[3];
            bspline3(c.cfx[j], wx);
            bspline3(c.cfx[N + j], wy);
            bspline3(c.cfx[2 * N + j], wz);
            const float wxi = part == 0 ? wx[0] : (part == 1 ? wx[1] : wx[2]);
            const int myseg = s_cseg[jc];
            float ov[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < nseg; s0 += CT_STAGE) {
                const int ns = min(CT_STAGE, nseg - s0);
                for (int t = tid; t < ns * 27; t += CT_WG) {
                    const int sg = t / 27, n = t - sg * 27;
                    const int g = c.cnode[(size_t)n * N + lo + s_seg[s0 + sg]];
                    float4 q = p.gv[max(g, 0)], D = c.gD[max(g, 0)];
                    D.w = g >= 0 ? 1.f : 0.f;
                    s_nv[t] = q;
                    s_nD[t] = D;
                }
                __syncthreads();
                if (part < 3 && myseg >= s0 && myseg < s0 + ns) {
                    const int at = (myseg - s0) * 27 + part * 9;
#pragma unroll
                    for (int m = 0; m < 9; ++m) {
                        const float4 q = s_nv[at + m], D = s_nD[at + m];
                        const float w = D.w != 0.f ? wxi * wy[m / 3] * wz[m % 3] : 0.f;
                        ov[0] += w * q.x; ov[1] += w * q.y; ov[2] += w * q.z;
                        dd[0] += w * D.x; dd[1] += w * D.y; dd[2] += w * D.z;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ov[t] += __shfl_xor(ov[t], 1);
                ov[t] += __shfl_xor(ov[t], 2);
                dd[t] += __shfl_xor(dd[t], 1);
                dd[t] += __shfl_xor(dd[t], 2);
            }
            float R[9], v0[3];
#pragma unroll
            for (int t = 0; t < 9; ++t) R[t] = c.cR[t * N + j];
#pragma unroll
            for (int t = 0; t < 3; ++t) v0[t] = c.cv0[t * N + j];
            const float phi0 = c.cphi0[j], mass = counted ? c.cmass[j] : 0.f;
            float ovl[3], ddl[3];
            {
                const float t[3] = {ov[0] - c.crv[j], ov[1] - c.crv[N + j], ov[2] - c.crv[2 * N + j]};
                mulv3(R, t, ovl);
                mulv3(R, dd, ddl);
            }
            if (!exact) {
                // shallow pass: lane `part` tries alpha = 2^-part; deep pass (none of those four was
                // accepted): alpha = 2^-(4 + 6 part + m), m = 0..5
                if (!deep) {
                    if (part == 0) e0 += (double)(mass * contact_cost(cp, phi0, v0, ovl));
                    const float al = ldexpf(1.f, -part);
                    const float nv[3] = {ovl[0] - al * ddl[0], ovl[1] - al * ddl[1], ovl[2] - al * ddl[2]};
                    acc[0] += (double)(mass * contact_cost(cp, phi0, v0, nv));
                } else {
                    float al = ldexpf(1.f, -(LS_SHALLOW + 6 * part));
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        const float nv[3] = {ovl[0] - al * ddl[0], ovl[1] - al * ddl[1], ovl[2] - al * ddl[2]};
                        acc[m] += (double)(mass * contact_cost(cp, phi0, v0, nv));
                        al *= .5f;
                    }
                }
            } else if (part == 0) {
                const float al = alpha_probe;
                const float nv[3] = {ovl[0] - al * ddl[0], ovl[1] - al * ddl[1], ovl[2] - al * ddl[2]};
                float CH[9], CG[3];
                contact_grad_hess(cp, phi0, v0, nv, CH, CG);
                float t[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) t[a] = ddl[0] * CH[a] + ddl[1] * CH[3 + a] + ddl[2] * CH[6 + a];
                acc[0] += (double)(mass * contact_cost(cp, phi0, v0, nv));
                acc[1] += (double)(mass * dot3(CG, ddl));
                acc[2] += (double)(mass * dot3(t, ddl));
            }
        }
        // lanes with the same `part` are added up (fixed butterfly), then the 4 waves
        for (int q = threadIdx.x; q < (CT_WG / 64) * CT_PART; q += CT_WG) (&s_red[0][0])[q] = 0.0;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            if (!exact && !deep && m > 0) break;
            double v = acc[m];
#pragma unroll
            for (int d = 4; d <= 32; d <<= 1) v += __shfl_xor(v, d);
            const int at = exact ? (lane == 0 ? m : LS_CAND + 1) : (deep ? LS_SHALLOW + lane * 6 + m : lane);
            if (lane < 4) s_red[wv][at] = v;
        }
        {
            double v = e0;
#pragma unroll
            for (int d = 4; d <= 32; d <<= 1) v += __shfl_xor(v, d);
            if (lane == 0) s_red[wv][LS_CAND] = v;
        }
        __syncthreads();
        if (threadIdx.x <= LS_CAND) {
            double v = 0;
            for (int q = 0; q < CT_WG / 64; ++q) v += s_red[q][threadIdx.x];
            c.part[(size_t)blockIdx.x * CT_PART + threadIdx.x] = v;
        }
    } else {
    // inertia part.  Only nodes that see contacts take part: elsewhere the direction is zero and v == v*
    // bit for bit (k_grid writes both from the same registers), so every energy term is an exact zero.
    double acc[LS_CAND - LS_SHALLOW], e0 = 0;
#pragma unroll
    for (int q = 0; q < LS_CAND - LS_SHALLOW; ++q) acc[q] = 0;
    const int n_nodes = c.st->n_nodes;
    const int b = (int)blockIdx.x - n_con_wg, nb = (int)gridDim.x - n_con_wg;
    for (int i = b * CT_WG + threadIdx.x; i < n_nodes; i += nb * CT_WG) {
        const int g = c.node_list[i];
        const float4 q = p.gv[g];
        if (!(q.w > 0.f) || !node_owned(p, g)) continue;
        const float4 vs = p.gvs[g], D = c.gD[g];
        const float o[3] = {q.x - vs.x, q.y - vs.y, q.z - vs.z};
        if (!exact) {
            if (!deep) {
                e0 += (double)(.5f * q.w * (o[0] * o[0] + o[1] * o[1] + o[2] * o[2]));
                float al = 1.f;
#pragma unroll
                for (int k = 0; k < LS_SHALLOW; ++k) {
                    const float n0 = o[0] - al * D.x, n1 = o[1] - al * D.y, n2 = o[2] - al * D.z;
                    acc[k] += (double)(.5f * q.w * (n0 * n0 + n1 * n1 + n2 * n2));
                    al *= .5f;
                }
            } else {
                float al = ldexpf(1.f, -LS_SHALLOW);
#pragma unroll
                for (int k = 0; k < LS_CAND - LS_SHALLOW; ++k) {
                    const float n0 = o[0] - al * D.x, n1 = o[1] - al * D.y, n2 = o[2] - al * D.z;
                    acc[k] += (double)(.5f * q.w * (n0 * n0 + n1 * n1 + n2 * n2));
                    al *= .5f;
                }
            }
        } else {
            const float al = alpha_probe;
            const float n0 = o[0] - al * D.x, n1 = o[1] - al * D.y, n2 = o[2] - al * D.z;
            acc[0] += (double)(.5f * q.w * (n0 * n0 + n1 * n1 + n2 * n2));
            acc[1] += (double)(-q.w * (n0 * D.x + n1 * D.y + n2 * D.z));
            acc[2] += (double)(q.w * (D.x * D.x + D.y * D.y + D.z * D.z));
        }
    }
    double* row = c.part + (size_t)(CT_ROWS_CON + b) * CT_PART;
    if (exact) {
        wg_reduce_store<3>(acc, row);
    } else if (!deep) {
        wg_reduce_store<LS_SHALLOW>(acc, row);
        wg_reduce_store<1>(&e0, row + LS_CAND);
    } else {
        wg_reduce_store<LS_CAND - LS_SHALLOW>(acc, row + LS_SHALLOW);
    }
    }
}

// S: fixed-order sum of the partial records, choice of the step, convergence test
// (cuda_mpm_solver.cu:472-528, 567-570).  1024 threads: thread (r, e) sums entry e of the rows
// r, r + 32, ... of every kind; wave 0 adds the 32 row groups in order and decides.
// `phase` (partitioned domain): 0 = sum and decide in one go; 1 = only leave this rank's sums in
// st->red[0..31] (the host all-reduces them); 2 = decide from the global sums put back into st->red.
// The decision of a Newton iteration from the partial sums (one workgroup of NT threads: the kernel
// below, or the last workgroup of k_ct_ls to finish).
// this thread's share of the partial rows: v = its entry of its rows, (d0, d1) = its (|Dir|^2, DoFs) records.  Every
// load is issued before the first is waited for (one round trip; the rows hold numbers in any state of the solve, so
// k_ct_decide calls this BEFORE it knows whether the solve is over or which pass this is, see there)
template <int NT>
MPM_DEV void ct_thread_sums(const ContactDev& c, int n_dir_wg, int n_con_wg, int n_grid_wg, int exact, bool deep_pass, double& v,
                            double& d0, double& d1) {
    // Only the entries this pass decides on are summed (the others hold sums of an earlier pass):
    // 4 candidates + E(0) in the shallow pass, 24 in the deep one, (E, dE, d2E) for the exact search.
    // Thread (slot, r) adds the slot's entry of rows r, r + RG, ... (contact rows first, then node
    // rows), 16 loads in flight at a time; wave 0 adds the RG row groups in order and decides.
    const int lgNE = exact ? 2 : (deep_pass ? 5 : 3);
    const int NE = 1 << lgNE, RG = NT >> lgNE;
    const int slot = threadIdx.x & (NE - 1), r = threadIdx.x >> lgNE;
    int e = -1;   // entry of this slot
    if (exact) e = slot < 3 ? slot : -1;
    else if (deep_pass) e = slot >= LS_SHALLOW && slot < LS_CAND ? slot : -1;
    else e = slot < LS_SHALLOW ? slot : (slot == LS_SHALLOW ? LS_CAND : -1);
    // the (|Dir|^2, DoFs) records of k_ct_node_dir: thread t adds records t, t + NT, ...
    d0 = 0; d1 = 0;
    double dq[2][2] = {{0, 0}, {0, 0}};
    static_assert(CT_DIR_WG <= 2 * 1024, "two records per thread at most");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int w = (int)threadIdx.x + k * NT;
        if (w < n_dir_wg) {
            dq[k][0] = c.part_dir[(size_t)w * 2];
            dq[k][1] = c.part_dir[(size_t)w * 2 + 1];
        }
    }
    v = 0;
    if (e >= 0) {
        const int rows = n_con_wg + n_grid_wg;
        for (int k0 = r; k0 < rows; k0 += 16 * RG) {
            double t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int w = k0 + k * RG;
                const int row = w < n_con_wg ? w : CT_ROWS_CON + (w - n_con_wg);
                t[k] = w < rows ? c.part[(size_t)row * CT_PART + e] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) v += t[k];
        }
    }
    for (int w = (int)threadIdx.x + 2 * NT; w < n_dir_wg; w += NT) {   // (never with today's CT_DIR_WG)
        dq[0][0] += c.part_dir[(size_t)w * 2];
        dq[0][1] += c.part_dir[(size_t)w * 2 + 1];
    }
    d0 = dq[0][0] + dq[1][0];
    d1 = dq[0][1] + dq[1][1];
}

// One thread leaves the state of the solve where the host polls it (ContactMailbox): no fence -- every word is one 8-byte
// store and carries the publication number, the host accepts four words with the same number.
MPM_DEV void ct_publish(const ContactDev& c) {
    ContactState* st = c.st;
    const unsigned seq = st->seq + 1u;
    st->seq = seq;
    const int done = st->done;
    const unsigned long long hi = (unsigned long long)seq << 32;
    const unsigned cnt = (unsigned)(done == CT_DONE_FAULT || done == CT_DONE_CORRUPT ? st->n_wanted : st->n);
    const unsigned nodes = ((unsigned)st->n_nodes & 0x7FFFFFFFu) | (st->changed_solve != st->solve_no ? 0x80000000u : 0u);
    unsigned long long* w = c.mbox->w;
    __hip_atomic_store(&w[1], hi | (unsigned long long)__float_as_uint(st->residual), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&w[2], hi | (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&w[3], hi | (unsigned long long)nodes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&w[4], hi | (unsigned long long)st->n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    {
        const Ctl* k = c.ctl;
        const float left = k->need_rebuild || k->error ? 0.f : fmaxf(0.f, k->quiet_time - k->time_since_resort);
        __hip_atomic_store(&w[5], hi | (unsigned long long)__float_as_uint(left), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&w[0], hi | ((unsigned long long)(done & 0xFF) << 24) | (unsigned long long)((unsigned)st->iters & 0xFFFFFFu),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the fields of the solver state the backtracking decision reads and updates, fetched once at the start of the kernel
// (read where they are used, each is one more dependent round trip of the single workgroup that decides)
struct CtSnap {
    int ls_phase, iters, ls_total;
    float E0;
};
MPM_DEV CtSnap ct_snapshot(const ContactState* st) { return CtSnap{st->ls_phase, st->iters, st->ls_total, st->E0}; }

// phase: 0 = sum and decide in one go; partitioned domain: 1 = only leave this rank's sums in st->red (the host, or
// ncclAllReduce on the engine's stream, adds the ranks' up); 2 = decide from the global sums put back into st->red;
// TEAM transport (mpm_team.h): 4 = this rank's sums straight into every rank's slot (+ flags); 3 = wait for all ranks'
// sums, add them in rank order, decide.
template <int NT>
MPM_DEV void ct_decide_from(const ContactDev& c, const TeamDev& tm, int exact, int phase, bool deep_pass, double v, double d0, double d1,
                            double (*s_sum)[CT_PART], double (*s_dir)[2], CtSnap sn) {
    ContactState* st = c.st;
    // entries 0..28: energies; 29: norm_dir; 30: dofs
    if (phase == 2) {
        if (threadIdx.x >= 64) return;
        v = threadIdx.x < 32 ? st->red[threadIdx.x] : 0.0;
    } else if (phase == 3) {
        if (threadIdx.x >= 64) return;
        bool ok = true;
        v = team_collect_sums(tm, const_cast<Ctl*>(c.ctl), &ok);
        if (!ok) {   // a rank never arrived (MPM_ERR_HALO is raised): the solve ends here, nothing waits again
            if (threadIdx.x == 0) st->done = 1;
            return;
        }
    } else {
    const int lgNE = exact ? 2 : (deep_pass ? 5 : 3);
    const int RG = NT >> lgNE;
    double* s_flat = &s_sum[0][0];   // NT doubles
    s_flat[threadIdx.x] = v;   // [r][slot]
    // fixed tree over the workgroup for the direction records
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            d0 += __shfl_down(d0, d);
            d1 += __shfl_down(d1, d);
        }
        if ((threadIdx.x & 63) == 0) {
            s_dir[threadIdx.x >> 6][0] = d0;
            s_dir[threadIdx.x >> 6][1] = d1;
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    // lane = entry from here on
    v = 0;
    if (threadIdx.x <= LS_CAND) {
        int sl = -1;   // the slot that carries this entry
        if (exact) sl = threadIdx.x < 3 ? (int)threadIdx.x : -1;
        else if (deep_pass) sl = threadIdx.x >= LS_SHALLOW && threadIdx.x < LS_CAND ? (int)threadIdx.x : -1;
        else sl = threadIdx.x < LS_SHALLOW ? (int)threadIdx.x : (threadIdx.x == LS_CAND ? LS_SHALLOW : -1);
        if (sl >= 0)
            for (int q = 0; q < RG; ++q) v += s_flat[(q << lgNE) + sl];
    } else if (threadIdx.x <= LS_CAND + 2) {
        for (int q = 0; q < NT / 64; ++q) v += s_dir[q][threadIdx.x - LS_CAND - 1];
    }
    if (phase == 1) {
        if (threadIdx.x < 32) st->red[threadIdx.x] = v;
        return;
    }
    if (phase == 4) {
        team_push_sums(tm, threadIdx.x < 32 ? v : 0.0);
        return;
    }
    }
    const int lane = threadIdx.x;
    if (lane == LS_CAND + 1) st->norm_dir_sq = (float)v;
    if (lane == LS_CAND + 2) st->dofs = (float)v;
    if (exact == 2) {
        // device-resident exact search: one thread advances the state machine of cuda_mpm_solver.cu:383-471
        if (lane < 3) st->scal[lane] = v;
        const float E = __shfl((float)v, 0), dE = __shfl((float)v, 1), d2E = __shfl((float)v, 2);
        if (lane != 0 || st->ls_phase == 3) return;
        if (st->ls_phase == 0) {
            st->f_lo[0] = E; st->f_lo[1] = dE; st->f_lo[2] = d2E;
            st->E0 = E;
            st->alpha_probe = 1.f;
            st->ls_phase = 1;
        } else if (st->ls_phase == 1) {
            float x_lo = 0.f, f_lo = st->f_lo[1];
            if (f_lo < 0.f && dE < 0.f) {   // both slopes negative: alpha = 1 is optimal (:395-398)
                x_lo = 1.f;
                f_lo = dE;
            }
            const float f_tol = 1e-8f, x_tol = f_tol * c.relax;
            st->rf.start(x_lo, f_lo, 1.f, dE, 1.f, x_tol, f_tol, 200, RF_SIGN3 | RF_NO_ENDS | RF_STEP_LAST);
            st->alpha_probe = st->rf.root;
            st->ls_phase = 2;
        } else {
            st->energy = E;
            st->rf.feed(dE, d2E);
            if (st->rf.status != 0) {
                st->alpha = st->rf.root;
                st->ls_evals = st->rf.evals;
                st->ls_phase = 3;
            } else {
                st->alpha_probe = st->rf.root;
            }
        }
        return;
    }
    if (exact) {
        if (lane < 3) st->scal[lane] = v;
        return;
    }
    const float en = (float)v;
    const bool deep = sn.ls_phase == 4;
    const float E0 = deep ? sn.E0 : __shfl(en, LS_CAND);
    const int lo = deep ? LS_SHALLOW : 0, hi = deep ? LS_CAND : LS_SHALLOW;
    const unsigned long long ok = __ballot(lane >= lo && lane < hi && en <= E0);
    if (!ok && !deep) {
        // (cuda_mpm_solver.cu's loop halves alpha until the energy does not increase: the rest of the
        // candidates are evaluated by the next pass, the iteration is not over)
        if (lane == 0) {
            st->E0 = E0;
            st->ls_phase = 4;
        }
        return;
    }
    int j = ok ? __builtin_ctzll(ok) : LS_CAND - 1;  // "Tiny Alpha": accept 2^-27 anyway
    const float Ej = __shfl(en, j);
    const float nd = __shfl(en, LS_CAND + 1), dofs = __shfl(en, LS_CAND + 2);
    if (lane == 0) {
        st->ls_phase = 0;
        st->alpha = ldexpf(1.f, -j);
        st->energy = Ej;
        st->E0 = E0;
        st->ls_total = sn.ls_total + j + 1;
        st->iters = sn.iters + 1;
        const float res = sqrtf(nd) / dofs;   // NaN when there is no DoF: the loop stops, as in the reference
        st->residual = res;
        if (!(res > c.tol) || sn.iters + 1 >= c.max_iters) st->done = 2;  // finish after this update
        if (sn.iters < CT_LOG) {
            // the host-side decisions of cuda_mpm_solver.cu:472-528, 567-570, one row per Newton iteration
            float* row = c.it_log + (size_t)sn.iters * CT_LOG_F;
            row[0] = res; row[1] = (float)(j + 1); row[2] = Ej; row[3] = ldexpf(1.f, -j); row[4] = E0; row[5] = nd; row[6] = dofs;
            row[7] = 0.f;
        }
    }
}

__global__ __launch_bounds__(1024) void k_ct_decide(ContactDev c, int n_dir_wg, int n_con_wg, int n_grid_wg, int exact,
                                                    int phase = 0, TeamDev tm = TeamDev{}) {
    __shared__ double s_sum[32][CT_PART];
    __shared__ double s_dir[16][2];
    ContactState* st = c.st;
    // One workgroup, nothing but dependent round trips: the state, then the rows, then the direction records.  So the
    // rows are read BEFORE the state has arrived, as if this were the usual pass (not finished, not the deep pass of
    // the backtracking): the loads are harmless in any state, and the rare other case reads again.
    const int done = st->done;
    const CtSnap sn = ct_snapshot(st);
    const int ls_phase = sn.ls_phase;
    double v = 0, d0 = 0, d1 = 0;
    const bool sums_here = phase != 2 && phase != 3;
    if (sums_here) ct_thread_sums<1024>(c, n_dir_wg, n_con_wg, n_grid_wg, exact, false, v, d0, d1);
    if (done && !c.force) {
        // "finish after this update" becomes "finished" once that update (k_ct_apply of the
        // previous iteration) has run
        if (threadIdx.x == 0) {
            if (done == 2) st->done = 1;
            if (c.mbox) ct_publish(c);
        }
        return;
    }
    const bool deep_pass = !exact && ls_phase == 4;
    if (deep_pass && sums_here) ct_thread_sums<1024>(c, n_dir_wg, n_con_wg, n_grid_wg, exact, true, v, d0, d1);
    ct_decide_from<1024>(c, tm, exact, phase, deep_pass, v, d0, d1, s_sum, s_dir, sn);
    // (thread 0 is the lane that took the decision: its own stores to the state precede this in program order)
    if (threadIdx.x == 0 && c.mbox) ct_publish(c);
}

// G3: v -= alpha Dir (cuda_mpm_kernels.cuh:1591-1614); only nodes that see contacts have a direction
// mode 0: after the decision of every iteration (nothing once the solve is finished); 1: device-resident
// exact search (only when the search of this direction has ended); 2: once after the loop, for the step
// of the last iteration when k_ct_node_dir applies the others (lazy)
__global__ __launch_bounds__(CT_WG) void k_ct_apply(DP p, ContactDev c, int mode = 0) {
    ContactState* st = c.st;
    if (st->done >= CT_DONE_FAULT) return;        // (nothing was solved)
    if (mode != 2 && st->done == 1) return;
    if (mode == 0 && st->ls_phase == 4) return;   // backtracking continues: no step accepted yet
    if (mode == 1 && st->ls_phase != 3) return;   // the search of this direction is still running
    const float al = st->alpha;
    const int n_nodes = st->n_nodes;
    for (int q = blockIdx.x * CT_WG + threadIdx.x; q < n_nodes; q += gridDim.x * CT_WG) {
        const int g = c.node_list[q];
        float4 v = p.gv[g];
        if (!(v.w > 0.f)) continue;
        const float4 D = c.gD[g];
        v.x = fmaf(-al, D.x, v.x); v.y = fmaf(-al, D.y, v.y); v.z = fmaf(-al, D.z, v.z);
        p.gv[g] = v;
    }
}

// Device-resident exact search: closes a Newton iteration after k_ct_apply (cuda_mpm_solver.cu:567-570)
__global__ void k_ct_exact_finish(ContactDev c) {
    ContactState* st = c.st;
    if (threadIdx.x != 0) return;
    if (!(st->done != 0 || st->ls_phase != 3)) {
        st->residual = sqrtf(st->norm_dir_sq) / st->dofs;
        if (st->iters < CT_LOG) {
            float* row = c.it_log + (size_t)st->iters * CT_LOG_F;
            row[0] = st->residual; row[1] = (float)st->ls_evals; row[2] = st->energy; row[3] = st->alpha; row[4] = st->E0;
            row[5] = st->norm_dir_sq; row[6] = st->dofs; row[7] = 0.f;
        }
        st->iters += 1;
        st->ls_total += st->ls_evals;
        st->ls_phase = 0;
        st->alpha_probe = 0.f;   // the next direction's search starts with (E, dE, d2E) at alpha = 0 (cuda_mpm_solver.cu:386-390)
        if (!(st->residual > c.tol) || st->iters >= c.max_iters) st->done = 1;
    }
    if (c.mbox) ct_publish(c);   // (the host learns "finished" from this launch, not from the next pattern's first decision)
}

// apply_contact_impulse_to_rigid_bodies (cuda_mpm_kernels.cuh:1616-1658).  The reference adds six floats per contact to
// the per-body accumulators with float atomics: sums whose last bits depend on the order of arrival.  Here every
// contribution is converted to 64-bit fixed point (the momentum scale of ParticleToGrid's tiles, DP::fix_p: total mass x
// 2^-47 of resolution) and added as an INTEGER -- per workgroup in LDS first (bodies 0..31; thousands of atomics on one
// address serialise at the memory side, and ds_add_u64 costs a twentieth of a float LDS atomic on this part), then once
// per workgroup and component to the accumulators: exact sums, the same bits whatever the order (round 6: two engines fed
// the same solve report the same impulses to the bit, which round 5's float atomics only did by luck of the scheduler).
constexpr int CT_LDS_BODIES = 32;
MPM_DEV long long imp_fixed(double v, double scale, bool& bad) {
    const double q = v * scale;
    bad |= !(fabs(q) < 0x1p62);   // (NaN included)
    return fabs(q) < 0x1p62 ? __double2ll_rn(q) : 0ll;
}
// (with the contact velocities after the solve, contact_vel, gathered here: k_ct_gather_vel's job)
__global__ __launch_bounds__(256) void k_ct_impulse(DP p, ContactDev c) {
    __shared__ unsigned long long s_acc[CT_LDS_BODIES][6];
    for (int q = threadIdx.x; q < CT_LDS_BODIES * 6; q += 256) (&s_acc[0][0])[q] = 0ull;
    __syncthreads();
    const int cn = c.st->done >= CT_DONE_FAULT ? 0 : ct_count(c);   // (a solve that did not run leaves no impulse: the host repeats it)
    bool bad = false;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < cn; j += gridDim.x * 256) {
        const int k = min((int)(c.order[j] & 0x7FFFFFFFu), c.stride - 1);   // sorted position j is the caller's contact k
        const float m = c.cmass[j];
        float v[3];
        gather_contact_velocity(p, c, j, v);
        float l[3], r[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c.vel[k * 3 + a] = v[a];
            l[a] = m * -(v[a] - c.vel0[k * 3 + a]);
            r[a] = c.pos[k * 3 + a] - c.p_WB[k * 3 + a];
        }
        const float h[3] = {r[1] * l[2] - l[1] * r[2], r[2] * l[0] - l[2] * r[0], r[0] * l[1] - l[0] * r[1]};
        const uint32_t b = c.body[k];
        if (b >= (uint32_t)c.n_bodies) continue;
        unsigned long long* dst = b < (uint32_t)CT_LDS_BODIES ? &s_acc[b][0] : reinterpret_cast<unsigned long long*>(c.body_acc + (size_t)b * 6);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned long long qh = (unsigned long long)imp_fixed((double)h[a], c.imp_fix, bad);
            const unsigned long long ql = (unsigned long long)imp_fixed((double)l[a], c.imp_fix, bad);
            if (b < (uint32_t)CT_LDS_BODIES) {
                __hip_atomic_fetch_add(dst + a, qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dst + 3 + a, ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_fetch_add(dst + a, qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(dst + 3 + a, ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_RANGE);
    __syncthreads();
    for (int q = threadIdx.x; q < min(c.n_bodies, CT_LDS_BODIES) * 6; q += 256) {
        const unsigned long long v = (&s_acc[0][0])[q];
        if (v != 0ull)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(c.body_acc) + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace mpm
