// Partitioned domain ("particle domains tile across the GPUs", particle migration):
// the kernels that hand particles from rank to rank.  See struct Dist in mpm_device.h for the model.
//
// Every `migrate_every` substeps, before the re-sort, a rank
//   1. classifies its active particles by the x cell of their base node (k_dist_classify):
//        owned, now in a neighbour's slab   -> record (role OWNED) to that neighbour; here the
//                                              particle becomes a ghost, or is released if it is
//                                              already beyond this rank's ghost band;
//        owned, newly inside the band the neighbour keeps ghosts in -> record (role GHOST);
//        ghost, outside this rank's band    -> released;
//   2. exchanges the two record buffers with its neighbours (transport: the caller / mpm_chain);
//   3. applies what it received (k_dist_apply): a particle it does not hold yet is appended behind
//      the active ones, one it holds (a ghost being promoted) is overwritten in place;
//   4. re-sorts (forced), which drops the released particles and merges the appended ones.
// Ghost copies are never refreshed in between: they take part in FEM and G2P with the same inputs
// and the same arithmetic as the owner's copy, so they stay bit-identical.
//
// Record = 9 x 16 bytes: (original id, role, F8, 0), q0..q3 (|vol| in q0.w, the authoritative C8 of a face in q1.w),
// F0..3, F4..7, and the particle's piece of the mesh by ORIGINAL ids -- a face: (Dm^-1 0, 1, 3, |vol|) and its three
// corner vertices; a vertex: its up to eight (face id << 2 | corner) adjacency records -- so that the receiving rank
// needs no table of the whole scene's topology (per-rank topology).
// Buffer = 16-byte header (record count) + capacity records.
#pragma once
#include "mpm_device.h"
#include "mpm_rebuild.h"   // time_to_travel

namespace mpm {

constexpr int DIST_REC_F4 = 9;
MPM_DEV float4 as_f4(const int4& v) { return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)); }
MPM_DEV int4 as_i4(const float4& v) { return make_int4(__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)); }
constexpr int ROLE_GHOST = 2, ROLE_OWNED = 1;

MPM_DEV int dist_cell_x(const DP& p, float x) {
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
    return (int)min(base_cell(x, p.dxinv), hi);
}

// mpm_dist_init: every particle is here and owned (replicated Finalize); keep what this rank owns,
// mark the ghosts, release the rest.  No communication: all ranks start from the same state.
__global__ __launch_bounds__(256) void k_dist_init_roles(DP p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Np) return;
    const PSet& S = p.set[p.ctl->cur];
    const Dist& d = p.dist;
    float4 q = S.q[0][i];
    const int bx = dist_cell_x(p, q.x);
    const float xi = dist_xi(q.x, p.dxinv);
    const float vol = fabsf(q.w);
    const bool mine = bx >= d.own_lo && bx < d.own_hi, face = i < p.Nf;
    q.w = mine ? vol : (dist_in_my_band(d, bx, xi, face) ? -vol : 0.f);
    S.q[0][i] = q;
    d.prev[S.pid[i]] = mine ? (unsigned char)dist_in_neighbour_bands(d, xi, face) : 0;
}

MPM_DEV void dist_emit(const DP& p, const PSet& S, unsigned slot, int gid, int role, float4 q0, bool emit, float4* buf,
                       unsigned cap) {
    const unsigned long long m = __ballot(emit);
    if (!m) return;
    const int lane = threadIdx.x & 63, lead = __builtin_ctzll(m);
    unsigned base = 0;
    // header: [0] records, [1] how many of them are faces (the receiver sizes its slot space from the two counts)
    const unsigned long long mf = __ballot(emit && slot < (unsigned)p.Nf);
    if (lane == lead) {
        base = atomicAdd(reinterpret_cast<unsigned*>(buf), (unsigned)__popcll(m));
        if (mf) atomicAdd(reinterpret_cast<unsigned*>(buf) + 1, (unsigned)__popcll(mf));
    }
    base = (unsigned)__shfl((int)base, lead);
    if (!emit) return;
    const unsigned at = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (at >= cap) {
        atomicOr(&p.ctl->error, ERR_CAPACITY);
        return;
    }
    float4* r = buf + 1 + (size_t)at * DIST_REC_F4;
    const bool face = slot < (unsigned)p.Nf;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    q0.w = fabsf(q0.w);
    r[0] = make_float4(__int_as_float(gid), __int_as_float(role), face ? S.f8[slot] : 0.f, 0.f);
    r[1] = q0;
    float4 q1 = S.q[1][slot];
    if (face) q1.w = S.c8[slot];   // (the authoritative C8 of a face particle, see PSet)
    r[2] = q1;
    r[3] = S.q[2][slot];
    r[4] = S.q[3][slot];
    r[5] = face ? S.fq[0][slot] : z;
    r[6] = face ? S.fq[1][slot] : z;
    const int cs = p.ctl->cur;
    if (face) {
        r[7] = S.fq[2][slot];
        r[8] = as_f4(p.fg[cs][slot]);
    } else {
        r[7] = as_f4(p.vg[cs][0][slot - p.Nf]);
        r[8] = as_f4(p.vg[cs][1][slot - p.Nf]);
    }
}

__global__ __launch_bounds__(256) void k_dist_classify(DP p, float4* send_l, float4* send_r, unsigned cap) {
    Ctl* c = p.ctl;
    const PSet& S = p.set[c->cur];
    const Dist& d = p.dist;
    const int nf = c->nfa, total = nf + c->nva;
    bool changed = false;
    float quiet = __int_as_float(0x7F800000);   // Ctl::mig_quiet: this thread's particles
    for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {   // (whole waves stay in the loop)
        const int idx = base + threadIdx.x;
        const bool live = idx < total;
        const unsigned slot = (unsigned)active_slot(p, live ? idx : total - 1, nf);
        float4 q0 = S.q[0][slot];
        const int gid = S.pid[slot];
        const int bx = dist_cell_x(p, q0.x);
        const float xi = dist_xi(q0.x, p.dxinv);
        const float vol = q0.w;
        const bool face = slot < (unsigned)p.Nf;
        if (live && vol != 0.f) {
            // how long until this particle has drifted mig_delta cells along x (ballistic: its velocity, gravity if that
            // acts along x); one that is further than mig_reach from both cuts has to get there first
            const float vx = fabsf(S.q[1][slot].x) * p.dxinv;
            const float ax = p.M.gravity_axis == 0 ? fabsf(p.M.gravity) * p.dxinv : 0.f;
            float far = __int_as_float(0x7F800000);
            if (d.has_left) far = fminf(far, fabsf(xi - (float)d.own_lo));
            if (d.has_right) far = fminf(far, fabsf(xi - (float)d.own_hi));
            quiet = fminf(quiet, time_to_travel(d.mig_delta + fmaxf(far - d.mig_reach, 0.f), vx, ax));
        }
        bool to_l = false, to_r = false;
        int role_l = ROLE_GHOST, role_r = ROLE_GHOST;
        float new_vol = vol;
        if (live && vol > 0.f) {
            // (ownership changes d.hyst cells beyond a cut, not at it: see Dist::hyst)
            const bool left = d.has_left && xi < (float)d.own_lo - d.hyst, right = d.has_right && xi >= (float)d.own_hi + d.hyst;
            if (left || right) {
                // crossed a cut: the neighbour takes over
                if ((left && bx < d.nbr_lo) || (right && bx >= d.nbr_hi))
                    atomicOr(&c->error, ERR_HALO);   // (beyond the neighbour's slab)
                to_l = left; to_r = !left;
                role_l = role_r = ROLE_OWNED;
                new_vol = dist_in_my_band(d, bx, xi, face) ? -vol : 0.f;
                d.prev[gid] = 0;
            } else {
                const unsigned char old = d.prev[gid];
                const int in = dist_in_neighbour_bands(d, xi, face, old);
                to_l = (in & 1) && !(old & 1);
                to_r = (in & 2) && !(old & 2);
                d.prev[gid] = (unsigned char)in;
            }
        } else if (live && vol < 0.f) {
            // a ghost stays while it is in this rank's band, d.hyst beyond it included (its owner applies the same test
            // to the same position); one that crossed INTO this rank's slab waits for its owner's record
            const bool mine = bx >= d.own_lo && bx < d.own_hi;
            if (!mine && !dist_in_my_band(d, bx, xi, face, d.hyst)) new_vol = 0.f;
        }
        dist_emit(p, S, slot, gid, role_l, q0, to_l, send_l, cap);
        dist_emit(p, S, slot, gid, role_r, q0, to_r, send_r, cap);
        if (live && new_vol != vol) {
            q0.w = new_vol;
            S.q[0][slot] = q0;
            changed |= new_vol == 0.f;
        }
    }
    if (__ballot(changed) && (threadIdx.x & 63) == 0) c->need_rebuild = 1;
    // minimum over the workgroup, then over 32 slots, 128 bytes apart (like k_rb_count's quiet time: complements, maxima)
    __shared__ unsigned s_quiet[4];
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) quiet = fminf(quiet, __shfl_xor(quiet, dd));
    if ((threadIdx.x & 63) == 0) s_quiet[threadIdx.x >> 6] = __float_as_uint(quiet);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = min(min(s_quiet[0], s_quiet[1]), min(s_quiet[2], s_quiet[3]));
        atomicMax(&d.mig_min[(blockIdx.x & 31u) * 32u], ~m);
    }
}

// the 32 slots of k_dist_classify -> Ctl::mig_quiet (seconds; infinity when the rank holds nothing that moves)
__global__ __launch_bounds__(64) void k_dist_mig_reduce(DP p) {
    unsigned q = threadIdx.x < 32 ? p.dist.mig_min[threadIdx.x * 32u] : 0u;
    if (threadIdx.x < 32) p.dist.mig_min[threadIdx.x * 32u] = 0u;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) q = max(q, (unsigned)__shfl_xor((int)q, dd));
    if (threadIdx.x == 0) p.ctl->mig_quiet = q ? __uint_as_float(~q) : __int_as_float(0x7F800000);
}

__global__ __launch_bounds__(256) void k_dist_apply(DP p, const float4* recv, unsigned cap) {
    Ctl* c = p.ctl;
    const PSet& S = p.set[c->cur];
    const unsigned n = min(*reinterpret_cast<const unsigned*>(recv), cap);
    for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const float4* r = recv + 1 + (size_t)k * DIST_REC_F4;
        const float4 h = r[0];
        const int gid = __float_as_int(h.x), role = __float_as_int(h.y);
        if (gid < 0 || gid >= p.NpG) {
            atomicOr(&c->error, ERR_HALO);
            continue;
        }
        const bool face = gid < p.NfG;
        int slot = p.imap[gid];
        if (slot < 0) {
            // not held yet: append behind the active particles of its type (the re-sort merges it)
            const int at = atomicAdd(face ? &c->add_f : &c->add_v, 1);
            slot = face ? c->nfa + at : p.Nf + c->nva + at;
            if (slot >= (face ? p.Nf : p.Np)) {
                atomicOr(&c->error, ERR_CAPACITY);
                continue;
            }
            S.pid[slot] = gid;
            p.imap[gid] = slot;
            const int cs = c->cur;
            if (face) {
                // its static record and its corners by original id come with it; the slots of the corners are
                // looked up by the re-sort that follows (-1 until then)
                S.fq[2][slot] = make_float4(r[7].x, r[7].y, r[7].z, fabsf(r[1].w));
                S.fq[3][slot] = make_float4(0.f, __int_as_float(-1), __int_as_float(-1), __int_as_float(-1));
                p.fg[cs][slot] = as_i4(r[8]);
            } else {
                p.vg[cs][0][slot - p.Nf] = as_i4(r[7]);
                p.vg[cs][1][slot - p.Nf] = as_i4(r[8]);
            }
        }
        if (face) {
            S.f8[slot] = h.z;
            S.c8[slot] = r[2].w;
        }
        float4 q0 = r[1];
        q0.w = role == ROLE_OWNED ? q0.w : -q0.w;
        S.q[0][slot] = q0;
        S.q[1][slot] = r[2];
        S.q[2][slot] = r[3];
        S.q[3][slot] = r[4];
        if (face) {
            S.fq[0][slot] = r[5];
            S.fq[1][slot] = r[6];
        }
        if (role == ROLE_OWNED) p.dist.prev[gid] = 0;
    }
    if (n && threadIdx.x == 0 && blockIdx.x == 0) c->need_rebuild = 1;
}

// role of the particle in each API slot: 0 not on this rank, 1 owned, 2 ghost
__global__ __launch_bounds__(256) void k_dist_roles(DP p, const int* pids_api, unsigned char* out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.NpG) return;
    const int j = p.imap[pids_api[s]];
    unsigned char r = 0;
    if (j >= 0) {
        const float v = p.set[p.ctl->cur].q[0][j].w;
        r = v > 0.f ? 1 : (v < 0.f ? 2 : 0);
    }
    out[s] = r;
}

// mpm_dist_init: per-slot topology by original id for the particles this rank keeps, from the tables of the whole
// scene that Finalize built (which are released afterwards -- except the adjacency of a mesh that has a vertex with more
// than eight adjacent faces: that vertex's record is a mark, and its force is summed over the scene's adjacency).
__global__ __launch_bounds__(256) void k_dist_build_topology(DP p) {
    const Ctl* c = p.ctl;
    const PSet& S = p.set[c->cur];
    const int cs = c->cur, nf = c->nfa, total = nf + c->nva;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int slot = active_slot(p, idx, nf);
        const int gid = S.pid[slot];
        if (slot < p.Nf) {
            p.fg[cs][slot] = make_int4(p.idx_orig[0][gid], p.idx_orig[1][gid], p.idx_orig[2][gid], __float_as_int(S.fq[3][slot].x));
        } else {
            const int vo = gid - p.NfG, e0 = p.adj_off[vo], e1 = p.adj_off[vo + 1];
            int rec[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) rec[q] = e0 + q < e1 ? p.adj_fc[e0 + q] : -1;
            if (e1 - e0 > 8) {   // the record holds eight: the mark sends this vertex to the scene's adjacency, which stays
                rec[0] = -2;     // resident on every rank for such a mesh (dist_resize)
#pragma unroll
                for (int q = 1; q < 8; ++q) rec[q] = -1;
            }
            p.vg[cs][0][slot - p.Nf] = make_int4(rec[0], rec[1], rec[2], rec[3]);
            p.vg[cs][1][slot - p.Nf] = make_int4(rec[4], rec[5], rec[6], rec[7]);
        }
    }
}

}  // namespace mpm
