// Rebuild = the engine's particle sort: a counting sort of the particle set by
// (type, block, cell) plus the block tables that the tile kernels use.  The
// cell-level order only has to be approximately right: the transfer kernels
// regroup each 512-particle chunk by its current base cell in LDS, a well
// sorted chunk just spans fewer cells.
// It replaces RebuildMapping's key + radix sort + compute_sorted_state
// (cuda_mpm_solver.cu:17-70, radix_sort.cuh, cuda_mpm_kernels.cuh:365-416).
//
// All four kernels are launched every substep and return immediately unless
// Ctl::need_rebuild is set (raised on the device by the P2G kernel), so the
// decision never costs a host round trip.
#pragma once
#include "mpm_device.h"

namespace mpm {

// R1: cell key of every particle, its arrival rank inside the cell and the
// per-block / per-cell histograms.  Consecutive slots mostly share a cell (the
// previous order was cell sorted), so each wave first merges its lanes by
// (cell, type) and issues one integer atomic per distinct cell and block
// instead of one per particle (device-scope atomics are resolved at the
// memory side and are expensive when thousands of them hit one address).
__global__ __launch_bounds__(256) void k_rb_count(DP p) {
    if (!p.ctl->need_rebuild) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < p.Np;
    const PSet& S = p.set[p.ctl->cur];
    const int ii = valid ? i : p.Np - 1;
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
    uint32_t bx = base_cell(S.x[0][ii], p.dxinv), by = base_cell(S.x[1][ii], p.dxinv),
             bz = base_cell(S.x[2][ii], p.dxinv);
    if (valid && (bx > hi || by > hi || bz > hi)) atomicOr(&p.ctl->error, ERR_DOMAIN);
    bx = min(bx, hi); by = min(by, hi); bz = min(bz, hi);
    const int t = ii >= p.Nf;
    const uint32_t key = cell_key(bx, by, bz);
    const int lane = threadIdx.x & 63;
    // a wave never straddles more than two types; handle them one after the other
    uint32_t rank = 0;
    for (int ty = 0; ty < 2; ++ty) {
        unsigned long long todo = __ballot(valid && t == ty);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const uint32_t lk = (uint32_t)__shfl((int)key, lead);
            const unsigned long long same = __ballot(key == lk) & todo;
            int base = 0;
            if (lane == lead) base = atomicAdd(&p.cellcnt[ty][lk], (int)__popcll(same));
            base = __shfl(base, lead);
            if (same & (1ull << lane)) rank = (uint32_t)base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            todo &= ~same;
        }
        // block histogram, merged per wave in the same way
        todo = __ballot(valid && t == ty);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const uint32_t lb = (uint32_t)__shfl((int)(key >> 6), lead);
            const unsigned long long same = __ballot((key >> 6) == lb) & todo;
            if (lane == lead) atomicAdd(&p.blkcnt[ty][lb], (int)__popcll(same));
            todo &= ~same;
        }
    }
    if (valid) {
        p.pkey[i] = key;
        p.prank[i] = rank;
    }
}

struct I3 {
    int a, b, c;
};
MPM_DEV I3 operator+(I3 l, I3 r) { return {l.a + r.a, l.b + r.b, l.c + r.c}; }

// exclusive scan of one I3 per thread across a 1024-thread workgroup
MPM_DEV I3 wg_scan_exclusive(I3 v, I3& total, I3 (*s_w)[1]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    I3 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        I3 t = {__shfl_up(inc.a, d), __shfl_up(inc.b, d), __shfl_up(inc.c, d)};
        if (lane >= d) inc = inc + t;
    }
    __syncthreads();
    if (lane == 63) s_w[w][0] = inc;
    __syncthreads();
    I3 pre = {0, 0, 0}, tot = {0, 0, 0};
    for (int k = 0; k < 16; ++k) {
        const I3 t = s_w[k][0];
        if (k < w) pre = pre + t;
        tot = tot + t;
    }
    total = tot;
    return {pre.a + inc.a - v.a, pre.b + inc.b - v.b, pre.c + inc.c - v.c};
}

// R2: one workgroup turns the histograms into
//   - the home-block list with its particle ranges and the scatter offsets,
//   - the active-block list (27-neighbourhood of the home blocks),
//   - both neighbour tables.
__global__ __launch_bounds__(1024) void k_rb_tables(DP p) {
    Ctl* c = p.ctl;
    if (!c->need_rebuild) return;
    __shared__ I3 s_w[16][1];
    __shared__ I3 s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = {0, 0, 0};
    __syncthreads();

    // A: prefix sums over all blocks
    for (unsigned base = 0; base < p.nblocks; base += 1024) {
        const unsigned b = base + tid;
        I3 v = {0, 0, 0};
        if (b < p.nblocks) {
            v.b = p.blkcnt[0][b];
            v.c = p.blkcnt[1][b];
            v.a = (v.b + v.c) > 0;
        }
        I3 tot;
        const I3 ex = wg_scan_exclusive(v, tot, s_w);
        const I3 carry = s_carry;
        if (b < p.nblocks) {
            const int f0 = carry.b + ex.b, v0 = carry.c + ex.c;
            p.blkstart[0][b] = f0;
            p.blkstart[1][b] = v0;
            p.act_flag[b] = 0;
            int slot = -1;
            if (v.a) {
                slot = carry.a + ex.a;
                if ((unsigned)slot < p.capH) {
                    p.home_block[slot] = b;
                    p.home_range[slot] = make_int4(f0, f0 + v.b, p.Nf + v0, p.Nf + v0 + v.c);
                } else {
                    slot = -1;
                }
            }
            p.lut_home[b] = slot;
        }
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    unsigned n_home = (unsigned)s_carry.a;
    if (n_home > p.capH) {
        if (tid == 0) atomicOr(&c->error, ERR_CAPACITY);
        n_home = p.capH;
    }

    // B: per home block, exclusive prefix of its 64 cell counters (in place)
    for (unsigned w = tid; w < n_home * 2; w += 1024) {
        const unsigned h = w >> 1, t = w & 1;
        int4* cc = reinterpret_cast<int4*>(p.cellcnt[t] + (size_t)p.home_block[h] * 64);
        int run = 0;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            int4 q = cc[k];
            const int s0 = run, s1 = s0 + q.x, s2 = s1 + q.y, s3 = s2 + q.z;
            run = s3 + q.w;
            cc[k] = make_int4(s0, s1, s2, s3);
        }
    }
    // C: flag the 27-neighbourhood of every home block
    for (unsigned w = tid; w < n_home * 27; w += 1024) {
        const int nbid = neighbor_block(p.home_block[w / 27], (int)(w % 27), p.nb);
        if (nbid >= 0) p.act_flag[nbid] = 1;
    }
    __syncthreads();
    if (tid == 0) s_carry = {0, 0, 0};
    __syncthreads();

    // D: compact the flags (ascending block id)
    for (unsigned base = 0; base < p.nblocks; base += 1024) {
        const unsigned b = base + tid;
        I3 v = {0, 0, 0};
        if (b < p.nblocks) v.a = p.act_flag[b];
        I3 tot;
        const I3 ex = wg_scan_exclusive(v, tot, s_w);
        const I3 carry = s_carry;
        if (b < p.nblocks) {
            int slot = -1;
            if (v.a) {
                slot = carry.a + ex.a;
                if ((unsigned)slot < p.capA) p.act_block[slot] = b; else slot = -1;
            }
            p.lut_act[b] = slot;
        }
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    unsigned n_active = (unsigned)s_carry.a;
    if (n_active > p.capA) {
        if (tid == 0) atomicOr(&c->error, ERR_CAPACITY);
        n_active = p.capA;
    }

    // F: work-queue order, heaviest blocks first (longest-processing-time-first
    // keeps the last workgroups short).  Counting sort on the chunk count.
    {
        __shared__ int s_bucket[64];
        if (tid < 64) s_bucket[tid] = 0;
        __syncthreads();
        auto bucket_of = [&](unsigned h) {
            const int4 rg = p.home_range[h];
            const int chunks = ((rg.y - rg.x) + (rg.w - rg.z) + 255) >> 8;
            return 63 - min(chunks, 63);  // descending
        };
        for (unsigned h = tid; h < n_home; h += 1024) atomicAdd(&s_bucket[bucket_of(h)], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < 64; ++k) {
                const int c = s_bucket[k];
                s_bucket[k] = run;
                run += c;
            }
        }
        __syncthreads();
        for (unsigned h = tid; h < n_home; h += 1024) p.home_order[atomicAdd(&s_bucket[bucket_of(h)], 1)] = h;
        __syncthreads();
    }

    // E: neighbour tables
    for (unsigned w = tid; w < n_home * 27; w += 1024) {
        const int nbid = neighbor_block(p.home_block[w / 27], (int)(w % 27), p.nb);
        p.home_nbr_act[w] = nbid >= 0 ? p.lut_act[nbid] : -1;
    }
    for (unsigned w = tid; w < n_active * 27; w += 1024) {
        const int nbid = neighbor_block(p.act_block[w / 27], (int)(w % 27), p.nb);
        p.act_nbr_home[w] = nbid >= 0 ? p.lut_home[nbid] : -1;
    }
    if (tid == 0) {
        c->n_home = n_home;
        c->n_active = n_active;
    }
}

// R3: move every particle to its sorted slot in the other PSet.
__global__ __launch_bounds__(256) void k_rb_scatter(DP p) {
    if (!p.ctl->need_rebuild) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Np) return;
    const int cur = p.ctl->cur;
    const PSet& S = p.set[cur];
    const PSet& D = p.set[cur ^ 1];
    const uint32_t key = p.pkey[i];
    const int t = i >= p.Nf;
    const int dst = (t ? p.Nf : 0) + p.blkstart[t][key >> 6] + p.cellcnt[t][key] + (int)p.prank[i];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        D.x[d][dst] = S.x[d][i];
        D.v[d][dst] = S.v[d][i];
    }
    D.vol[dst] = S.vol[i];
#pragma unroll
    for (int d = 0; d < 9; ++d) D.C[d][dst] = S.C[d][i];
    const int pid = S.pid[i];
    D.pid[dst] = pid;
    p.imap[pid] = dst;
    if (!t) {
#pragma unroll
        for (int d = 0; d < 9; ++d) D.F[d][dst] = S.F[d][i];
#pragma unroll
        for (int d = 0; d < 4; ++d) D.Dm[d][dst] = S.Dm[d][i];
    }
}

// R4: refresh face -> vertex slots, re-zero the histograms, flip the sets.
__global__ __launch_bounds__(256) void k_rb_finish(DP p) {
    Ctl* c = p.ctl;
    if (!c->need_rebuild) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const PSet& D = p.set[c->cur ^ 1];
    if (i < p.Nf) {
        const int pid = D.pid[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) p.fv[k][i] = p.imap[p.idx_orig[k][pid]];
    }
    // every cell row of a home block holds prefix values now: clear whole rows
    const unsigned n_home = c->n_home;
    for (unsigned w = (unsigned)i; w < n_home * 32u; w += gridDim.x * 256u) {
        const unsigned h = w >> 5, t = (w >> 4) & 1u, k = w & 15u;
        const uint32_t b = p.home_block[h];
        reinterpret_cast<int4*>(p.cellcnt[t] + (size_t)b * 64)[k] = make_int4(0, 0, 0, 0);
        if (k == 0) p.blkcnt[t][b] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&c->ticket, 1u) == gridDim.x - 1) {
            c->ticket = 0;
            c->cur ^= 1;
            c->need_rebuild = 0;
            c->rebuilds += 1;
            __threadfence();
        }
    }
}

}  // namespace mpm
