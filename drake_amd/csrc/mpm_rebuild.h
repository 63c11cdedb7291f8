// Rebuild = the engine's particle sort: a counting sort of the particle set by
// (type, block, cell) plus the block tables that the tile kernels use.  The
// cell-level order only has to be approximately right: P2G regroups each
// 64-particle wave group by its current base cell with ballots, a well sorted
// group just spans fewer cells.
// It replaces RebuildMapping's key + radix sort + compute_sorted_state
// (cuda_mpm_solver.cu:17-70, radix_sort.cuh, cuda_mpm_kernels.cuh:365-416).
//
// All four kernels are launched every substep and return immediately unless
// Ctl::need_rebuild is set (raised on the device by the G2P kernel), so the
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
// time until a point at distance d (cells) from a bound is there, moving at v towards it with acceleration a
MPM_DEV float time_to_travel(float d, float v, float a) {
    if (!(d > 0.f)) return 0.f;
    const float disc = v * v + 2.f * a * d;
    if (!(disc >= 0.f)) return __int_as_float(0x7F800000);   // turns around before it gets there
    // (the hardware's square root and reciprocal, one instruction and one ulp each: this is an ESTIMATE that the host
    // halves before it uses it, computed six times per particle -- with IEEE sqrtf and division it was a third of the
    // vector instructions of k_rb_count)
    const float den = v + __builtin_amdgcn_sqrtf(disc);
    return den > 0.f ? 2.f * d * __builtin_amdgcn_rcpf(den) : __int_as_float(0x7F800000);
}

__global__ __launch_bounds__(256) void k_rb_count(DP p) {
    if (!p.ctl->need_rebuild) return;
    const PSet& S = p.set[p.ctl->cur];
    float quiet = __int_as_float(0x7F800000);   // Ctl::quiet_time: this thread's particles
    // the particles to sort: the active ones and what a migration appended behind them
    const int nf_in = p.ctl->nfa + p.ctl->add_f, total = nf_in + p.ctl->nva + p.ctl->add_v;
    // grid-stride over 256-particle chunks: a small fixed grid keeps the idle launches cheap
#if MPM_DIAG
    const unsigned long long tc0 = __builtin_readcyclecounter();
    int stamp_it = 0;
    auto cstamp = [&](int k) {
        if (blockIdx.x == 1000 && threadIdx.x == 0 && (diag_flags(p) & 1024) && stamp_it < 2) p.dbgbuf[stamp_it * 4 + k] = __builtin_readcyclecounter() - tc0;
    };
#else
    auto cstamp = [&](int) {};
#endif
    for (int base = blockIdx.x * 256; base < total; base += gridDim.x * 256) {
    const int idx = base + threadIdx.x;
    const bool listed = idx < total;
    const int i = active_slot(p, listed ? idx : total - 1, nf_in);
    const int ii = i;
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
    // A face particle sits at the centroid of its corners and moves with their mean velocity: CalcFemStateAndForce
    // puts it there before anything else looks at it, so that is where it is binned.  Its own position / velocity
    // records may be a substep old (GridToParticle's lean mode does not write them).  In a partitioned domain the
    // record is current (no lean mode there) and carries the particle's role in its w.
    const bool face_from_corners = listed && ii < p.Nf && !p.dist.on;
    float4 xq;
    float4 rec = make_float4(.5f, .5f, .5f, 0.f);   // a face's own position record (only checked against the grid)
    float vx = 0.f, vy = 0.f, vz = 0.f;
    if (face_from_corners) {
        rec = S.q[0][ii];
        const float4 f3 = S.fq[3][ii];
        const int s0 = __float_as_int(f3.y), s1 = __float_as_int(f3.z), s2 = __float_as_int(f3.w);
        const float4 xa = S.q[0][s0], xb = S.q[0][s1], xc = S.q[0][s2];
        const float4 a = S.q[1][s0], b = S.q[1][s1], c = S.q[1][s2];
        // (the centroid as k_fem forms it: a division, or -- fast math, DP::fem_fast -- a third as a product: an IEEE
        // division is ten vector instructions, and there are six)
        const bool fast = p.fem_fast != 0;
        auto mean3 = [fast](float u, float v, float w) { return fast ? (u + v + w) * (1.f / 3.f) : (u + v + w) / 3.f; };
        xq = make_float4(mean3(xa.x, xb.x, xc.x), mean3(xa.y, xb.y, xc.y), mean3(xa.z, xb.z, xc.z), 1.f);
        vx = mean3(a.x, b.x, c.x); vy = mean3(a.y, b.y, c.y); vz = mean3(a.z, b.z, c.z);
    } else {
        xq = S.q[0][ii];
        if (!p.dist.on) {   // (a partitioned domain bins by position and checks with every substep)
            const float4 vq = S.q[1][ii];
            vx = vq.x; vy = vq.y; vz = vq.z;
        }
    }
    const bool valid = listed && xq.w != 0.f;   // volume 0: released by the migration, dropped here
#if MPM_DIAG
    if (diag_flags(p) & 1024) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cstamp(0); }
#endif
    // Anticipatory binning: a particle may sit up to FREE_ZONE cells outside its home block, on either side.
    // Binned by where it will be a few substeps from now (at most 1.75 cells ahead, which leaves it
    // >= 0.25 cells inside the upstream free zone today), it has up to 2 + 4 + 2 cells to travel before
    // the next re-sort instead of as little as 2: in free fall the re-sorts come ~3x less often.
    // Slow particles are binned where they are.  (The cell part of the key follows the shifted position
    // too: the order inside a block only has to be approximately by cell.)
    float px = xq.x, py = xq.y, pz = xq.z;
    if (p.anticip > 0.f) {
        const float A = 1.75f;
        px += fminf(fmaxf(vx * p.anticip, -A), A) * p.dx;
        py += fminf(fmaxf(vy * p.anticip, -A), A) * p.dx;
        pz += fminf(fmaxf(vz * p.anticip, -A), A) * p.dx;
    }
    uint32_t bx = base_cell(px, p.dxinv), by = base_cell(py, p.dxinv), bz = base_cell(pz, p.dxinv);
    // (a negative coordinate saturates to cell 0 in the conversion, so test the float)
    const float lim = (float)(hi + 1u);
    auto inside = [&](float x) { const float t = x * p.dxinv - .5f; return t >= 0.f && t < lim; };
    // (a face's own record too: stale inside a batch of substeps, but then it was inside the grid when it was written;
    // a caller's upload that puts it outside is reported like any other particle's)
    if (valid && !(inside(xq.x) && inside(xq.y) && inside(xq.z) &&
                   (!face_from_corners || (inside(rec.x) && inside(rec.y) && inside(rec.z)))) &&
        !(p.ctl->error & ERR_DOMAIN))
        atomicOr(&p.ctl->error, ERR_DOMAIN);
    bx = min(bx, hi); by = min(by, hi); bz = min(bz, hi);
    if (valid && !p.dist.on) {
        // the test k_g2p applies to the advected position (guard band included), in the tile of the block the
        // particle is binned to
        const float guard = .125f, top = (float)(TILE_W - 2) - guard;
        const float pos[3] = {xq.x, xq.y, xq.z}, vel[3] = {vx, vy, vz};
        const uint32_t bc[3] = {bx, by, bz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos[k] * p.dxinv - .5f - (float)((int)(bc[k] & ~3u) - FREE_ZONE);
            const float v = vel[k] * p.dxinv, a = k == p.M.gravity_axis ? p.M.gravity * p.dxinv : 0.f;
            quiet = fminf(quiet, fminf(time_to_travel(top - t, v, a), time_to_travel(t - guard, -v, -a)));
        }
    }
    if (p.dist.on && valid && xq.w > 0.f) {
        // an owned particle may sit up to zone - 3 cells beyond a cut: its stencil (2 more cells) and the
        // motion until the next re-sort (the tile's free zone, < 1 cell past this test on average) stay
        // inside the blocks whose sums the neighbour receives
        const int slack = p.dist.zone_cells - 3;
        if (((int)bx < p.dist.own_lo - slack || (int)bx >= p.dist.own_hi + slack) && !(p.ctl->error & ERR_HALO))
            atomicOr(&p.ctl->error, ERR_HALO);
    }
    const int t = ii >= p.Nf;
    const uint32_t key = cell_key(bx, by, bz);
    const int lane = threadIdx.x & 63;
    // The lanes of a wave are merged by (type, cell) and by (type, block) with ballots, and then ALL group leaders
    // issue their atomic in ONE wave instruction per histogram (different addresses in different lanes).  Round 1-3 issued
    // one single-lane atomic instruction per distinct cell and block from inside the ballot loops: a CU retires a
    // memory-side atomic instruction every ~50 ns whatever its active lanes (MI355X_MICROARCH.md, global float
    // atomics: a wave instruction with 64 scattered lanes takes 17 of those, 13 ns per request), so ~7 instructions per
    // wave were 0.35 us per wave and the kernel's bound; now 2.
    int ret_cell = 0, ret_blk = 1, my_lead = lane, blk_leader = lane;
    unsigned long long my_same = 0ull, blk_same = 0ull;
    {
        // Merged by RUNS of equal keys in lane order (one shuffle, one ballot), not by distinct keys (a ballot loop with
        // one trip per distinct cell and block of the wave: 5.6k cycles per pass with eight waves per SIMD taking turns,
        // scratch/count_diag.py).  The particles come in the order of the last sort, so a cell is one run unless its
        // particles have moved apart; a cell in two runs gets two atomics and two rank ranges -- any unique rank inside
        // the cell will do (the canonical order of deterministic mode is established afterwards, k_rb_canon).
        auto runs = [&](uint32_t k, int& lead, unsigned long long& same) {
            const uint32_t prev = (uint32_t)__shfl_up((int)k, 1);
            const unsigned long long heads = __ballot(lane == 0 || k != prev);
            const unsigned long long upto = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
            lead = 63 - (int)__builtin_clzll(upto);                                    // (lane 0 is a head: never zero)
            const unsigned long long above = lane < 63 ? heads & ~((2ull << lane) - 1ull) : 0ull;
            const int next = above ? (int)__builtin_ctzll(above) : 64;
            same = (next == 64 ? ~0ull : ((1ull << next) - 1ull)) & ~((1ull << lead) - 1ull);
        };
        const uint32_t ckey = valid ? (key | ((uint32_t)t << 31)) : 0xFFFFFFFFu;   // (type in the top bit: one pass for both)
        runs(ckey, my_lead, my_same);
        const uint32_t bkey = valid ? ((key >> 6) | ((uint32_t)t << 31)) : 0xFFFFFFFFu;
        runs(bkey, blk_leader, blk_same);
    }
    cstamp(1);
    const bool cell_lead = valid && lane == my_lead, blk_lead = valid && lane == blk_leader;
    if (cell_lead) ret_cell = atomicAdd(&p.cellcnt[t][key], (int)__popcll(my_same));
    if (blk_lead) ret_blk = atomicAdd(&p.blkcnt[t][key >> 6], (int)__popcll(blk_same));
    // same-address device atomics serialise at ~30 ns each: only the first arrival of a
    // (type, block) pair touches the non-empty bitmap
    if (blk_lead && ret_blk == 0) atomicOr(&p.home_bits[key >> 11], 1u << ((key >> 6) & 31u));   // (one instruction too)
    const uint32_t rank = (uint32_t)__shfl(ret_cell, my_lead) + (uint32_t)__popcll(my_same & ((1ull << lane) - 1ull));
#if MPM_DIAG
    if (diag_flags(p) & 1024) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cstamp(2); }
#endif
    if (listed) {
        p.pkey[i] = valid ? key : 0xFFFFFFFFu;
        p.prank[i] = rank;
    }
#if MPM_DIAG
    if (diag_flags(p) & 1024) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cstamp(3); ++stamp_it; }
#endif
    }
    if (!p.dist.on) {
        // minimum over the workgroup, then over 32 slots, 128 bytes apart (same-address device atomics serialise at the
        // memory side: one slot for everybody cost 80 us per re-sort); k_rb_finish takes the minimum of the slots.
        // The slots hold the COMPLEMENT of the float's bits and take maxima, so that zero-filled memory means "none".
        __shared__ unsigned s_quiet[4];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) quiet = fminf(quiet, __shfl_xor(quiet, d));
        if ((threadIdx.x & 63) == 0) s_quiet[threadIdx.x >> 6] = __float_as_uint(quiet);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned m = min(min(s_quiet[0], s_quiet[1]), min(s_quiet[2], s_quiet[3]));
            atomicMax(&p.tickets[(blockIdx.x & 31u) * 32u + 1u], ~m);
        }
    }
}

// exclusive scan of one int per thread across a 1024-thread workgroup
MPM_DEV int wg_scan_exclusive(int v, int& total, int* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int t = s_w[k];
        pre += k < w ? t : 0;
        tot += t;
    }
    total = tot;
    return pre + inc - v;
}

constexpr int MAX_BITMAP_WORDS = 8192;  // 2^18 blocks = 256^3 cells

// R2: one workgroup turns the histograms into
//   - the home-block list (ascending block id) with its particle ranges and scatter offsets,
//   - the active-block list (27-neighbourhood of the home blocks),
//   - the work items of the tile kernels in heaviest-first order.
// All work is proportional to the number of home blocks: non-empty blocks come from the bitmap
// k_rb_count filled, the neighbourhood union is built in an LDS bitmap.
__global__ __launch_bounds__(1024) void k_rb_tables(DP p) {
    Ctl* c = p.ctl;
    if (!c->need_rebuild) return;
#if MPM_DIAG
    const unsigned long long tk0 = __builtin_readcyclecounter();
    auto stamp = [&](int k) { if (threadIdx.x == 0 && (diag_flags(p) & 512)) p.dbgbuf[k] = __builtin_readcyclecounter() - tk0; };
#else
    auto stamp = [&](int) {};
#endif
    if (blockIdx.x > 0) {
        // B (all other workgroups, concurrently with workgroup 0): per non-empty block and type,
        // exclusive prefix of its 64 cell counters, in place.  16 lanes per row (one int4 each).
        const unsigned nwords = p.nblocks >> 5;
        const int sub = threadIdx.x & 15, rowl = threadIdx.x >> 4;  // 64 row slots per workgroup
        for (unsigned w = blockIdx.x - 1; w < nwords; w += gridDim.x - 1) {
            const unsigned bits = p.home_bits[w];
            if (!bits) continue;
            // row slot -> (k-th set bit of the word, type)
            const int nrows = 2 * __popc(bits);
            if (rowl >= nrows) continue;
            unsigned rest = bits;
            for (int k = 0; k < (rowl >> 1); ++k) rest &= rest - 1;
            const unsigned b = (w << 5) + (unsigned)__builtin_ctz(rest);
            int4* cc = reinterpret_cast<int4*>(p.cellcnt[rowl & 1] + (size_t)b * 64) + sub;
            const int4 q = *cc;
            const int tot = q.x + q.y + q.z + q.w;
            int inc = tot;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const int t = __shfl_up(inc, d, 16);
                if (sub >= d) inc += t;
            }
            const int s0 = inc - tot;
            *cc = make_int4(s0, s0 + q.x, s0 + q.x + q.y, s0 + q.x + q.y + q.z);
        }
        if (blockIdx.x == 1) stamp(15);
        return;
    }
    __shared__ int s_w[16];
    __shared__ unsigned s_bits[MAX_BITMAP_WORDS];
    __shared__ int s_bucket[64];
    const int tid = threadIdx.x;
    const unsigned words = p.nblocks >> 5;
    const unsigned wpt = (words + 1023u) / 1024u;          // bitmap words per thread
    const unsigned w0 = min(tid * wpt, words), w1 = min(w0 + wpt, words);

    // forget the previous tables
    for (unsigned k = tid; k < c->n_home; k += 1024) p.lut_home[p.home_block[k]] = -1;
    for (unsigned k = tid; k < c->n_active; k += 1024) p.lut_act[p.act_block[k]] = -1;
    __syncthreads();
    stamp(0);

    // A: home list = set bits of the non-empty bitmap, in ascending order
    int mine = 0;
    for (unsigned w = w0; w < w1; ++w) mine += __popc(p.home_bits[w]);
    int total_home = 0;
    int slot = wg_scan_exclusive(mine, total_home, s_w);
    for (unsigned w = w0; w < w1; ++w) {
        unsigned bits = p.home_bits[w];
        while (bits) {
            const unsigned b = (w << 5) + (unsigned)__builtin_ctz(bits);
            bits &= bits - 1;
            if ((unsigned)slot < p.capH) {
                p.home_block[slot] = b;
                p.lut_home[b] = slot;
            }
            ++slot;
        }
    }
    unsigned n_home = (unsigned)total_home;
    if (n_home > p.capH) {
        if (tid == 0) atomicOr(&c->error, ERR_CAPACITY);
        n_home = p.capH;
    }
    for (unsigned w = tid; w < words; w += 1024) s_bits[w] = 0;
    __syncthreads();
    stamp(1);

    // A2: particle ranges = prefix sums of the block counts over the home list
    const unsigned hpt = (n_home + 1023u) / 1024u;
    const unsigned h0 = min(tid * hpt, n_home), h1 = min(h0 + hpt, n_home);
    int held = 0;   // particles this engine holds after the re-sort
    {
        int cf = 0, cv = 0;
        for (unsigned h = h0; h < h1; ++h) {
            const uint32_t b = p.home_block[h];
            cf += p.blkcnt[0][b];
            cv += p.blkcnt[1][b];
        }
        int tf = 0, tv = 0;
        int rf = wg_scan_exclusive(cf, tf, s_w);
        int rv = wg_scan_exclusive(cv, tv, s_w);
        held = tf + tv;
        if (tid == 0) {
            c->nfa_new = tf;
            c->nva_new = tv;
        }
        for (unsigned h = h0; h < h1; ++h) {
            const uint32_t b = p.home_block[h];
            const int nf = p.blkcnt[0][b], nv = p.blkcnt[1][b];
            p.blkstart[0][b] = rf;
            p.blkstart[1][b] = rv;
            p.home_range[h] = make_int4(rf, rf + nf, p.Nf + rv, p.Nf + rv + nv);
            rf += nf;
            rv += nv;
        }
    }

    stamp(2);
    // C: union of the 27-neighbourhoods in an LDS bitmap
    // (a thread per home block, its 27 neighbours in a row: ONE global load, ONE Morton decode and nine bit spreads per
    // block.  A thread per (block, neighbour) pair put a dependent load in front of every LDS atomic and decoded and
    // encoded a Morton id per pair -- ~80 vector instructions each: 21k of this workgroup's 58k cycles at 1M particles)
    for (unsigned h = tid; h < n_home; h += 1024) {
        int bx, by, bz;
        block_coords(p.home_block[h], bx, by, bz);
        uint32_t sx[3], sy[3], sz[3];
        bool vx[3], vy[3], vz[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int x = bx + d - 1, y = by + d - 1, z = bz + d - 1;
            vx[d] = x >= 0 && x < p.nb; vy[d] = y >= 0 && y < p.nb; vz[d] = z >= 0 && z < p.nb;
            sx[d] = spread3((uint32_t)x) * 4u; sy[d] = spread3((uint32_t)y) * 2u; sz[d] = spread3((uint32_t)z);
        }
#pragma unroll
        for (int o = 0; o < 27; ++o) {
            const int a = o / 9, b = (o / 3) % 3, cz = o % 3;
            const uint32_t nbid = sx[a] + sy[b] + sz[cz];   // (= neighbor_block(home_block[h], o, nb))
            if (vx[a] && vy[b] && vz[cz]) atomicOr(&s_bits[nbid >> 5], 1u << (nbid & 31));
        }
    }
    __syncthreads();
    stamp(3);

    // D: active list = set bits, ascending
    mine = 0;
    for (unsigned w = w0; w < w1; ++w) mine += __popc(s_bits[w]);
    int total_act = 0;
    slot = wg_scan_exclusive(mine, total_act, s_w);
    for (unsigned w = w0; w < w1; ++w) {
        unsigned bits = s_bits[w];
        while (bits) {
            const unsigned b = (w << 5) + (unsigned)__builtin_ctz(bits);
            bits &= bits - 1;
            if ((unsigned)slot < p.capA) {
                p.act_block[slot] = b;
                p.lut_act[b] = slot;
            }
            ++slot;
        }
    }
    unsigned n_active = (unsigned)total_act;
    if (n_active > p.capA) {
        if (tid == 0) atomicOr(&c->error, ERR_CAPACITY);
        n_active = p.capA;
    }
    if (tid < 64) s_bucket[tid] = 0;
    __syncthreads();
    stamp(4);

    // F: work items.  A home block with more than item_groups wave groups is split evenly into
    // several items (so a dense pile still spreads over the CUs); the static round-robin order is
    // heaviest first (longest-processing-time-first keeps the last workgroups short).
    // Measured (scratch/item_sweep.py, 1M and 500k particles): the tile kernels take the same time for
    // items of 16 to 48 groups, in any dealing order (round robin, snake, a queue counter), and ~15%
    // longer for 8 to 12 (more slabs): items stay as large as the split for dense piles allows.
    unsigned n_items;
    {
        // (a small share -- a partitioned rank -- is cut into smaller items: scratch/share_scaling.py, an eighth of the 1M
        // workload: k_p2g 24.9 -> 15.8 us, k_g2p 8.6 -> 5.6 us with items of 16 groups; the whole workload is fastest at 48)
        const int ig = ((held + 63) >> 6) < p.item_small_below ? min(p.item_groups, p.item_groups_small) : p.item_groups;
        auto items_of = [&](int ng) { return (ng + ig - 1) / ig; };
        int mine_items = 0;
        for (unsigned h = h0; h < h1; ++h) {
            const int4 rg = p.home_range[h];
            const int ng = ((rg.y - rg.x) + (rg.w - rg.z) + 63) >> 6;
            mine_items += items_of(ng);
        }
        int total_items = 0;
        int it = wg_scan_exclusive(mine_items, total_items, s_w);
        for (unsigned h = h0; h < h1; ++h) {
            const int4 rg = p.home_range[h];
            const int ng = ((rg.y - rg.x) + (rg.w - rg.z) + 63) >> 6;
            const int ni = items_of(ng);
            p.home_items[h] = make_int2(it, ni);
            for (int k = 0; k < ni; ++k)
                if ((unsigned)(it + k) < p.capI)
                    p.item_desc[it + k] = make_int4((int)h, (int)((long long)k * ng / ni), (int)((long long)(k + 1) * ng / ni), 0);
            it += ni;
        }
        n_items = (unsigned)total_items;
        if (tid == 0) c->n_items_wanted = n_items;
        // (capS <= capI: the slab pool is sized by the blocks actually in use and grown by the host when it fills
        // up.  Running out is reported with the count that was wanted: substeps of mpm_run_substeps skip themselves
        // until the host has grown the pool and repeated this re-sort; phase-by-phase substeps cannot be repeated,
        // for them it ends as a capacity error)
        if (n_items > p.capS) {
            if (tid == 0) atomicOr(&c->error, ERR_SLABS);
            n_items = p.capS;
        }
        __syncthreads();
        stamp(5);
        auto bucket_of = [&](unsigned w) {
            const int4 d = p.item_desc[w];
            return 63 - min(d.z - d.y, 63);  // descending group count
        };
        for (unsigned w = tid; w < n_items; w += 1024) atomicAdd(&s_bucket[bucket_of(w)], 1);
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < 64; ++k) {
                const int cnt = s_bucket[k];
                s_bucket[k] = run;
                run += cnt;
            }
        }
        __syncthreads();
        for (unsigned w = tid; w < n_items; w += 1024) p.item_order[atomicAdd(&s_bucket[bucket_of(w)], 1)] = w;
        __syncthreads();
        stamp(6);
        // the same, flattened in processing order (see DP::item_flat)
        for (unsigned q = tid; q < n_items; q += 1024) {
            const unsigned w = p.item_order[q];
            p.item_pos[w] = q;
            const int4 d = p.item_desc[w];
            const unsigned h = (unsigned)d.x;
            const int4 rg = p.home_range[h];
            p.item_flat[2 * q] = make_int4((int)w, (int)h, (int)p.home_block[h], d.z - d.y);
            p.item_flat[2 * q + 1] = make_int4(rg.x, rg.y, rg.z, (int)(group_pool_offset(p, rg, h) + (size_t)d.y));
        }
    }

    stamp(7);
    if (tid == 0) {
        c->n_home = n_home;
        c->n_active = n_active;
        c->n_items = n_items;
    }
}

// R3: move every particle to its sorted slot in the other PSet.
__global__ __launch_bounds__(256) void k_rb_scatter(DP p) {
    if (!p.ctl->need_rebuild) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    {
        // neighbour tables (spread over the whole grid: one workgroup would be latency bound)
        const unsigned n_home = p.ctl->n_home, n_active = p.ctl->n_active;
        const unsigned gs = gridDim.x * 256u;
        for (unsigned w = (unsigned)i; w < n_home * 27; w += gs) {
            const int nbid = neighbor_block(p.home_block[w / 27], (int)(w % 27), p.nb);
            p.home_nbr_act[w] = nbid >= 0 ? p.lut_act[nbid] : -1;
        }
        for (unsigned w = (unsigned)i; w < n_active * 27; w += gs) {
            const int nbid = neighbor_block(p.act_block[w / 27], (int)(w % 27), p.nb);
            const int hn = nbid >= 0 ? p.lut_home[nbid] : -1;
            p.act_nbr_home[w] = hn;
            int packed = -1;
            if (hn >= 0) {
                const int2 it = p.home_items[hn];
                // (first item | count << 24) must stay a non-negative int: 127 items per block, 2^24 items
                if (it.y > 127 || it.x >= (1 << 24)) atomicOr(&p.ctl->error, ERR_CAPACITY);
                packed = it.x | (min(it.y, 127) << 24);
            }
            p.act_nbr_items[w] = packed;
        }
        // P2G wave groups: per home block, consecutive 64-particle windows of the merged sequence
        // (cell 0 faces, cell 0 vertices, cell 1 faces, ...), so a group holds faces and vertices of
        // the same ~3 cells.  Window g starts at merged position 64 g; every boundary is found
        // independently by a binary search over the block's 64 cell prefixes.  The groups stay valid
        // until the next rebuild (they are slot ranges, not positions).
        {
            const unsigned lane = threadIdx.x & 63u;
            const unsigned wave = (unsigned)i >> 6, nwaves = gs >> 6;
            for (unsigned h = wave; h < n_home; h += nwaves) {
                const uint32_t b = p.home_block[h];
                const int4 rg = p.home_range[h];
                const int* cf = p.cellcnt[0] + (size_t)b * 64;   // exclusive prefixes at this point
                const int* cv = p.cellcnt[1] + (size_t)b * 64;
                const int nf = rg.y - rg.x, nv = rg.w - rg.z, total = nf + nv;
                const int ng = (total + 63) >> 6;
                int4* out = p.home_groups + group_pool_offset(p, rg, h);
                // split of merged position m into (faces before it, vertices before it).  The 64 cell prefixes of the
                // block sit in the lanes of the wave (lane = cell): the binary search walks registers (six shuffles)
                // instead of memory (six dependent round trips per boundary -- the waves that build groups were this
                // kernel's critical path).  Every lane of the wave calls it.
                const int pf = cf[lane], pv = cv[lane], ps = pf + pv;
                auto split = [&](int m, int& f, int& v) {
                    int lo = 0, hi = 63;                       // last cell whose start is <= m
#pragma unroll
                    for (int it = 0; it < 6; ++it) {
                        const int mid = (lo + hi + 1) >> 1;
                        const int s_mid = __shfl(ps, mid);
                        if (lo < hi) { if (s_mid <= m) lo = mid; else hi = mid - 1; }
                    }
                    const int fs = __shfl(pf, lo), vs = __shfl(pv, lo);
                    const int fnext = __shfl(pf, min(lo + 1, 63));
                    const int fe = lo < 63 ? fnext : nf;
                    const int r = m - (fs + vs);
                    if (r < fe - fs) { f = fs + r; v = vs; } else { f = fe; v = vs + (r - (fe - fs)); }
                    if (m >= total) { f = nf; v = nv; }
                };
                for (int g0 = 0; g0 < ng; g0 += 64) {
                    const int g = g0 + (int)lane;
                    int f0, v0, f1, v1;
                    split(g * 64, f0, v0);
                    split(g * 64 + 64, f1, v1);
                    if (g < ng) out[g] = make_int4(rg.x + f0, rg.x + f1, rg.z + v0, rg.z + v1);
                }
                if (lane == 0) p.home_ngroups[h] = ng;
                // the slot ranges of the block's work items (what the tile kernels would otherwise fetch from the
                // first and the last group of the item, behind one more dependent load)
                const int2 hi = p.home_items[h];
                const unsigned n_items = p.ctl->n_items;
                for (int k0 = 0; k0 < hi.y; k0 += 64) {
                    const int k = k0 + (int)lane;
                    const unsigned it = (unsigned)(hi.x + k);
                    const bool on = k < hi.y && it < n_items;   // (slab pool exhausted: the item is not scheduled)
                    const int4 d = on ? p.item_desc[it] : make_int4(0, 0, 0, 0);
                    int f0, v0, f1, v1;
                    split(d.y * 64, f0, v0);
                    split(d.z * 64, f1, v1);
                    if (on) p.item_rng[p.item_pos[it]] = make_int4(rg.x + f0, rg.x + f1, rg.z + v0, rg.z + v1);
                }
            }
        }
    }
    const PSet& S = p.set[p.ctl->cur];
    const int nf_in = p.ctl->nfa + p.ctl->add_f, total = nf_in + p.ctl->nva + p.ctl->add_v;
    for (int idx = i; idx < total; idx += (int)(gridDim.x * 256u)) {
        const int j = active_slot(p, idx, nf_in);
        const uint32_t key = p.pkey[j];
        if (key == 0xFFFFFFFFu) {   // released: the particle leaves this rank
            p.dst_of[j] = -1;
            p.imap[S.pid[j]] = -1;
            continue;
        }
        const int t = j >= p.Nf;
        const int dst = (t ? p.Nf : 0) + p.blkstart[t][key >> 6] + p.cellcnt[t][key] + (int)p.prank[j];
        if ((unsigned)dst >= (unsigned)p.Np) {   // cannot happen with intact histograms (a diagnostic build can ablate them)
            atomicOr(&p.ctl->error, ERR_CAPACITY);
            continue;
        }
        // only the permutation is scattered (4 bytes per particle); the particle planes are moved by
        // k_rb_finish as a gather, whose writes are fully coalesced
        p.src_of[dst] = (uint32_t)j;
        p.dst_of[j] = dst;
        p.imap[S.pid[j]] = dst;
    }
}

// R3b (optional, mpm_set_deterministic): the counting sort places the particles of one (type, cell)
// segment in the arrival order of their atomics, which differs from run to run.  Sorting every
// segment by the previous slot makes the new order a pure function of the old one, hence the whole
// trajectory bitwise reproducible.  One thread per (home block, type, cell); segments are short
// (tens of particles) and sit in L2.
__global__ __launch_bounds__(256) void k_rb_canon(DP p) {
    Ctl* c = p.ctl;
    if (!c->need_rebuild) return;
    const PSet& S = p.set[c->cur];
    const unsigned n = c->n_home * 128u;
    for (unsigned w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
        const unsigned h = w >> 7, t = (w >> 6) & 1u, cell = w & 63u;
        const uint32_t b = p.home_block[h];
        const int4 rg = p.home_range[h];
        const int* pre = p.cellcnt[t] + (size_t)b * 64;      // exclusive prefixes at this point
        const int lo = (t ? rg.z : rg.x) + pre[cell];
        const int hi = cell < 63u ? (t ? rg.z : rg.x) + pre[cell + 1] : (t ? rg.w : rg.y);
        // insertion sort of src_of[lo, hi)
        for (int j = lo + 1; j < hi; ++j) {
            const uint32_t v = p.src_of[j];
            int k = j - 1;
            while (k >= lo && p.src_of[k] > v) {
                p.src_of[k + 1] = p.src_of[k];
                --k;
            }
            p.src_of[k + 1] = v;
        }
        for (int j = lo; j < hi; ++j) {
            const uint32_t src = p.src_of[j];
            p.dst_of[src] = j;
            p.imap[S.pid[src]] = j;
        }
    }
}

// R4: refresh face -> vertex slots, re-zero the histograms, flip the sets.  Launched with a
// fixed, small number of workgroups (grid-stride) so that the closing ticket costs few atomics.
__global__ __launch_bounds__(256) void k_rb_finish(DP p) {
    Ctl* c = p.ctl;
    if (!c->need_rebuild) return;
    const unsigned gs = gridDim.x * 256u, i0 = xcd_chunk(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
    const PSet& S = p.set[c->cur];
    const PSet& D = p.set[c->cur ^ 1];
    // move the particle records into their sorted slots (16-byte gathers, coalesced 16-byte stores);
    // the slot references inside a face's record (its corner vertices) are translated through dst_of, whose
    // accesses stay local because mesh neighbours were neighbours in the old order too
    const int nf_out = c->nfa_new, total_out = nf_out + c->nva_new;
    for (unsigned idx = i0; idx < (unsigned)total_out; idx += gs) {
        const unsigned j = (unsigned)active_slot(p, (int)idx, nf_out);
        const unsigned i = p.src_of[j];
        // (a face particle's x / v records are not moved when CalcFemStateAndForce is the next thing to run -- it
        // writes both before anything reads them: DP::lean_resort, the re-sorts of whole substeps)
        const bool xv = !(p.lean_resort && j < (unsigned)p.Nf);
        const float4 a2 = S.q[2][i], a3 = S.q[3][i];
        const int pid = S.pid[i];
        if (xv) {
            const float4 a0 = S.q[0][i], a1 = S.q[1][i];
            D.q[0][j] = a0; D.q[1][j] = a1;
        }
        D.q[2][j] = a2; D.q[3][j] = a3;
        D.pid[j] = pid;
        if (j < (unsigned)p.Nf) {
            const float4 b0 = S.fq[0][i], b1 = S.fq[1][i], b2 = S.fq[2][i];
            float4 b3 = S.fq[3][i];
            if (!p.dist.on) {
                b3.y = __int_as_float(p.dst_of[__float_as_int(b3.y)]);
                b3.z = __int_as_float(p.dst_of[__float_as_int(b3.z)]);
                b3.w = __int_as_float(p.dst_of[__float_as_int(b3.w)]);
            } else {
                // partitioned domain: particles come and go, so the references are rebuilt from the face's corner
                // ids (per-slot topology, carried along) through the new id -> slot map; -1 = that corner is not here
                const int4 cg = p.fg[c->cur][i];
                p.fg[c->cur ^ 1][j] = cg;
                b3.x = __int_as_float(cg.w);   // (the corners' ranks around their vertices: DP::VF; a face that has just arrived has none in fq)
                b3.y = __int_as_float(p.imap[cg.x]);
                b3.z = __int_as_float(p.imap[cg.y]);
                b3.w = __int_as_float(p.imap[cg.z]);
            }
            D.fq[0][j] = b0; D.fq[1][j] = b1; D.fq[2][j] = b2; D.fq[3][j] = b3;
            D.f8[j] = S.f8[i];
            D.c8[j] = S.c8[i];
        } else if (!p.dist.on) {
            // The vertex's entries of DP::VF at the new slot: zeros where it has no face (k_fem rewrites the others
            // before anything reads them), or the mark that sends a vertex with more than eight faces to the CSR.
            const int vo = pid - p.NfG;
            const int valence = p.adj_off[vo + 1] - p.adj_off[vo];
            {
                const size_t kk = (size_t)(j - p.Nf);
                const float3 z = make_float3(0.f, 0.f, 0.f);
                if (valence > 8) *reinterpret_cast<float3*>(p.VF + vf_entry((unsigned)kk, 0) * 3) = make_float3(__uint_as_float(VF_MARK), 0.f, 0.f);
                // (entries below the valence are k_fem's to write, before anything reads them: for the inner vertices
                // of a cloth only planes 6 and 7 are touched here)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q >= valence) *reinterpret_cast<float3*>(p.VF + vf_entry((unsigned)kk, (unsigned)q) * 3) = z;
            }
        } else {
            // Partitioned domain: the vertex's adjacent faces by original id travel with it (ascending face id, -1 = none,
            // -2 in the first = more than eight: the scene's adjacency, vertex_force_csr).  Its row of DP::VF at the new
            // slot as in a single domain: zeros past the valence, or the mark.  A face that is not on this rank is legal
            // around a ghost vertex (nobody uses its force) and an error around an owned one: particles come and go at
            // migrations only, each of which forces this re-sort, so looking here finds every case.
            const int4 g0 = p.vg[c->cur][0][i - p.Nf], g1 = p.vg[c->cur][1][i - p.Nf];
            p.vg[c->cur ^ 1][0][j - p.Nf] = g0;
            p.vg[c->cur ^ 1][1][j - p.Nf] = g1;
            const int fcs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const size_t kk = (size_t)(j - p.Nf);
            const float3 z = make_float3(0.f, 0.f, 0.f);
            bool missing = false;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (fcs[q] >= 0) missing |= p.imap[fcs[q] >> 2] < 0;
                else if (!(q == 0 && fcs[0] == -2)) *reinterpret_cast<float3*>(p.VF + vf_entry((unsigned)kk, (unsigned)q) * 3) = z;
            }
            if (fcs[0] == -2) *reinterpret_cast<float3*>(p.VF + vf_entry((unsigned)kk, 0) * 3) = make_float3(__uint_as_float(VF_MARK), 0.f, 0.f);
            if (missing && S.q[0][i].w > 0.f) atomicOr(&c->error, ERR_HALO);
        }
    }
    // every cell row of a home block holds prefix values now: clear whole rows
    const unsigned n_home = c->n_home;
    for (unsigned w = i0; w < n_home * 32u; w += gs) {
        const unsigned h = w >> 5, t = (w >> 4) & 1u, k = w & 15u;
        const uint32_t b = p.home_block[h];
        reinterpret_cast<int4*>(p.cellcnt[t] + (size_t)b * 64)[k] = make_int4(0, 0, 0, 0);
        if (k == 0) p.blkcnt[t][b] = 0;
        if (k == 1 && t == 0) p.home_bits[b >> 5] = 0;
    }
    // The last workgroup to get here flips the particle sets.  Every thread has read `cur` and
    // `need_rebuild` by now; the end of the kernel makes the flip visible to the next one.
    // All workgroups arrive within a few microseconds of each other and same-address device
    // atomics serialise at ~60 ns each, so the count is taken in two levels: 32 counters (one
    // cache line each) and a final one.
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned lanes = min(gridDim.x, 32u), grp = blockIdx.x & 31u;
        const unsigned members = (gridDim.x - grp + 31u) / 32u;
        if (atomicAdd(&p.tickets[grp * 32u], 1u) == members - 1u) {
            p.tickets[grp * 32u] = 0;
            if (atomicAdd(&c->ticket, 1u) == lanes - 1u) {
                c->ticket = 0;
                c->cur ^= 1;
                c->need_rebuild = 0;
                c->rebuilds += 1;
                c->nfa = c->nfa_new;
                c->nva = c->nva_new;
                c->add_f = c->add_v = 0;
                // k_rb_count's minimum (0 where it was not taken: a partitioned domain)
                unsigned q = 0u;
                for (unsigned k = 0; k < 32u; ++k) {
                    q = max(q, p.tickets[k * 32u + 1u]);
                    p.tickets[k * 32u + 1u] = 0u;
                }
                c->quiet_time = q ? __uint_as_float(~q) : 0.f;
                c->time_since_resort = 0.f;
            }
        }
    }
}

}  // namespace mpm
