// Stable LSD radix sort of (key, value) pairs on the device, 8 to 11 bits per pass (as few passes as
// the key width allows: a pass is three launches).
//
// Used where an order has to be *stable* and reproducible: the slot order of
// RebuildMapping(sort = true) (the reference's 16-bit radix sort,
// radix_sort.cuh / cuda_mpm_solver.cu:47-68) and the cell order of the contact
// pairs in UpdateContact.  The engine's own particle order does not use it (see
// mpm_rebuild.h: that one is a counting sort fused with the block tables).
//
// A workgroup of 4 waves owns a tile of 64 * items consecutive pairs (a quarter of its chunks per wave):
//   k_sort_hist    digit histogram of the tile                -> hist[tile][digit]
//   k_sort_scatter every workgroup derives ITS OWN output offsets from the histograms of all tiles (coalesced rows of
//                  2^DB ints, all loads independent: the whole table is a few hundred KB of L2 reads per workgroup),
//                  then ranks the tile's pairs chunk by chunk with wave ballots (lanes in order, chunks in order =>
//                  stable) and writes them to their final position
// Two launches per pass.  (Rounds 1 - 4 had a single-workgroup scan of the table between the two: 8 - 11 us per pass of
// one CU walking 40k - 80k entries 16384 at a time, more than the two kernels around it together.)
// No global atomics; LDS holds the running offsets of every wave of the tile.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <utility>

#include "mpm_device.h"

namespace mpm {

// A tile of 64 * items consecutive pairs belongs to one workgroup of SORT_WAVES waves; wave w owns the
// w-th quarter of the tile's chunks (contiguous, so that the order of the waves is the order of the pairs).
constexpr int SORT_WAVES = 4;

template <int DB>
// (n_dev: the number of pairs is on the device -- the contact pairs of mpm_generate_contact_pairs --; `n` is then an
// upper bound that sized the grid, tiles beyond the count do nothing)
__global__ __launch_bounds__(64 * SORT_WAVES) void k_sort_hist(const uint32_t* keys, int n, int shift, int items, int* hist,
                                                              int ntiles, const int* n_dev = nullptr) {
    constexpr int ND = 1 << DB;
    if (n_dev) n = min(n, *n_dev);
    __shared__ int s_cnt[ND];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tile = blockIdx.x;
    for (int d = tid; d < ND; d += 64 * SORT_WAVES) s_cnt[d] = 0;
    __syncthreads();
    const int per_wave = items / SORT_WAVES;
    const int begin = tile * 64 * items + wv * per_wave * 64;
    for (int it = 0; it < per_wave; ++it) {
        const int i = begin + it * 64 + lane;
        if (i < n) atomicAdd(&s_cnt[(keys[i] >> shift) & (uint32_t)(ND - 1)], 1);
    }
    __syncthreads();
    for (int d = tid; d < ND; d += 64 * SORT_WAVES) hist[(size_t)tile * ND + d] = s_cnt[d];
}

// Exclusive scan of one int per thread across a 1024-thread workgroup (shared by the scans below).
MPM_DEV int wg1024_exclusive(int v, int& total, int* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    __syncthreads();   // s_w may still be read from the previous call
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

// exclusive scan of `total` ints in place, one 1024-thread workgroup; every thread owns 16
// consecutive entries (four 16-byte loads) of each 16384-entry block.  `a` must be 16-byte
// aligned and padded to a multiple of 4 entries.
MPM_DEV void wg1024_scan_inplace(int* a, int total, int* s_w /* [16] shared */) {
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < total; base += 16384) {
        int4 v[4];
        int sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = base + tid * 16 + q * 4;
            v[q] = i < total ? *reinterpret_cast<const int4*>(a + i) : make_int4(0, 0, 0, 0);
            if (i + 1 >= total) v[q].y = 0;
            if (i + 2 >= total) v[q].z = 0;
            if (i + 3 >= total) v[q].w = 0;
            sum += v[q].x + v[q].y + v[q].z + v[q].w;
        }
        int block_total;
        int run = carry + wg1024_exclusive(sum, block_total, s_w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = base + tid * 16 + q * 4;
            int4 o;
            o.x = run; run += v[q].x;
            o.y = run; run += v[q].y;
            o.z = run; run += v[q].z;
            o.w = run; run += v[q].w;
            if (i + 3 < total) {
                *reinterpret_cast<int4*>(a + i) = o;
            } else {
                if (i < total) a[i] = o.x;
                if (i + 1 < total) a[i + 1] = o.y;
                if (i + 2 < total) a[i + 2] = o.z;
            }
        }
        carry += block_total;
    }
}
__global__ __launch_bounds__(1024) void k_sort_scan(int* a, int total) {
    __shared__ int s_w[16];
    wg1024_scan_inplace(a, total, s_w);
}

// Large exclusive scan: k_scan_blocks scans 4096-entry blocks in place and records their totals,
// k_sort_scan scans the totals, k_scan_add adds them back.  `n` entries, padded storage to a
// multiple of 4096.
__global__ __launch_bounds__(256) void k_scan_blocks(int* a, int n, int* sums) {
    __shared__ int s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * 4096 + tid * 16;
    int4 v[4];
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = base + q * 4;
        v[q] = *reinterpret_cast<const int4*>(a + i);
        if (i >= n) v[q].x = 0;
        if (i + 1 >= n) v[q].y = 0;
        if (i + 2 >= n) v[q].z = 0;
        if (i + 3 >= n) v[q].w = 0;
        sum += v[q].x + v[q].y + v[q].z + v[q].w;
    }
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
    if (tid == 255) sums[blockIdx.x] = run + sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int4 o;
        o.x = run; run += v[q].x;
        o.y = run; run += v[q].y;
        o.z = run; run += v[q].z;
        o.w = run; run += v[q].w;
        *reinterpret_cast<int4*>(a + base + q * 4) = o;
    }
}
__global__ __launch_bounds__(256) void k_scan_add(int* a, const int* sums) {
    const int off = sums[blockIdx.x];
    int4* q = reinterpret_cast<int4*>(a + blockIdx.x * 4096) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int4 v = q[k * 256];
        v.x += off; v.y += off; v.z += off; v.w += off;
        q[k * 256] = v;
    }
}

template <int DB>
__global__ __launch_bounds__(64 * SORT_WAVES) void k_sort_scatter(const uint32_t* keys, const uint32_t* vals, uint32_t* keys_out,
                                                                 uint32_t* vals_out, int n, int shift, int items, const int* hist,
                                                                 int ntiles, const int* n_dev = nullptr) {
    constexpr int ND = 1 << DB;
    if (n_dev) n = min(n, *n_dev);
    __shared__ int s_off[SORT_WAVES][ND];   // per wave and digit: first the count, then the first output position
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tile = blockIdx.x;
    for (int d = tid; d < SORT_WAVES * ND; d += 64 * SORT_WAVES) (&s_off[0][0])[d] = 0;
    __syncthreads();
    const int per_wave = items / SORT_WAVES;
    const int begin = tile * 64 * items + wv * per_wave * 64;
    // digits of this wave's pairs, counted per wave ...
    for (int it = 0; it < per_wave; ++it) {
        const int i = begin + it * 64 + lane;
        if (i < n) atomicAdd(&s_off[wv][(keys[i] >> shift) & (uint32_t)(ND - 1)], 1);
    }
    __syncthreads();
    // ... and turned into every wave's first output position per digit.  The tile's own first position of digit d is
    //   (pairs with a smaller digit, all tiles) + (pairs with digit d in the tiles before this one):
    // both from the table hist[tile][digit] of ALL tiles, read here in coalesced rows (thread = digit), every load
    // independent of the others -- what a single-workgroup scan kernel between the two launches used to produce.
    __shared__ int s_tot[ND];       // pairs with digit d, all tiles -> exclusive prefix over the digits
    __shared__ int s_before[ND];    // pairs with digit d in the tiles before this one
    __shared__ int s_scan[SORT_WAVES];
    {
        // A row of the table is ND ints: LPR lanes read it as one int4 each, the workgroup's G = T / LPR lane groups take
        // every G-th row (ND = 2048: two int4 per lane).  All loads of a lane are independent: sixteen rows in flight per
        // step.  (Round 5's first version looped over the digits in steps of 256 AND over the rows in steps of eight, each
        // step waiting for its loads: 36 dependent round trips to L2, 17.7 us of a 22 us pass at 69 tiles.)
        constexpr int T = 64 * SORT_WAVES;
        constexpr int LPR = ND / 4 < T ? ND / 4 : T, G = T / LPR, VPT = (ND / 4 + T - 1) / T;
        const int lr = tid % LPR, grp = tid / LPR;
        int4 tot[VPT], bef[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) tot[v] = bef[v] = make_int4(0, 0, 0, 0);
        if (G > 1) {
            for (int d = tid; d < ND; d += T) s_tot[d] = s_before[d] = 0;
            __syncthreads();
        }
        // (explicit batches: left as a plain loop the compiler waits for every row before it requests the next -- 69
        // round trips to L2, 16 us)
        auto rows = [&](int t0, auto KC) {
            constexpr int K = decltype(KC)::value;
            int4 x[K][VPT];
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int v = 0; v < VPT; ++v) x[q][v] = reinterpret_cast<const int4*>(hist + (size_t)(t0 + q * G) * ND)[lr + v * LPR];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const bool pre = t0 + q * G < tile;
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    tot[v].x += x[q][v].x; tot[v].y += x[q][v].y; tot[v].z += x[q][v].z; tot[v].w += x[q][v].w;
                    bef[v].x += pre ? x[q][v].x : 0; bef[v].y += pre ? x[q][v].y : 0;
                    bef[v].z += pre ? x[q][v].z : 0; bef[v].w += pre ? x[q][v].w : 0;
                }
            }
        };
        int t = grp;
        constexpr int KB = VPT > 1 ? 8 : 16;
        for (; t + (KB - 1) * G < ntiles; t += KB * G) rows(t, std::integral_constant<int, KB>{});
        for (; t + 3 * G < ntiles; t += 4 * G) rows(t, std::integral_constant<int, 4>{});
        for (; t < ntiles; t += G) rows(t, std::integral_constant<int, 1>{});
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int d = (lr + v * LPR) * 4;
            if (G > 1) {   // (integer sums: the order of the groups' additions does not matter)
                atomicAdd(&s_tot[d], tot[v].x); atomicAdd(&s_tot[d + 1], tot[v].y); atomicAdd(&s_tot[d + 2], tot[v].z); atomicAdd(&s_tot[d + 3], tot[v].w);
                atomicAdd(&s_before[d], bef[v].x); atomicAdd(&s_before[d + 1], bef[v].y); atomicAdd(&s_before[d + 2], bef[v].z); atomicAdd(&s_before[d + 3], bef[v].w);
            } else {
                s_tot[d] = tot[v].x; s_tot[d + 1] = tot[v].y; s_tot[d + 2] = tot[v].z; s_tot[d + 3] = tot[v].w;
                s_before[d] = bef[v].x; s_before[d + 1] = bef[v].y; s_before[d + 2] = bef[v].z; s_before[d + 3] = bef[v].w;
            }
        }
    }
    __syncthreads();
    // exclusive prefix of s_tot over the digits (ND <= 2048: every thread owns ND / 256 consecutive digits)
    {
        constexpr int PER = (ND + 64 * SORT_WAVES - 1) / (64 * SORT_WAVES);
        int loc[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int d = tid * PER + q;
            loc[q] = d < ND ? s_tot[d] : 0;
            sum += loc[q];
        }
        // scan of the per-thread sums across the workgroup (4 waves)
        int inc = sum;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int t = __shfl_up(inc, dd);
            if (lane >= dd) inc += t;
        }
        if (lane == 63) s_scan[wv] = inc;
        __syncthreads();
        int pre = inc - sum;
        for (int w = 0; w < wv; ++w) pre += s_scan[w];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int d = tid * PER + q;
            if (d < ND) s_tot[d] = pre;
            pre += loc[q];
        }
    }
    __syncthreads();
    // per-wave counts -> per-wave first positions: the tile's, then wave by wave
    for (int d = tid; d < ND; d += 64 * SORT_WAVES) {
        int run = s_tot[d] + s_before[d];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const int cnt = s_off[w][d];
            s_off[w][d] = run;
            run += cnt;
        }
    }
    __syncthreads();
    int* off = s_off[wv];   // (from here on every wave works on its own: wave-level fences only)
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int it = 0; it < per_wave; ++it) {
        const int i = begin + it * 64 + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? keys[i] : 0u;
        const uint32_t val = valid ? vals[i] : 0u;
        const uint32_t digit = (key >> shift) & (uint32_t)(ND - 1);
        // lanes of this chunk with the same digit
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const unsigned long long m = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? m : ~m;
        }
        int dst = 0;
        if (valid) dst = off[digit] + (int)__popcll(same & lt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the last lane of every digit group moves the running offset on
        if (valid && (same >> lane) == 1ull) off[digit] = dst + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[dst] = key;
            vals_out[dst] = val;
        }
    }
}

}  // namespace mpm

// In-place exclusive scan of n ints (storage padded to a multiple of 4096); sums: n/4096 + 2 ints,
// sums[nblocks] receives the grand total.
static int device_exclusive_scan(hipStream_t s, int* a, size_t n, int* sums) {
    using namespace mpm;
    const int nb = (int)((n + 4095) / 4096);
    if (hipMemsetAsync(sums + nb, 0, 4, s) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(256), 0, s, a, (int)n, sums);
    hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, s, sums, nb + 1);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, s, a, (const int*)sums);
    return 0;
}

static inline int sort_items_for(size_t n) { return n > (1u << 18) ? 64 : 16; }

constexpr int SORT_MAX_DIGIT_BITS = 11;
// ints of histogram scratch radix_sort_pairs needs for n pairs
static inline size_t sort_hist_ints(size_t n) {
    const size_t items = sort_items_for(n);
    const size_t tiles = (n + 64 * items - 1) / (64 * items);
    return ((size_t)1 << SORT_MAX_DIGIT_BITS) * (tiles + 1);
}

// ... for ANY count up to n (the tile size changes at 2^18 pairs: a count just below makes four times the tiles of one
// just above)
static inline size_t sort_hist_ints_upto(size_t n) {
    const size_t knee = (size_t)1 << 18;
    return n > knee ? std::max(sort_hist_ints(n), sort_hist_ints(knee)) : sort_hist_ints(n);
}

template <int DB>
static void radix_pass(hipStream_t s, const uint32_t* ki, const uint32_t* vi, uint32_t* ko, uint32_t* vo, int* hist, int n,
                       int shift, int items, int ntiles, const int* n_dev) {
    using namespace mpm;
    hipLaunchKernelGGL(k_sort_hist<DB>, dim3(ntiles), dim3(64 * SORT_WAVES), 0, s, ki, n, shift, items, hist, ntiles, n_dev);
    hipLaunchKernelGGL(k_sort_scatter<DB>, dim3(ntiles), dim3(64 * SORT_WAVES), 0, s, ki, vi, ko, vo, n, shift, items, (const int*)hist,
                       ntiles, n_dev);
}

// Sorts n pairs by the key bits [0, bits) (stable).  `a` holds the input; `b` and `hist` are scratch (hist:
// sort_hist_ints(n) ints).  The result ends in `a` after an even number of passes and in `b` after an odd one:
// *in_b says which when the caller can live with either (no copy), else (in_b == nullptr) it is copied back to `a`.
// n_dev: the count lives on the device, n is its upper bound (see k_sort_hist).
static int radix_sort_pairs(hipStream_t s, uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, int* hist, size_t n,
                            int bits, bool* in_b = nullptr, const int* n_dev = nullptr) {
    using namespace mpm;
    if (in_b) *in_b = false;
    if (n < 2 || bits <= 0) return 0;
    const int items = sort_items_for(n);
    const int ntiles = (int)((n + (size_t)64 * items - 1) / ((size_t)64 * items));
    // digit width: a pass costs two launches (~11 us at the contact solve's sizes) plus, in every workgroup of the
    // scatter, a walk over the 2^db x ntiles table of all tiles' histograms (L2 reads, ~1 us per 64 KB)
    int db = 8, passes = (bits + 7) / 8;
    {
        double best = 1e30;
        for (int d = 8; d <= SORT_MAX_DIGIT_BITS; ++d) {
            const int ps = (bits + d - 1) / d;
            const double cost = ps * (11.0 + (double)(((size_t)1 << d) * ntiles * 4) / 65536.0);
            if (cost < best) {
                best = cost;
                db = d;
                passes = ps;
            }
        }
    }
    uint32_t *ki = ka, *vi = va, *ko = kb, *vo = vb;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * db;
        switch (db) {
            case 8: radix_pass<8>(s, ki, vi, ko, vo, hist, (int)n, shift, items, ntiles, n_dev); break;
            case 9: radix_pass<9>(s, ki, vi, ko, vo, hist, (int)n, shift, items, ntiles, n_dev); break;
            case 10: radix_pass<10>(s, ki, vi, ko, vo, hist, (int)n, shift, items, ntiles, n_dev); break;
            default: radix_pass<11>(s, ki, vi, ko, vo, hist, (int)n, shift, items, ntiles, n_dev); break;
        }
        std::swap(ki, ko);
        std::swap(vi, vo);
    }
    if (ki != ka) {  // odd number of passes: the result sits in the scratch pair
        if (in_b) {
            *in_b = true;
        } else {
            if (hipMemcpyAsync(ka, ki, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
            if (hipMemcpyAsync(va, vi, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
        }
    }
    return 0;
}
