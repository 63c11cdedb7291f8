// The four kernels of a contact-free substep.
//   k_fem     : CalcFemStateAndForce        (cuda_mpm_kernels.cuh:183-294)
//   k_vforce  : vertex forces as a gather over adjacent faces (replaces the 9
//               float atomics per face of :287-292)
//   k_p2g     : ParticleToGrid              (cuda_mpm_kernels.cuh:418-543)
//   k_grid    : touched-block sum + UpdateGrid (cuda_mpm_kernels.cuh:545-796)
//   k_g2p     : GridToParticle              (cuda_mpm_kernels.cuh:798-924)
#pragma once
#include "mpm_device.h"

// (A/B switches used before their kernels are defined: all at the top, an undefined macro in an #if reads as 0)
#ifndef MPM_P2G_PEEL
#define MPM_P2G_PEEL 0
#endif
#ifndef MPM_FEM_SETPRIO
#define MPM_FEM_SETPRIO 1    // (A/B switch, round 4: no difference measured)
#endif
#ifndef MPM_G2P_ZPAIR
// the z components of the tile nodes broadcast by the packed instructions' operand select from the (z, m) pair (inline
// assembly: the compiler loads 12 bytes per node and then moves every z into a pair of its own): 27 moves less per
// particle, k_g2p 21.85 -> 20.95 us (event time, same box)
#define MPM_G2P_ZPAIR 1
#endif
#ifndef MPM_G2P_PREFETCH
#define MPM_G2P_PREFETCH 1   // 0 (experiment): a particle's position is loaded when its turn comes
#endif

namespace mpm {

// ---------------------------------------------------------------------------
// FEM: one thread per face particle (slot order => coalesced face arrays).
// ---------------------------------------------------------------------------
// Gated substeps: the four re-sort launches only precede every few substeps of mpm_run_substeps (they
// cost ~3 us each when idle).  A substep enqueued without them checks here whether a re-sort is pending
// (raised by the G2P of an earlier substep): if so all of its kernels return at once and the host runs
// that substep again, with the re-sort, at its next synchronisation point.
MPM_DEV bool gated_out(const DP& p) {
    return ((p.gated & 1) && p.ctl->need_rebuild) || ((p.gated & 2) && (p.ctl->error & ERR_SLABS)) ||
           ((p.gated & 4) && (int)(p.ctl->watch_hit - p.watch_base) >= 0);
}

// FM: the arithmetic of the divisions and square roots (mpm_math.h: 0 = correctly rounded, the default; 1 = hardware
// approximation + one Newton step, mpm_set_fast_math)
template <int FM>
__global__ __launch_bounds__(256) void k_fem(DP p, float dt) {
    if (gated_out(p)) return;
    const unsigned nfa = (unsigned)p.ctl->nfa;
    const unsigned chunk = xcd_chunk_active(blockIdx.x, nfa);
    const unsigned i = chunk * 256 + threadIdx.x;
    if (chunk == 0xFFFFFFFFu || i >= nfa) return;
    const PSet& S = p.set[p.ctl->cur];
    // 104 bytes in (F 36, Dm^-1 | vol | corners 32, C 36) + the corner gathers; 116 bytes out (F 36, face x v 32,
    // tau factor a 12 -- the other one is F's normal column, see pack_F --, corner forces 36).  The face particle's own q[0] / q[1] are written, never read: its
    // volume comes from the static record, C8 from the c8 plane (see PSet).
    const float4 f0 = S.fq[0][i], f1 = S.fq[1][i], f2 = S.fq[2][i], f3 = S.fq[3][i];
    const float F8 = S.f8[i], C8 = S.c8[i];
    const unsigned s0 = (unsigned)__float_as_int(f3.y), s1 = (unsigned)__float_as_int(f3.z),
                   s2 = (unsigned)__float_as_int(f3.w);
    if (p.dist.on && (int)(s0 | s1 | s2) < 0) {
        // partitioned domain: a corner vertex of this face is not on this rank, so its state cannot
        // be advanced here.  The vertex band is wider than the face band by ghost_margin_cells exactly
        // so that this never happens; if it does, a mesh edge is longer than that margin.
        atomicOr(&p.ctl->error, ERR_HALO);
        const float nan = __int_as_float(0x7FC00000);
#pragma unroll
        for (int c = 0; c < 3; ++c) p.G3[(size_t)i * 3 + c] = make_float3(nan, nan, nan);
        p.ta[i] = make_float3(nan, nan, nan);
        return;
    }
    const float4 xa = S.q[0][s0], xb = S.q[0][s1], xc = S.q[0][s2];
    const float4 va = S.q[1][s0], vb = S.q[1][s1], vc = S.q[1][s2];
    const float4 q2 = S.q[2][i], q3 = S.q[3][i];
    // partitioned domain: the sign of q[0].w is the particle's role (ghost copies are negative) and changes
    // with migration; a single-domain engine never looks at the face particle's own record
    const float volw = p.dist.on ? S.q[0][i].w : f2.w;
    if (MPM_FEM_SETPRIO) __builtin_amdgcn_s_setprio(2);   // (a wave that has its data computes and stores ahead of waves still issuing loads)
    const float x0[3] = {xa.x, xa.y, xa.z}, x1[3] = {xb.x, xb.y, xb.z}, x2[3] = {xc.x, xc.y, xc.z};
    // the face particle sits at the centroid and moves with the mean velocity (:203-207);
    // vol and C8 ride along unchanged
    // (a third as a product: an IEEE division costs ten vector instructions, see f_rcp in mpm_math.h)
    const float third = FM == 0 ? 0.f : (1.f / 3.f);
    auto mean3 = [&](float a, float b, float c) { return FM == 0 ? (a + b + c) / 3.f : (a + b + c) * third; };
    S.q[0][i] = make_float4(mean3(xa.x, xb.x, xc.x), mean3(xa.y, xb.y, xc.y), mean3(xa.z, xb.z, xc.z), volw);
    S.q[1][i] = make_float4(mean3(va.x, vb.x, vc.x), mean3(va.y, vb.y, vc.y), mean3(va.z, vb.z, vc.z), C8);
    float F[9];
    unpack_F(f0, f1, F8, F);
    const float Dm0 = f2.x, Dm1 = f2.y, Dm3 = f2.z;   // Dm^-1 = [Dm0 Dm1; 0 Dm3]
    const float C[9] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, C8};
    const float vol = f2.w;

    // normal column evolves with the affine velocity field (:216-226)
    float cF[9];
    cF[0] = F[0]; cF[1] = F[1];
    cF[2] = (1.f + dt * C[0]) * F[2] + dt * C[1] * F[5] + dt * C[2] * F[8];
    cF[3] = F[3]; cF[4] = F[4];
    cF[5] = dt * C[3] * F[2] + (1.f + dt * C[4]) * F[5] + dt * C[5] * F[8];
    cF[6] = F[6]; cF[7] = F[7];
    cF[8] = dt * C[6] * F[2] + dt * C[7] * F[5] + (1.f + dt * C[8]) * F[8];
    project_strain<FM>(p.M, cF);
    // in-plane columns from the deformed edges (:230-250); the Dm^-1[2] = 0 terms are left out
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float e0 = x1[d] - x0[d], e1 = x2[d] - x0[d];
        cF[d * 3 + 0] = e0 * Dm0;
        cF[d * 3 + 1] = e0 * Dm1 + e1 * Dm3;
    }
    pack_F(cF, S.fq[0][i], S.fq[1][i], S.f8[i]);

    float P[9];
    cloth_dphi_dF<FM>(p.M, cF, P);
#pragma unroll
    for (int d = 0; d < 9; ++d) P[d] *= vol;
    // tau = (V P[:,2]) (x) F[:,2]  (:265-267), kept factored: the second factor is in fq[0] already
    p.ta[i] = make_float3(P[2], P[5], P[8]);
    // grad_N = Dm^-T [[-1,1,0],[-1,0,1]]  (:269-276)
    const float g00 = -Dm0, g01 = Dm0;
    const float g10 = -Dm1 - Dm3, g11 = Dm1, g12 = Dm3;
    float Gm[9];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float a = P[d * 3 + 0], b = P[d * 3 + 1];
        Gm[d * 3 + 0] = a * g00 + b * g10;
        Gm[d * 3 + 1] = a * g01 + b * g11;
        Gm[d * 3 + 2] = b * g12;
    }
    // one 12-byte record per corner, at its place among the vertex's entries (DP::VF) -- or, for a vertex with more than
    // eight faces, in the face's own triple of G3 (a corner that is not on this rank of a partitioned domain: nowhere; that
    // face belongs to a ghost band's outer edge, or the missing corner is an error the kernel has raised above)
    const unsigned jb = (unsigned)__float_as_int(f3.x);
    const unsigned sc[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float3 rec = make_float3(Gm[c], Gm[3 + c], Gm[6 + c]);
        const unsigned j = (jb >> (4 * c)) & 15u;
        if (j < 8u) {
            if (!p.dist.on || (int)sc[c] >= p.Nf) *reinterpret_cast<float3*>(p.VF + vf_entry(sc[c] - (unsigned)p.Nf, j) * 3u) = rec;
        } else {
            p.G3[(size_t)i * 3 + c] = rec;
        }
    }
}

// the force on vertex k from its entries of DP::VF; false = they say "walk the CSR" (nothing usable summed)
MPM_DEV bool vertex_force_vf(const DP& p, int k, float& f0, float& f1, float& f2) {
    // (one address for all eight loads: the planes of a chunk are 384 bytes apart, an immediate offset)
    const float* e0 = p.VF + vf_entry((unsigned)k, 0) * 3;
    float3 g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = *reinterpret_cast<const float3*>(e0 + (size_t)j * VF_CHUNK * 3);
    f0 = f1 = f2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // (ascending original face id, as vertex_force_from; an empty entry adds -0)
        f0 += -g[j].x;
        f1 += -g[j].y;
        f2 += -g[j].z;
    }
    return __float_as_uint(g[0].x) != VF_MARK;
}

// more than 8 faces around vertex k: walk the original adjacency (the triples of these faces' corners are in G3)
MPM_DEV void vertex_force_csr(const DP& p, const PSet& S, int k, float& f0, float& f1, float& f2) {
    const int s = p.Nf + k;
    f0 = f1 = f2 = 0.f;
    const int vo = S.pid[s] - p.NfG;
    for (int e = p.adj_off[vo]; e < p.adj_off[vo + 1]; ++e) {
        const int fc = p.adj_fc[e];
        const int fs = p.imap[fc >> 2];
        const float3 g = fs >= 0 ? p.G3[(size_t)fs * 3 + (fc & 3)]
                                 : make_float3(S.q[0][s].w > 0.f ? __int_as_float(0x7FC00000) : 0.f, 0.f, 0.f);
        f0 += -g.x;
        f1 += -g.y;
        f2 += -g.z;
    }
}

// Vertex force = - sum over adjacent (face, corner) of that corner's force triple, summed in ascending original face
// id (the order sequential atomics would produce): the vertex's row of DP::VF, or the adjacency CSR for a vertex with
// more than eight faces.
MPM_DEV void vertex_force_value(const DP& p, const PSet& S, int k, float& f0, float& f1, float& f2) {
    if (!vertex_force_vf(p, k, f0, f1, f2)) vertex_force_csr(p, S, k, f0, f1, f2);
}
MPM_DEV void vertex_force(const DP& p, const PSet& S, int k) {   // ... written to p.f
    float f0, f1, f2;
    vertex_force_value(p, S, k, f0, f1, f2);
    const int s = p.Nf + k;
    p.f[0][s] = f0;
    p.f[1][s] = f1;
    p.f[2][s] = f2;
}

// (the phase-by-phase API and the partitioned-domain chains; mpm_run_substeps lets k_p2g do this per work
// item, see k_p2g's FORCES)
__global__ __launch_bounds__(256) void k_vforce(DP p) {
    if (gated_out(p)) return;
    const unsigned nva = (unsigned)p.ctl->nva;
    const unsigned chunk = xcd_chunk_active(blockIdx.x, nva);
    const int k = (int)(chunk * 256 + threadIdx.x);
    if (chunk == 0xFFFFFFFFu || k >= (int)nva) return;
    vertex_force(p, p.set[p.ctl->cur], k);
}

// ---------------------------------------------------------------------------
// P2G
//
// One workgroup per work item (a home block, or a run of the wave groups of a
// heavy one) accumulates the block's (TILE_W)^3 node tile in LDS and stores it
// as a slab.  Inside the item the transfer is organised per
// base cell: all particles that share a base cell scatter to the same 27 nodes,
//     node(n) += sum_p w_n(p) * (m_p, q_p + Bdx_p * (i,j,k)_n)
// which is a small dense contraction over the cell's P particles: 16 staged columns per particle (3 x (momentum
// term, 3 affine terms), mass) against 27 weights.  It runs on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: exact f32
// FMA chains, the VALU rate, but the sum over particles needs no cross-lane shuffles and no per-particle LDS atomics).
// Every wave works on its own 64-particle groups, without workgroup barriers:
//   1. every lane loads one particle (four 16-byte records), finds its base cell in the tile and builds its columns,
//   2. the wave groups its 64 particles by base cell (ballot loop, ranks by v_mbcnt) and stages
//      them in a wave-private LDS area,
//   3. per cell: 4 particles per MFMA step, 2 MFMAs per step (nodes 0-15 and 16-26).  Since round 6 the staged
//      columns are the A operand and the weights the B operand (MPM_P2G_SWAP): a lane's four accumulator registers
//      are then the four TERMS (1, i, j, k) of one (node, component), folded in the lane with 4 products and 3
//      sums -- rounds 1-5 had nodes as rows and folded across a quad with DPP -- and added to the tile: 2 LDS
//      atomics (64 distinct words each) per cell, doubles or 64-bit fixed point (EXACT).
// Replaces the warp-segmented scatter of cuda_mpm_kernels.cuh:418-543; the
// order of particles inside a block is irrelevant.
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// A/B switches of the round-3 P2G experiments (defaults = what ships; scratch/ab_build.py builds variants,
// scratch/ab_run.py times them on one box; results in DESIGN_HISTORY.md section 8)
#ifndef MPM_P2G_STG16
#define MPM_P2G_STG16 1
#endif
#if MPM_P2G_STG16
// staged floats per particle: the 13 columns of Y that carry numbers, then fx, fy, fz in the three columns of the
// mass component that Y leaves empty (their products are discarded by a zero in `fac`)
constexpr int STG = 16, STG_FX = 13;
#else
constexpr int STG = 20, STG_FX = 16;  // 16 columns of Y, fx, fy, fz, pad
#endif

template <int CTRL>
MPM_DEV float quad_perm(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// LDS accumulation is done in 64-bit fixed point: on gfx950 a wave-wide ds_add_f32
// costs ~190 LDS cycles per instruction (measured, scratch/lds_atomic_bench.hip)
// against ~10 for ds_add_u64, and integer sums are exact and order independent.
// float -> Q-format int64 with single-precision instructions only: q = v * scale is exact (power
// of two); |q| = hi * 2^32 + lo with hi = floor(|q| / 2^32) and lo = |q| - hi * 2^32 in [0, 2^32),
// both exact because they are parts of the same 24-bit mantissa (doing this on the SIGNED value
// would form 2^32 - |q| for small negative q, which does not fit a float: +-128 quanta of noise).
// Below 2^24 quanta q has fractional bits: they are rounded to nearest even, so the conversion is
// unbiased whatever the particle count.  `worst` collects the bit pattern of the largest |q| (NaN and
// infinity have the largest patterns of all): the caller compares it with 2^62 once, at the end -- a
// boolean per call lives in a scalar register pair and costs scalar instructions in every loop.
MPM_DEV void lds_add_fixed(long long* a, float q, unsigned& worst) {   // q = value * scale
    const float aq = fabsf(q);
    const float h = floorf(aq * 0x1p-32f);
    const unsigned lo = (unsigned)rintf(fmaf(-h, 0x1p32f, aq));
    const unsigned long long mag = ((unsigned long long)(unsigned)h << 32) | lo;
    const unsigned long long fx = q < 0.f ? 0ull - mag : mag;
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(a), fx, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    worst = max(worst, __float_as_uint(aq));
}

// coefficients (c0 + c1 f + c2 f^2) of the quadratic B-spline weight of stencil offset a
MPM_DEV void bspline_coeff(int a, bool on, float& c0, float& c1, float& c2) {
    c0 = a == 0 ? 1.125f : (a == 1 ? -.25f : .125f);
    c1 = a == 0 ? -1.5f : (a == 1 ? 2.f : -.5f);
    c2 = a == 1 ? -1.f : .5f;
    if (!on) c0 = c1 = c2 = 0.f;
}

struct Stencil {
    int rx, ry, rz;      // base cell relative to the tile origin (block origin - FREE_ZONE)
    float fx[3];
    float wx[3], wy[3], wz[3];
    unsigned out_bits;   // rx | ry | rz before the clamp: > TILE_W - 3 when the base cell is outside the tile (MPM_ERR_DRIFT)
};

MPM_DEV Stencil make_stencil(const DP& p, float x, float y, float z, int ox, int oy, int oz) {
    Stencil s;
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
    const uint32_t bx = min(base_cell(x, p.dxinv), hi), by = min(base_cell(y, p.dxinv), hi),
                   bz = min(base_cell(z, p.dxinv), hi);
    s.fx[0] = x * p.dxinv - (float)bx;
    s.fx[1] = y * p.dxinv - (float)by;
    s.fx[2] = z * p.dxinv - (float)bz;
    bspline3(s.fx[0], s.wx);
    bspline3(s.fx[1], s.wy);
    bspline3(s.fx[2], s.wz);
    int rx = (int)bx - ox, ry = (int)by - oy, rz = (int)bz - oz;
    const int hi_h = TILE_W - 3;
    static_assert(((TILE_W - 3) & (TILE_W - 2)) == 0, "rx | ry | rz <= hi_h is only equivalent to all three in range for hi_h = 2^k - 1");
    s.out_bits = (unsigned)(rx | ry | rz);   // (a negative coordinate sets the high bits)
    rx = min(max(rx, 0), hi_h);
    ry = min(max(ry, 0), hi_h);
    rz = min(max(rz, 0), hi_h);
    s.rx = rx; s.ry = ry; s.rz = rz;
    return s;
}

// neighbour blocks (27-bit set) reached by the 3^3 stencil of base cell (rx, ry, rz) of a tile
// (an outer product of three 3-bit sets, formed with shifts and masks: this runs on the scalar unit once
// per cell group of every wave, and the scalar unit is shared by the 16 waves of a CU)
MPM_DEV unsigned tile_reach_mask(int rx, int ry, int rz) {
    // Per axis the stencil of base cell r in 0..7 (tile coordinates, block origin at FREE_ZONE = 2) reaches the
    // block offsets {-1,0}, {0}, {0,+1}, {+1} for r >> 1 = 0, 1, 2, 3: as 3-bit sets 3, 2, 6, 4.
    static_assert(FREE_ZONE == 2, "the packed tables below are for a free zone of 2 cells");
    const int ix = rx >> 1, iy = ry >> 1, iz = rz >> 1;
    // bit (a * 9 + b * 3 + c) = mx[a] & my[b] & mz[c].  z set replicated at bits 0, 3, 6 and y set spread over
    // bits 3b..3b+2, both 9 bits per table entry:
    const unsigned long long Z = 0x926D924DBull, Y = 0xE07E0703Full;
    const unsigned m9 = (unsigned)((Z >> (9 * iz)) & (Y >> (9 * iy))) & 0x1FFu;
    const unsigned rep = m9 * 0x40201u;                       // m9 at bits 0, 9, 18
    const int lo = (ix + 1) >> 1, hi = (ix >> 1) + 1;         // x offsets lo..hi
    const unsigned xexp = ((1u << (9 * (hi + 1))) - 1u) ^ ((1u << (9 * lo)) - 1u);
    return rep & xexp;
}

#ifndef MPM_P2G_WAVES
#define MPM_P2G_WAVES 8
#endif
#ifndef MPM_P2G_SETPRIO
#define MPM_P2G_SETPRIO 1
#endif
#ifndef MPM_P2G_DYNAMIC
#define MPM_P2G_DYNAMIC 0   // 1 (experiment): the waves of a workgroup take the item's groups from an LDS counter
#endif
#ifndef MPM_P2G_DESC_AHEAD
// the group descriptor alone fetched a group ahead (4 registers): nothing with the vertex forces in a prologue (49.2 vs
// 49.3 us); with the lazy forces a group's chain is descriptor -> adjacency -> corner records, and it is worth 1 us
#define MPM_P2G_DESC_AHEAD 1
#endif
#ifndef MPM_P2G_PIPE2
#define MPM_P2G_PIPE2 0      // 1 (experiment): MFMA operands fetched two steps ahead in two named register sets
#endif
#ifndef MPM_P2G_SWAP
// 1 (round 6): the contraction transposed -- the particles' 16 columns are the A operand of the MFMA and the 27 weights the
// B operand, so that a lane's four accumulator registers are the four TERMS (1, i, j, k) of one (node, component) instead of
// four node rows of one term: the fold over the terms is four products and three sums in the lane (same pairing as the
// quad exchange it replaces: bit-identical) instead of 4 + 6 selects + 3 DPP adds, per cell and MFMA.
#define MPM_P2G_SWAP 1
#endif
#ifndef MPM_P2G_REACH_TABLE
#define MPM_P2G_REACH_TABLE 1   // (round 6) the reach mask of a cell read from a per-lane table instead of ~25 scalar instructions
#endif
#ifndef MPM_P2G_LOOP3
#define MPM_P2G_LOOP3 1   // (round 6) first / middle / last steps of a cell as three bodies, see the contraction loop
#endif
#ifndef MPM_P2G_PREFETCH
// 0 (what ships since round 4): a group's records are loaded when its turn comes.  Rounds 1-3 fetched them one group
// ahead; with four waves per SIMD the other waves cover the two round trips anyway, and the 25 registers the prefetched
// records occupied through the contraction cost more than the wait: 128 VGPRs + 12 bytes of scratch -> 103, none;
// k_p2g 51.6 -> 49.4 us (event time), 20-substep window 105.0 -> 101.5 us (same-box A/B, scratch/ab_run.py).
#define MPM_P2G_PREFETCH 0
#endif
constexpr int P2G_WAVES = MPM_P2G_WAVES, P2G_THREADS = 64 * P2G_WAVES;
// two workgroups per CU: 8 waves each at <= 128 VGPRs (4 per SIMD), or 10 at <= 96 (5 per SIMD; -DMPM_P2G_WAVES=10)
// FORCES: where a vertex lane finds the internal force on its vertex.
//   0  in p.f (k_vforce ran before this kernel: the phase-by-phase API, meshes with a vertex of more than eight faces)
//   1  in the eight planes of DP::VF, summed here (one round trip of coalesced loads; a partitioned domain went through
//      per-slot adjacency records and G3 until round 5: two dependent round trips, eight gathers)
// 1 saves the k_vforce launch of every substep.  Rounds 2-3 computed the forces of ALL vertices of an item in a
// prologue of their own (two dependent round trips and a barrier that every workgroup of the launch walks through at
// the same time: ~6 us in which nothing else happens); now each vertex lane fetches them with its particle records:
// k_p2g 48.9 -> 45.3 us (event time, same box), -> 43.9 with the group descriptor a group ahead, -> 41 with the vertex-side records (DP::VF).
// EXACT: how the node sums of a work item are accumulated in its LDS tile.
//   1  64-bit fixed point (lds_add_fixed): integer sums, exact, independent of the order in which the waves' atomics
//      arrive -- what makes a trajectory reproducible to the bit in deterministic mode (mpm_set_deterministic), at 13
//      vector instructions per value and cell for the conversion (a third of the per-cell epilogue);
//   0  double precision (ds_add_f64: native on gfx950, 17-21 cycles per wave instruction, scratch/lds_atomic_bench.hip --
//      ds_add_f32 is the one that takes 194): one conversion instruction per value.  The sum of the <= ~100 float
//      contributions to a node is exact in double, hence independent of the order, as long as they span less than 2^29
//      in magnitude; beyond that its LAST bit (2^-53 of the sum, rounded to float afterwards) may depend on the order.
//      The default: without the canonical particle order of deterministic mode the order INSIDE a cell already
//      differs from run to run at float level.
// one step of the per-cell contraction: acc += W^T Y (MPM_P2G_SWAP: rows = the 16 columns of the staged particles,
// columns = 16 node rows) or acc += W Y^T (rows = nodes)
#if MPM_P2G_SWAP
#define MPM_P2G_MFMA(w, y, acc) __builtin_amdgcn_mfma_f32_16x16x4f32(y, w, acc, 0, 0, 0)
#else
#define MPM_P2G_MFMA(w, y, acc) __builtin_amdgcn_mfma_f32_16x16x4f32(w, y, acc, 0, 0, 0)
#endif
template <int FORCES, int EXACT>
__global__ __launch_bounds__(P2G_THREADS) __attribute__((amdgpu_waves_per_eu(P2G_WAVES / 2, P2G_WAVES / 2))) void k_p2g(DP p, float dt) {
    if (gated_out(p)) return;
    // chain substep: the entry counters of the halo send buffers, which the k_grid<0> behind this kernel fills
    if (blockIdx.x == 0 && threadIdx.x < 2 && p.halo_hdr[threadIdx.x]) p.halo_hdr[threadIdx.x][0] = 0u;
    __shared__ long long tile[TILE_N * 4];  // (mvx, mvy, mvz, m) per node: fixed point, or the bits of doubles (EXACT)
    // wave-private staging: 64 particles (+8 slack rows touched by the operand prefetch)
    __shared__ __attribute__((aligned(16))) float stage_all[P2G_WAVES][(64 + 8) * STG];
    __shared__ unsigned s_mask;
#if MPM_P2G_DYNAMIC
    __shared__ int s_next;   // next unclaimed group of the item (waves take groups as they finish, not round robin)
#endif
    static_assert(sizeof(long long) * TILE_N * 4 + sizeof(float) * P2G_WAVES * (64 + 8) * STG + 4 <= 81920,
                  "two workgroups per CU need <= 80 KB of LDS each");
    Ctl* ctl = p.ctl;
    const PSet& S = p.set[ctl->cur];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float gdt = p.M.gravity * dt;
    const float sdt = -dt * p.Dinv;
    float* stage = stage_all[wv];

    // ---- lane constants of the contraction ---------------------------------
    const int j16 = lane & 15, g4 = lane >> 4, tt = lane & 3, dcomp = (lane >> 2) & 3;
    (void)tt; (void)dcomp;   // (only the untransposed contraction, MPM_P2G_SWAP=0, uses them)
    float ax[2][3], ay[2][3], az[2][3];  // A operand: weight polynomials of node rows j16 and 16 + j16
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * t + j16;
        const bool on = n < 27;
        bspline_coeff(n / 9, on, ax[t][0], ax[t][1], ax[t][2]);
        bspline_coeff((n / 3) % 3, true, ay[t][0], ay[t][1], ay[t][2]);
        bspline_coeff(n % 3, true, az[t][0], az[t][1], az[t][2]);
    }
    const f32x2 cx0 = {ax[0][0], ax[1][0]}, cx1 = {ax[0][1], ax[1][1]}, cx2 = {ax[0][2], ax[1][2]};
    const f32x2 cy0 = {ay[0][0], ay[1][0]}, cy1 = {ay[0][1], ay[1][1]}, cy2 = {ay[0][2], ay[1][2]};
    const f32x2 cz0 = {az[0][0], az[1][0]}, cz1 = {az[0][1], az[1][1]}, cz2 = {az[0][2], az[1][2]};
    // epilogue: (1, i, j, k)[tt] of node row 16 t + 4 g4 + r, times the fixed-point scale of this lane's
    // component (a power of two: scaling before or after the sums gives the same bits)
    float fac[2][4];
    int delta[2];        // float offset of this lane's node/component in the tile, -1 if none
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#if MPM_P2G_SWAP
        // lane (j16, g4) of MFMA t ends up with the four terms of node 16 t + j16, component g4
        const int n = 16 * t + j16;
        const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            fac[t][r] = n >= 27 ? 0.f : (r == 0 ? 1.f : (float)(r == 1 ? ni : (r == 2 ? nj : nk)));
            if (MPM_P2G_STG16 && g4 == 3 && r != 0) fac[t][r] = 0.f;   // (columns 13..15 carry fx, fy, fz)
        }
        delta[t] = n < 27 ? ((ni * TILE_W + nj) * TILE_W + nk) * 4 + g4 : -1;
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * t + 4 * g4 + r;
            const int ni = n / 9, nj = (n / 3) % 3, nk = n % 3;
            fac[t][r] = n >= 27 ? 0.f : (tt == 0 ? 1.f : (float)(tt == 1 ? ni : (tt == 2 ? nj : nk)));
            if (MPM_P2G_STG16 && dcomp == 3 && tt != 0) fac[t][r] = 0.f;   // (columns 13..15 carry fx, fy, fz)
        }
        const int n = 16 * t + 4 * g4 + tt;
        delta[t] = n < 27 ? (((n / 9) * TILE_W + (n / 3) % 3) * TILE_W + n % 3) * 4 + dcomp : -1;
#endif
    }
    {
        const float fscale = EXACT ? (float)((MPM_P2G_SWAP ? g4 : dcomp) == 3 ? p.fix_m : p.fix_p) : 1.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) fac[t][r] *= fscale;
    }
#if MPM_P2G_REACH_TABLE
    // lane l = (ix, iy, iz) in 2-bit fields: the neighbour blocks reached from base cells (2 ix.., 2 iy.., 2 iz..) of the tile
    const unsigned reach_of_lane = tile_reach_mask(2 * (lane >> 4), 2 * ((lane >> 2) & 3), 2 * (lane & 3));
#endif
    // A step reads 4 staged rows; rows that do not belong to the cell (the next cell's particles,
    // rows never written) are masked in the B operand only, so every row must hold finite numbers
    for (int k = lane; k < (64 + 8) * STG; k += 64) stage[k] = 0.f;

    // Work items (a home block, or a run of the wave groups of a heavy one) are taken round-robin
    // from the heaviest-first order: workgroup w processes entries w, w + G, ...; with G resident
    // workgroups that is one heavy item each plus the light tail, and it needs neither a queue
    // atomic nor extra barriers per item.
    const unsigned n_items = ctl->n_items;
    for (unsigned q = blockIdx.x; q < n_items; q += gridDim.x) {
        __syncthreads();  // the previous item's slab has been written
        const int4 fa = p.item_flat[2 * q], fb = p.item_flat[2 * q + 1];
        const unsigned item = (unsigned)fa.x;
        const unsigned long long tb0 = (diag_flags(p) & 4) ? __builtin_readcyclecounter() : 0ull;
        for (int n = tid; n < TILE_N * 4; n += P2G_THREADS) tile[n] = 0;
        if (tid == 0) s_mask = 0;
#if MPM_P2G_DYNAMIC
        if (tid == 0) s_next = P2G_WAVES;
#endif
        int bx, by, bz;
        block_coords((uint32_t)fa.z, bx, by, bz);
        const int ox = bx * 4 - FREE_ZONE, oy = by * 4 - FREE_ZONE, oz = bz * 4 - FREE_ZONE;
        const int4 rg = make_int4(fb.x, fb.y, fb.z, 0);
        const int nfb = rg.y - rg.x;
        // Every wave streams through its own groups of <= 64 particles; no workgroup barrier inside.
        // Groups are runs of whole cells (faces and vertices of the same cells together), laid out
        // at the last rebuild: a group usually spans two base cells.
        const int ngroups = fa.w;
        const int4* groups = p.home_groups + fb.w;
        struct Raw {
            float x[3], v[3], vol, C[9];
            // tau factors a, b (faces) and force (vertices) in separate registers: merging them
            // into one array makes the compiler route the prefetch through scratch memory
            float ta[3], tb[3], frc[3];
            bool act, is_face;
        };
        // the force on a vertex particle from k_vforce's p.f
        auto force_of = [&](unsigned ii, float* f) {
            const float* fbase = p.f[0] + ii;
#pragma unroll
            for (int d = 0; d < 3; ++d) f[d] = fbase[(size_t)d * p.f_stride];
        };
        // a group's descriptor is fetched one group ahead of its records (which are fetched one group ahead of their
        // use): no dependent round trip in front of the record loads
        auto desc_of = [&](int g) { return groups[min(g, ngroups - 1)]; };
        int4 gd_next = make_int4(0, 0, 0, 0);
        auto load_raw = [&](int4 gr) {
            Raw r;
            const int gf = gr.y - gr.x, gn = gf + (gr.w - gr.z);
            r.act = lane < gn;
            r.is_face = lane < gf;
            // unsigned index: lets the loads use the scalar-base + 32-bit-offset addressing form
            const unsigned any_slot = (unsigned)(nfb ? rg.x : rg.z);   // (a particle of the item: always valid)
            const unsigned ii = r.act ? (unsigned)(r.is_face ? gr.x + lane : gr.z + (lane - gf)) : any_slot;
            // (one base pointer and a stride for the four planes: see DP::q_stride)
            const float4* qb = S.q[0] + ii;
            const float4 q0 = qb[0], q1 = qb[p.q_stride], q2 = qb[2 * (size_t)p.q_stride], q3 = qb[3 * (size_t)p.q_stride];
            r.x[0] = q0.x; r.x[1] = q0.y; r.x[2] = q0.z; r.vol = q0.w;
            r.v[0] = q1.x; r.v[1] = q1.y; r.v[2] = q1.z;
            unpack_C(q1, q2, q3, r.C);
            // Face lanes need the tau factors, vertex lanes the force.  Both are loaded by ALL lanes, from a
            // valid slot of the other kind where the lane has no use for them (one cache line for the wave), instead
            // of `if (face) load ab else load f` over zero-filled registers: the fill of registers that the other
            // lanes' loads are still writing made the compiler put `s_waitcnt vmcnt` in front of it, i.e. the wave
            // waited for its whole prefetch (a memory round trip per group) before starting the contraction.
            {
                const unsigned fi = r.is_face ? ii : (unsigned)(nfb ? rg.x : 0), vi = r.is_face ? any_slot : ii;
                const float3 a = p.ta[fi];
                const float3 b = *reinterpret_cast<const float3*>(&S.fq[0][fi]);   // F[:,2], see pack_F
                r.ta[0] = a.x; r.ta[1] = a.y; r.ta[2] = a.z; r.tb[0] = b.x; r.tb[1] = b.y; r.tb[2] = b.z;
                const bool vert = r.act && !r.is_face;
                if (FORCES == 1) {
                    vertex_force_vf(p, vert ? (int)ii - p.Nf : 0, r.frc[0], r.frc[1], r.frc[2]);   // (vertex 0 for the other lanes: valid memory)
                } else {
                    force_of(vi, r.frc);
                }
                // (p.f is what a caller downloads as the forces: between the substeps of one batch nobody can)
                if (FORCES != 0 && vert && !p.lean_g2p) {
                    p.f[0][ii] = r.frc[0]; p.f[1][ii] = r.frc[1]; p.f[2][ii] = r.frc[2];
                }
            }
            return r;
        };
        Raw cur;
        __syncthreads();
        unsigned mymask = 0;
        bool halo_bad = false;
        unsigned out_worst = 0, fix_worst = 0;   // error conditions, collected as integers in vector registers
        if (MPM_P2G_PREFETCH && wv < ngroups) {
            cur = load_raw(desc_of(wv));
            gd_next = desc_of(wv + P2G_WAVES);
        }
        // (without the record prefetch only the group DESCRIPTOR is fetched a group ahead: four registers, and the
        // records' loads no longer wait for a dependent round trip)
        if (!MPM_P2G_PREFETCH && MPM_P2G_DESC_AHEAD && wv < ngroups) gd_next = desc_of(wv);
        const bool prof = (diag_flags(p) & 4) != 0 && wv == 0;
        unsigned long long tq[3] = {0, 0, 0}, pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (prof && lane == 0) atomicAdd(&p.dbgbuf[14], (unsigned long long)__builtin_readcyclecounter() - tb0);   // prologue
#if MPM_P2G_DYNAMIC
        static_assert(!MPM_P2G_PREFETCH, "dynamic group assignment has no place for a prefetched group");
        auto next_group = [&]() {
            int n = 0;
            if (lane == 0) n = atomicAdd(&s_next, 1);
            return __builtin_amdgcn_readfirstlane(n);
        };
        for (int g = wv; g < ngroups; g = next_group()) {
#else
        for (int g = wv; g < ngroups; g += P2G_WAVES) {
#endif
            // ---- 1. one particle per lane (its raw state was prefetched) ----------
            if (prof) tq[0] = __builtin_readcyclecounter();
            if (!MPM_P2G_PREFETCH) {
                if (MPM_P2G_DESC_AHEAD) {
                    cur = load_raw(gd_next);
                    gd_next = desc_of(g + P2G_WAVES);
                } else {
                    cur = load_raw(desc_of(g));
                }
            }
            const bool act = cur.act, is_face = cur.is_face;
            const Stencil st = make_stencil(p, cur.x[0], cur.x[1], cur.x[2], ox, oy, oz);
            // partitioned domain: a ghost copy (vol < 0) scatters nothing, its owner does
            const bool own = cur.vol > 0.f;
            const float m = own ? cur.vol * p.M.density : 0.f;
            float Y[16];
            {
                float B[9];
                float fext[3] = {0.f, 0.f, 0.f};
                // (explicit fused multiply-adds where a sum of products could be fused in more than one way: left to the
                // compiler, the choice -- and the last bit of the result -- changes from build to build with the code around it)
#pragma unroll
                for (int r = 0; r < 9; ++r) B[r] = cur.C[r] * m;
                if (is_face) {
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) B[r * 3 + c] = fmaf(sdt, cur.ta[r] * cur.tb[c], B[r * 3 + c]);
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) fext[r] = cur.frc[r] * dt;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float qq = r == p.M.gravity_axis ? fmaf(cur.v[r], m, m * gdt) : cur.v[r] * m;
                    qq += fext[r];
                    const float dot = fmaf(B[r * 3 + 2], st.fx[2], fmaf(B[r * 3], st.fx[0], B[r * 3 + 1] * st.fx[1]));
                    qq = fmaf(-p.dx, dot, qq);
                    Y[r * 4 + 0] = qq;
                    Y[r * 4 + 1] = B[r * 3 + 0] * p.dx;
                    Y[r * 4 + 2] = B[r * 3 + 1] * p.dx;
                    Y[r * 4 + 3] = B[r * 3 + 2] * p.dx;
                }
                Y[12] = m; Y[13] = 0.f; Y[14] = 0.f; Y[15] = 0.f;
                if (p.dist.on && !own) {   // (only a partitioned domain has ghost copies: a scalar branch otherwise)
#pragma unroll
                    for (int k = 0; k < 12; ++k) Y[k] = 0.f;
                }
            }
            out_worst = max(out_worst, act ? st.out_bits : 0u);
            if (p.dist.on && act && own) {
                // the stencil of an owned particle must stay inside the blocks the neighbour receives
                const int gx = ox + st.rx;
                halo_bad |= gx < p.dist.own_lo - p.dist.zone_cells || gx + 2 >= p.dist.own_hi + p.dist.zone_cells;
            }
            // the raw registers are dead now: start the next group's loads, they complete while this
            // group goes through the LDS / matrix-pipe phases below (which never wait on vmcnt)
            if (MPM_P2G_PREFETCH && g + P2G_WAVES < ngroups) {
                cur = load_raw(gd_next);
                gd_next = desc_of(g + 2 * P2G_WAVES);
            }
            if (diag_flags(p) & 2) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += Y[k];
                if (acc == 1.2345e30f) tile[0] = (long long)acc;
                continue;
            }
            // ---- 2. group the wave's particles by base cell ---------------------
            const int key = (st.rx * 8 + st.ry) * 8 + st.rz;
            const unsigned long long actmask = __ballot(act);
            int pos = 0;
            {
                unsigned long long todo = actmask;
                int base = 0;
                while (todo) {
                    const int lk = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
                    const bool mine = key == lk;
                    const unsigned long long same = __ballot(mine) & todo;
                    // (an active lane whose key is lk is in `same`: a key leaves `todo` with all of its lanes at once; the
                    // position of an inactive lane is never used.  Rank inside the run: v_mbcnt counts the bits below the
                    // lane and adds the run's base in the same two instructions)
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, (unsigned)base));
                    if (mine) pos = rank;
                    base += (int)__popcll(same);
                    todo &= ~same;
                }
            }
            // ---- 3. stage at the grouped position (wave-private LDS, in-order) --
            if (act) {
                float4* sp = reinterpret_cast<float4*>(stage + pos * STG);
                sp[0] = make_float4(Y[0], Y[1], Y[2], Y[3]);
                sp[1] = make_float4(Y[4], Y[5], Y[6], Y[7]);
                sp[2] = make_float4(Y[8], Y[9], Y[10], Y[11]);
#if MPM_P2G_STG16
                sp[3] = make_float4(Y[12], st.fx[0], st.fx[1], st.fx[2]);
#else
                sp[3] = make_float4(Y[12], Y[13], Y[14], Y[15]);
                sp[4] = make_float4(st.fx[0], st.fx[1], st.fx[2], 0.f);
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (diag_flags(p) & 1) continue;
            // ---- 4. per-cell contraction on the matrix pipe ----------------------
            unsigned long long todo = actmask;
            int s0 = 0;
#if MPM_P2G_PIPE2
            // Operands two steps deep: set A feeds step s, set B step s + 4; A is re-loaded (for s + 8) while B's step
            // computes and the other way round, so every LDS read has a whole step of arithmetic to arrive in.  The
            // two sets are named, not rotated: no moves.  Rows beyond the staged ones are clamped to the last row
            // (they are masked in the B operand anyway).
            float afx, afy, afz, ay, bfx, bfy, bfz, by;
            auto ld = [&](int row, float& fx_, float& fy_, float& fz_, float& y_) {
                const float* sn = stage + min(row + g4, 64 + 8 - 1) * STG;
                fx_ = sn[STG_FX]; fy_ = sn[STG_FX + 1]; fz_ = sn[STG_FX + 2]; y_ = sn[j16];
            };
            ld(0, afx, afy, afz, ay);
            ld(4, bfx, bfy, bfz, by);
            if (prof) tq[1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_setprio(2);
            while (todo) {
                const int ckey = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
                const unsigned long long same = __ballot(key == ckey) & todo;
                todo &= ~same;
                const int s1 = s0 + (int)__popcll(same);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if (prof) { pc[4] += 1; pc[5] += (unsigned)((s1 - s0 + 3) >> 2); tq[2] = __builtin_readcyclecounter(); }
                auto step = [&](int s, float fx, float fy, float fz, float y) {
                    const bool ok = s + g4 < s1;
                    const f32x2 fx2 = {fx, fx}, fy2 = {fy, fy}, fz2 = {fz, fz};
                    const f32x2 w01 = __builtin_elementwise_fma(__builtin_elementwise_fma(cx2, fx2, cx1), fx2, cx0) *
                                      __builtin_elementwise_fma(__builtin_elementwise_fma(cy2, fy2, cy1), fy2, cy0) *
                                      __builtin_elementwise_fma(__builtin_elementwise_fma(cz2, fz2, cz1), fz2, cz0);
                    if (!ok) y = 0.f;
                    acc0 = MPM_P2G_MFMA(w01.x, y, acc0);
                    acc1 = MPM_P2G_MFMA(w01.y, y, acc1);
                };
                for (int s = s0; s < s1; s += 8) {
                    step(s, afx, afy, afz, ay);
                    if (s + 8 < s1) ld(s + 8, afx, afy, afz, ay);
                    if (s + 4 < s1) {
                        step(s + 4, bfx, bfy, bfz, by);
                        if (s + 12 < s1) ld(s + 12, bfx, bfy, bfz, by);
                    }
                }
                // the next cell starts at s1: its first two steps' operands arrive during this cell's epilogue
                ld(s1, afx, afy, afz, ay);
                ld(s1 + 4, bfx, bfy, bfz, by);
#else
            float nfx, nfy, nfz, ny;
            {
                const float* sn = stage + g4 * STG;
                nfx = sn[STG_FX]; nfy = sn[STG_FX + 1]; nfz = sn[STG_FX + 2]; ny = sn[j16];
            }
            if (prof) tq[1] = __builtin_readcyclecounter();
            if (MPM_P2G_SETPRIO) __builtin_amdgcn_s_setprio(2);   // (waves in the contraction keep the matrix pipe fed: ahead of waves that derive / group)
            while (todo) {
                const int ckey = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
                const unsigned long long same = __ballot(key == ckey) & todo;
                todo &= ~same;
                const int s1 = s0 + (int)__popcll(same);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if (prof) { pc[4] += 1; pc[5] += (unsigned)((s1 - s0 + 3) >> 2); tq[2] = __builtin_readcyclecounter(); }
                // operands of a step are fetched one step ahead (the first step's during the previous
                // cell's epilogue), so the LDS latency hides behind the MFMAs
#if !MPM_P2G_LOOP3
                auto one_step = [&](int s, bool first_step) {
                    const bool ok = g4 < s1 - s;   // (rows of this cell; the difference is wave-uniform: one vector instruction)
                    const float fx = nfx, fy = nfy, fz = nfz;
                    float y = ny;
                    {
                        const float* sn = stage + (s + 4 + g4) * STG;
                        nfx = sn[STG_FX]; nfy = sn[STG_FX + 1]; nfz = sn[STG_FX + 2]; ny = sn[j16];
                    }
                    // both rows' weights in one chain of packed operations (v_pk_fma_f32: the same fused
                    // multiply-adds as two scalar chains, half the issue slots)
                    const f32x2 fx2 = {fx, fx}, fy2 = {fy, fy}, fz2 = {fz, fz};
                    const f32x2 w01 = __builtin_elementwise_fma(__builtin_elementwise_fma(cx2, fx2, cx1), fx2, cx0) *
                                      __builtin_elementwise_fma(__builtin_elementwise_fma(cy2, fy2, cy1), fy2, cy0) *
                                      __builtin_elementwise_fma(__builtin_elementwise_fma(cz2, fz2, cz1), fz2, cz0);
                    float w0 = w01.x, w1 = w01.y;
                    if (diag_flags(p) & 128) { w0 = fx; w1 = fy; }
                    if (!ok) y = 0.f;   // (weights of foreign rows are finite: 0 * w = 0)
                    if (diag_flags(p) & 64) {
                        acc0[0] = fmaf(w0, y, acc0[0]);
                        acc1[0] = fmaf(w1, y, acc1[0]);
                    } else if (MPM_P2G_PEEL && first_step) {
                        // (the first step of a cell accumulates onto the constant 0 -- an inline operand of the MFMA --
                        // instead of onto eight registers that have to be cleared first)
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc0 = MPM_P2G_MFMA(w0, y, zero);
                        acc1 = MPM_P2G_MFMA(w1, y, zero);
                    } else {
                        acc0 = MPM_P2G_MFMA(w0, y, acc0);
                        acc1 = MPM_P2G_MFMA(w1, y, acc1);
                    }
                };
#endif
#if MPM_P2G_LOOP3
                {
                    // Three kinds of step: the first accumulates onto the inline constant 0 (no eight moves to clear the
                    // accumulators of every cell), only the last one has rows of the NEXT cell to mask (a cell's rows are
                    // contiguous: every row of an earlier step is the cell's own).  Same operations on the same operands.
                    auto step3 = [&](int s, bool first_step, bool last_step) {
                        const float fx = nfx, fy = nfy, fz = nfz;
                        float y = ny;
                        {
                            const float* sn = stage + (s + 4 + g4) * STG;
                            nfx = sn[STG_FX]; nfy = sn[STG_FX + 1]; nfz = sn[STG_FX + 2]; ny = sn[j16];
                        }
                        const f32x2 fx2 = {fx, fx}, fy2 = {fy, fy}, fz2 = {fz, fz};
                        const f32x2 w01 = __builtin_elementwise_fma(__builtin_elementwise_fma(cx2, fx2, cx1), fx2, cx0) *
                                          __builtin_elementwise_fma(__builtin_elementwise_fma(cy2, fy2, cy1), fy2, cy0) *
                                          __builtin_elementwise_fma(__builtin_elementwise_fma(cz2, fz2, cz1), fz2, cz0);
                        if (last_step && !(g4 < s1 - s)) y = 0.f;
                        if (first_step) {
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                            acc0 = MPM_P2G_MFMA(w01.x, y, zero);
                            acc1 = MPM_P2G_MFMA(w01.y, y, zero);
                        } else {
                            acc0 = MPM_P2G_MFMA(w01.x, y, acc0);
                            acc1 = MPM_P2G_MFMA(w01.y, y, acc1);
                        }
                    };
                    const int last = s0 + ((s1 - s0 - 1) & ~3);   // (a cell has at least one particle: s0 < s1)
                    if (last == s0) {
                        step3(s0, true, true);
                    } else {
                        step3(s0, true, false);
                        for (int s = s0 + 4; s < last; s += 4) step3(s, false, false);
                        step3(last, false, true);
                    }
                }
#else
                {
                    int s = (diag_flags(p) & 8) ? s1 : s0;
                    if (MPM_P2G_PEEL && s < s1) {   // (a cell has at least one particle: s0 < s1)
                        one_step(s, true);
                        s += 4;
                    }
                    for (; s < s1; s += 4) one_step(s, false);
                }
#endif
                // the loop leaves row block (last step + 4) preloaded; the next cell starts at s1
                if (((s1 - s0) & 3) != 0 || (diag_flags(p) & 8)) {
                    const float* sn = stage + (s1 + g4) * STG;
                    nfx = sn[STG_FX]; nfy = sn[STG_FX + 1]; nfz = sn[STG_FX + 2]; ny = sn[j16];
                }
#endif
                if (prof) { asm volatile("" :: "v"(acc0), "v"(acc1)); const unsigned long long tm = __builtin_readcyclecounter(); pc[2] += tm - tq[2]; tq[2] = tm; }
                s0 = s1;
                // rows = nodes, columns = (component d, term tt): fold the 4 terms of each component
                const int crx = ckey >> 6, cry = (ckey >> 3) & 7, crz = ckey & 7;
#if MPM_P2G_REACH_TABLE
                // the mask depends on (rx >> 1, ry >> 1, rz >> 1) only: 64 combinations, one per lane of `reach_of_lane`
                mymask |= (unsigned)__builtin_amdgcn_readlane((int)reach_of_lane, ((ckey >> 3) & 0x30) | ((ckey >> 2) & 0xC) | ((ckey >> 1) & 3));
#else
                mymask |= tile_reach_mask(crx, cry, crz);   // wave-uniform: scalar unit
#endif
                long long* tb = tile + ((crx * TILE_W + cry) * TILE_W + crz) * 4;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 a = t ? acc1 : acc0;
#if MPM_P2G_SWAP
                    // the four terms of this lane's (node, component): products, then the sums paired as (0 + 1) + (2 + 3)
                    float val;
                    {
#pragma clang fp contract(off)
                        const float x0 = a[0] * fac[t][0], x1 = a[1] * fac[t][1], x2 = a[2] * fac[t][2], x3 = a[3] * fac[t][3];
                        val = (x0 + x1) + (x2 + x3);
                    }
#else
                    // Lane (g4, dcomp, tt) of a quad holds the four terms' products of rows r = 0..3 and has to end up
                    // with the sum over the quad's four lanes of row r = tt: a 4 x 4 transpose-and-add in two
                    // exchanges (lane ^ 1, then lane ^ 2), each lane passing on what its partner keeps -- 4 + 2
                    // selects and 3 DPP adds instead of folding all four rows on every lane (8 DPP adds + 4 moves)
                    // and selecting afterwards.  Same operands in the same order: the sums are bit-identical.
                    const float x0 = a[0] * fac[t][0], x1 = a[1] * fac[t][1], x2 = a[2] * fac[t][2], x3 = a[3] * fac[t][3];
                    const bool b0 = (tt & 1) != 0, b1 = (tt & 2) != 0;
                    const float ua = (b0 ? x1 : x0) + quad_perm<0xB1>(b0 ? x0 : x1);   // row b0, lanes l and l ^ 1
                    const float ub = (b0 ? x3 : x2) + quad_perm<0xB1>(b0 ? x2 : x3);   // row 2 + b0
                    const float val = (b1 ? ub : ua) + quad_perm<0x4E>(b1 ? ua : ub);
#endif
                    if (delta[t] >= 0 && !(diag_flags(p) & 16)) {
                        if (EXACT) lds_add_fixed(tb + delta[t], val, fix_worst);
                        else __hip_atomic_fetch_add(reinterpret_cast<double*>(tb + delta[t]), (double)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (prof) pc[3] += __builtin_readcyclecounter() - tq[2];
            }
            if (prof) {
                const unsigned long long te = __builtin_readcyclecounter();
                pc[0] += tq[1] - tq[0];   // derive + group + stage
                pc[1] += te - tq[1];      // contraction phase
                pc[6] += 1;
            }
            __builtin_amdgcn_s_setprio(0);
            // the next group's staging writes must not overtake this group's reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (prof) pc[7] = __builtin_readcyclecounter() - tb0;  // whole block, before the final barrier + slab
        if (prof && lane == 0)
            for (int q = 0; q < 8; ++q) atomicAdd(&p.dbgbuf[q], pc[q]);
        if (mymask && lane == 0) atomicOr(&s_mask, mymask);
        if (__ballot(out_worst > (unsigned)(TILE_W - 3)) && lane == 0) atomicOr(&ctl->error, ERR_DRIFT);
        if (EXACT && __ballot(fix_worst >= __float_as_uint(0x1p62f)) && lane == 0) atomicOr(&ctl->error, ERR_RANGE);
        if (p.dist.on && __ballot(halo_bad) && lane == 0) atomicOr(&ctl->error, ERR_HALO);
        __syncthreads();
        if (prof && lane == 0) atomicAdd(&p.dbgbuf[15], (unsigned long long)__builtin_readcyclecounter() - tb0 - pc[7]);   // wave 0 at the closing barrier
        float4* out = p.slab + (size_t)item * TILE_N;
        bool not_finite = false;
        for (int n = tid; n < TILE_N; n += P2G_THREADS) {
            const long long* q = tile + n * 4;
            float4 o;
            if (EXACT) {
                o = make_float4((float)((double)q[0] * p.unfix_p), (float)((double)q[1] * p.unfix_p),
                                (float)((double)q[2] * p.unfix_p), (float)((double)q[3] * p.unfix_m));
            } else {
                const double* d = reinterpret_cast<const double*>(q);
                o = make_float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]);
                // (NaN or infinity in a sum: what the fixed-point conversion reports per contribution)
                not_finite |= !(fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + fabsf(o.w) < __int_as_float(0x7F800000));
            }
            out[n] = o;
        }
        if (!EXACT && __ballot(not_finite) && lane == 0) atomicOr(&ctl->error, ERR_RANGE);
        if (tid == 0) p.slab_mask[item] = s_mask;
        if ((diag_flags(p) & 4) && tid == 0) {
            atomicAdd(&p.dbgbuf[12], (unsigned long long)__builtin_readcyclecounter() - tb0);
            atomicAdd(&p.dbgbuf[13], 1ull);
        }
    }
}

// ---------------------------------------------------------------------------
// Grid: per active block, sum the overlapping tiles (fixed order => the grid is
// a pure function of the slabs, no float atomics), then the explicit update.
// MODE 0: store raw sums (mvx,mvy,mvz,m): the state "after ParticleToGrid" (mpm_download_array)
//         and the payload of the multi-GPU halo exchange.
// MODE 1: gather + v = mv/m, walls, analytic colliders per mpm_bc, store v and v*.
// MODE 2: like 1 but starting from the raw sums already in gv (after neighbours' sums were added).
// ---------------------------------------------------------------------------
// Analytic colliders of the grid update: a runtime table in place of the reference's compile-time
// scenes (update_grid_kernel<T, MPM_BOUNDARY_CONDITION>, cuda_mpm_kernels.cuh:660-789).  The first
// collider of the list whose region contains the node decides (the reference's `else` / `break`
// chains, :694-734, :752-774).
struct GridCollider {       // mirrors mpm_grid_collider_t (include/mpm_hip.h)
    int shape;              // 0 sphere (centre p, radius), 1 half-space (inside: n . (x - p) < 0)
    int mode;               // 0 fixed, 1 slip while approaching (scene 0), 2 slip whenever inside (scene 2)
    float p[3], n[3], radius, v[3], friction;
};
constexpr int MAX_GRID_COLLIDERS = 16;
struct GridColliders {
    int n;
    GridCollider c[MAX_GRID_COLLIDERS];
};

constexpr int GRID_LIST = 160;   // slabs over one block: 27 neighbours x splits
MPM_DEV int __reduce_max_sync_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

// halo buffer layout: [0] count, [4..) block ids (cap), then cap * 64 float4 (see the multi-GPU section below)
MPM_DEV size_t halo_ids_offset() { return 4; }                       // in uint32 units
MPM_DEV size_t halo_data_offset(unsigned cap) { return ((size_t)(4 + cap) * 4 + 15) / 16; }  // in float4 units
template <int MODE>
__global__ __launch_bounds__(256) void k_grid(DP p, GridColliders gc) {
    __shared__ int2 s_list[4][GRID_LIST];   // (item, offset index)
    const Ctl* ctl = p.ctl;
    {
        const bool out = gated_out(p);
        if (blockIdx.x == 0 && threadIdx.x == 0) p.ctl->skip_this = out;
        if (out) return;
    }
    const unsigned n_active = ctl->n_active;
    const int tid = threadIdx.x;
    const int cell = tid & 63;
    const int cx = cell >> 4, cy = (cell >> 2) & 3, cz = cell & 3;
    // the multi-GPU path packs halo buffers right after this kernel: reset their entry counters here
    if (MODE == 0 && blockIdx.x == 0 && tid < 2 && p.halo_hdr[tid]) p.halo_hdr[tid][0] = 0u;
    if (MODE == 2 && p.halo_pn > 0 && tid < 64 && blockIdx.x * 64u < p.halo_pcap) {
        // Diagnostics of the received lists (what k_halo_add2 reports for the public mpm_halo_add): the add
        // below looks a zone block up in ITS zone's buffer, so a received block that lies in this rank's grid but outside
        // the zone its buffer belongs to would be dropped without a trace.  The first few workgroups look at 64 ids each.
        for (int k = 0; k < p.halo_pn; ++k) {
            const uint32_t* buf = p.halo_pbuf[k];
            const unsigned n = min(buf[0], p.halo_pcap);
            for (unsigned e = blockIdx.x * 64u + (unsigned)tid; e < n; e += gridDim.x * 64u) {
                const uint32_t id = buf[halo_ids_offset() + e];
                if (id >= p.nblocks) {   // (no sender writes such an id: the buffer is not a halo buffer of this grid)
                    atomicOr(&p.ctl->error, ERR_HALO);
                    continue;
                }
                if (p.lut_act[id] < 0) continue;   // nothing of ours reaches that block
                int hx, hy, hz;
                block_coords(id, hx, hy, hz);
                if (hx < p.halo_plo[k] || hx > p.halo_phi[k]) atomicOr(&p.ctl->error, ERR_CAPACITY);
            }
        }
    }
    for (unsigned a = blockIdx.x * 4 + (tid >> 6); a < n_active; a += gridDim.x * 4) {
        if (MODE == 2 && p.halo_cls >= 0) {   // split update around the halo exchange (wave-uniform)
            int hx, hy, hz;
            block_coords(p.act_block[a], hx, hy, hz);
            if (!halo_block_selected(p, hx)) continue;
        }
        const int* nbr = p.act_nbr_items + (size_t)a * 27;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 2) {
            s = p.gv[(size_t)a * 64 + cell];
            if (p.halo_pn > 0) {
                // chain substep: the neighbour's sums for this block, straight from the received buffer (what k_halo_add2
                // does as a launch of its own for the public mpm_halo_add).  The buffer lists its blocks in the order its
                // sender's atomics fell: the wave scans the ids, 64 per step (a zone holds a few hundred blocks).  own +
                // received is the same pair of numbers on both ranks, so both compute identical node values.
                const uint32_t myid = p.act_block[a];
                int bx, by, bz;
                block_coords(myid, bx, by, bz);
                for (int k = 0; k < p.halo_pn; ++k) {
                    if (bx < p.halo_plo[k] || bx > p.halo_phi[k]) continue;   // wave-uniform
                    const uint32_t* buf = p.halo_pbuf[k];
                    const unsigned n = min(buf[0], p.halo_pcap);
                    int found = -1;
                    for (unsigned base = 0; base < n && found < 0; base += 64) {
                        const uint32_t id = base + (unsigned)cell < n ? buf[halo_ids_offset() + base + (unsigned)cell] : 0xFFFFFFFFu;
                        const unsigned long long m = __ballot(id == myid);
                        if (m) found = (int)base + __builtin_ctzll(m);
                    }
                    if (found >= 0) {
                        const float4 r = (reinterpret_cast<const float4*>(buf) + halo_data_offset(p.halo_pcap))[(size_t)found * 64 + cell];
                        s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
                    }
                }
            }
        }
        if (MODE != 2) {
            // Phase 1: lanes 0..26 look at one neighbour home block each and list the slabs (one per
            // work item of that block) whose stencils reached this block, in a fixed order
            // (split index, then offset).  Phase 2 streams through the list: all table walks are
            // done, the slab reads are independent and overlap.
            int2* list = s_list[tid >> 6];
            const int packed = cell < 27 ? nbr[cell] : -1;
            const int it0 = packed & 0xFFFFFF, ni = packed < 0 ? 0 : (packed >> 24);
            int cnt = 0;
            const int nimax = __reduce_max_sync_i32(ni);
            for (int k = 0; k < nimax; ++k) {
                // this block seen from the home block is at offset -o
                const bool hit = k < ni && ((p.slab_mask[it0 + k] >> (26 - cell)) & 1u);
                const unsigned long long m = __ballot(hit);
                if (hit) {
                    const int at = cnt + (int)__popcll(m & ((1ull << cell) - 1ull));
                    if (at < GRID_LIST) list[at] = make_int2(it0 + k, cell);
                }
                cnt += (int)__popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (cnt > GRID_LIST) {   // > 160 slabs over one block (dozens of split, very heavy neighbours)
                if (cell == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
                cnt = GRID_LIST;
            }
#pragma unroll 4
            for (int e = 0; e < cnt; ++e) {
                const int2 en = list[e];
                const int it = en.x, o = en.y;
                const int tx = cx - 4 * (o / 9 - 1) + FREE_ZONE;
                const int ty = cy - 4 * ((o / 3) % 3 - 1) + FREE_ZONE;
                const int tz = cz - 4 * (o % 3 - 1) + FREE_ZONE;
                // branch-free so that the unrolled loads are issued back to back: lanes outside the
                // slab read its node 0 and add nothing
                const bool in_tile = !(tx < 0 || ty < 0 || tz < 0 || tx >= TILE_W || ty >= TILE_W || tz >= TILE_W);
                const int node = in_tile ? (tx * TILE_W + ty) * TILE_W + tz : 0;
                const float4 t = p.slab[(size_t)it * TILE_N + node];
                s.x += in_tile ? t.x : 0.f; s.y += in_tile ? t.y : 0.f;
                s.z += in_tile ? t.z : 0.f; s.w += in_tile ? t.w : 0.f;
            }
            __builtin_amdgcn_wave_barrier();   // the list is reused by the wave's next block
        }
        const size_t gi = (size_t)a * 64 + cell;
        if (MODE == 0) {
            p.gv[gi] = s;
            if (p.halo_pn > 0) {   // chain substep: blocks next to a cut go into the send buffers from here
                int bx, by, bz;
                block_coords(p.act_block[a], bx, by, bz);
                for (int k = 0; k < p.halo_pn; ++k) {
                    if (bx < p.halo_plo[k] || bx > p.halo_phi[k]) continue;   // wave-uniform
                    const int nbx = bx + p.halo_pshift[k];
                    if (nbx < 0 || nbx >= p.nb) continue;
                    uint32_t* buf = p.halo_pbuf[k];
                    unsigned slot = 0;
                    if (cell == 0) slot = atomicAdd(p.halo_pcnt[k] ? p.halo_pcnt[k] : &buf[0], 1u);   // (a LOCAL word: see DP::halo_pcnt)
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (slot >= p.halo_pcap) {
                        if (cell == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
                        continue;
                    }
                    if (cell == 0) buf[halo_ids_offset() + slot] = block_id((uint32_t)nbx, (uint32_t)by, (uint32_t)bz);
                    (reinterpret_cast<float4*>(buf) + halo_data_offset(p.halo_pcap))[(size_t)slot * 64 + cell] = s;
                }
            }
            continue;
        }
        float4 vs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.w > 0.f) {
            float v[3] = {s.x / s.w, s.y / s.w, s.z / s.w};
            int bx, by, bz;
            block_coords(p.act_block[a], bx, by, bz);
            const int gx = bx * 4 + cx, gy = by * 4 + cy, gz = bz * 4 + cz;
            const int N = 1 << p.bits, wl = p.M.wall;
            if (gx < wl && v[0] < 0.f) v[0] = 0.f;
            if (gx >= N - wl && v[0] > 0.f) v[0] = 0.f;
            if (gy < wl && v[1] < 0.f) v[1] = 0.f;
            if (gy >= N - wl && v[1] > 0.f) v[1] = 0.f;
            if (gz < wl && v[2] < 0.f) v[2] = 0.f;
            if (gz >= N - wl && v[2] > 0.f) v[2] = 0.f;
            if (gc.n > 0) {
                const float pos[3] = {((float)gx + .5f) * p.dx, ((float)gy + .5f) * p.dx, ((float)gz + .5f) * p.dx};
                for (int k = 0; k < gc.n; ++k) {      // (uniform trip count: the table is a kernel argument)
                    const GridCollider& cl = gc.c[k];
                    float n[3], dist;
                    if (cl.shape == 0) {
                        const float d0 = pos[0] - cl.p[0], d1 = pos[1] - cl.p[1], d2 = pos[2] - cl.p[2];
                        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                        const float inv = 1.f / len;
                        n[0] = d0 * inv; n[1] = d1 * inv; n[2] = d2 * inv;
                        dist = len - cl.radius;
                    } else {
                        n[0] = cl.n[0]; n[1] = cl.n[1]; n[2] = cl.n[2];
                        dist = n[0] * (pos[0] - cl.p[0]) + n[1] * (pos[1] - cl.p[1]) + n[2] * (pos[2] - cl.p[2]);
                    }
                    if (!(dist < 0.f)) continue;
                    // diff_vel = v_collider - v, dotnv = n . diff_vel  (:683-687)
                    const float dv[3] = {cl.v[0] - v[0], cl.v[1] - v[1], cl.v[2] - v[2]};
                    const float dn = n[0] * dv[0] + n[1] * dv[1] + n[2] * dv[2];
                    if (cl.mode == 0) {
                        v[0] += dv[0]; v[1] += dv[1]; v[2] += dv[2];          // :778-781
                    } else if (cl.mode == 2 || dn > 0.f) {
                        // :783-786; `dotnv * (1. - SDF_FRICTION)` is a double product in the reference
                        const float fr = cl.friction;
                        const float frac = (float)((double)dn * (1.0 - (double)fr));
                        v[0] += dv[0] * fr + n[0] * frac;
                        v[1] += dv[1] * fr + n[1] * frac;
                        v[2] += dv[2] * fr + n[2] * frac;
                    }
                    break;
                }
            }
            s.x = v[0]; s.y = v[1]; s.z = v[2];
            vs = make_float4(v[0], v[1], v[2], 0.f);
        }
        p.gv[gi] = s;
        p.gvs[gi] = vs;
    }
}

// ---------------------------------------------------------------------------
// G2P
// ---------------------------------------------------------------------------
// Stage the (TILE_W)^3 node velocities around a home block into LDS.
#ifndef MPM_G2P_THREADS
#define MPM_G2P_THREADS 512
#endif
#ifndef MPM_G2P_WAVES
#define MPM_G2P_WAVES 4
#endif
#ifndef MPM_G2P_SPLIT
#define MPM_G2P_SPLIT 1   // workgroups per work item (each takes every SPLIT-th batch of the item's particles)
#endif
constexpr int G2P_THREADS = MPM_G2P_THREADS, G2P_SPLIT = MPM_G2P_SPLIT;
constexpr int LOAD_TILE_NQ = (TILE_N + G2P_THREADS - 1) / G2P_THREADS;
MPM_DEV void load_tile(const DP& p, unsigned h, float4* tile, const float4* field, int nthreads, unsigned long long* stamps = nullptr) {
    const int* nbr = p.home_nbr_act + (size_t)h * 27;
    // NQ nodes per thread (TILE_N <= NQ * nthreads), branch-free and in three sweeps so that the
    // table loads and then the node loads of all are in flight together
    constexpr int NQ = LOAD_TILE_NQ;
    int a[NQ], cell[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = min((int)threadIdx.x + q * nthreads, TILE_N - 1);
        const int tx = n / (TILE_W * TILE_W), ty = (n / TILE_W) % TILE_W, tz = n % TILE_W;
        const int qx = tx - FREE_ZONE + 4, qy = ty - FREE_ZONE + 4, qz = tz - FREE_ZONE + 4;  // >= 2
        a[q] = nbr[(qx >> 2) * 9 + (qy >> 2) * 3 + (qz >> 2)];
        cell[q] = ((qx & 3) << 4) + ((qy & 3) << 2) + (qz & 3);
    }
    if (MPM_DIAG && stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[0] = __builtin_readcyclecounter(); }
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = field[(size_t)max(a[q], 0) * 64 + cell[q]];
    if (MPM_DIAG && stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[1] = __builtin_readcyclecounter(); }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int n = (int)threadIdx.x + q * nthreads;
        if (n < TILE_N) tile[n] = a[q] >= 0 ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// returns bit 0: the advected particle no longer fits the tile (re-sort needed); bit 1: partitioned
// domain, the stencil of a ghost copy reached nodes whose sums this rank does not have in full
MPM_DEV int g2p_particle(const DP& p, const PSet& S, const float4* tile, unsigned i, float x, float y, float z,
                            float vol, int ox, int oy, int oz, float dt) {
    const Stencil st = make_stencil(p, x, y, z, ox, oy, oz);
    int bits = 0;
    if (p.dist.on && vol < 0.f) {
        const int gx = ox + st.rx;
        if (gx < p.dist.own_lo - p.dist.zone_cells || gx + 2 >= p.dist.own_hi + p.dist.zone_cells) bits = 2;
    }
    float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4* base = tile + ((st.rx * TILE_W + st.ry) * TILE_W + st.rz);
    // The stencil weight is a product w_i(x) w_j(y) w_k(z), so the 27-node sums
    //     v   = sum w v_n,      C[r][c] = sum w v_n[r] (n - fx)[c]
    // factorise: contract z along each row of 3 nodes, then y over the 3 rows of a plane, then x
    // over the 3 planes (279 multiply-adds instead of 27 x 14).  One z-row is loaded at a time:
    // keeping all 27 node loads in flight costs ~200 VGPRs.
    const float ez[3] = {st.wz[0] * (0.f - st.fx[2]), st.wz[1] * (1.f - st.fx[2]), st.wz[2] * (2.f - st.fx[2])};
    const float ey[3] = {st.wy[0] * (0.f - st.fx[1]), st.wy[1] * (1.f - st.fx[1]), st.wy[2] * (2.f - st.fx[1])};
    // Packed arithmetic (v_pk_fma_f32, two lanes of fused multiply-adds per issue slot): the z contraction
    // carries the pair (w_k, w_k (k - fz)) so that A and Az come out of one chain, and so do (B, Bz) and
    // (v, C[:,2]).  Fully unrolled: no selects for the row weights, constant LDS offsets.
    const f32x2 W0 = {st.wz[0], ez[0]}, W1 = {st.wz[1], ez[1]}, W2 = {st.wz[2], ez[2]};
    const f32x2 zero2 = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        f32x2 BB[3] = {zero2, zero2, zero2};   // (B, Bz)[r]
        float By[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float4* row = base + (a * TILE_W + b) * TILE_W;
            const float4 g0 = row[0], g1 = row[1], g2 = row[2];
            const float wb = st.wy[b], eb = ey[b];
            const f32x2 wb2 = {wb, wb};
            const float c0[3] = {g0.x, g0.y, g0.z}, c1[3] = {g1.x, g1.y, g1.z}, c2[3] = {g2.x, g2.y, g2.z};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f32x2 AA;   // (A, Az)[r]
                if (MPM_G2P_ZPAIR && r == 2) {
                    // the z components sit in the low halves of the (z, m) pairs of the three nodes: broadcast by the
                    // instruction's operand select instead of by a move each (the compiler loads 12 bytes per node and
                    // then has no pair to select from)
                    const f32x2 z0 = {g0.z, g0.w}, z1 = {g1.z, g1.w}, z2 = {g2.z, g2.w};
                    // (one block with its own wait states between the dependent packed operations: the compiler puts an
                    // s_nop between its own, and must not be relied on to know what is inside an asm statement)
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 0\n\t"
                        "v_pk_fma_f32 %0, %3, %4, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                        "v_pk_fma_f32 %0, %5, %6, %0 op_sel_hi:[1,0,1]"
                        : "=&v"(AA) : "v"(W0), "v"(z0), "v"(W1), "v"(z1), "v"(W2), "v"(z2));
                } else {
                    const f32x2 s0 = {c0[r], c0[r]}, s1 = {c1[r], c1[r]}, s2 = {c2[r], c2[r]};
                    AA = __builtin_elementwise_fma(W2, s2, __builtin_elementwise_fma(W1, s1, W0 * s0));
                }
                BB[r] = __builtin_elementwise_fma(wb2, AA, BB[r]);
                By[r] = fmaf(eb, AA.x, By[r]);
            }
        }
        // plane a is complete
        const float wa = st.wx[a];
        const float ea = wa * ((float)a - st.fx[0]);
        const f32x2 wa2 = {wa, wa};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x2 vc = {nv[r], nC[r * 3 + 2]};
            vc = __builtin_elementwise_fma(wa2, BB[r], vc);
            nv[r] = vc.x;
            nC[r * 3 + 2] = vc.y;
            nC[r * 3 + 0] = fmaf(ea, BB[r].x, nC[r * 3 + 0]);
            nC[r * 3 + 1] = fmaf(wa, By[r], nC[r * 3 + 1]);
        }
    }
    if (diag_flags(p) & 256) {  // ablation: no stores
        if (nv[0] + nC[0] + nC[4] + nC[8] == 1.2345e30f) S.q[1][i].x = nv[0];
        return 0;
    }
    const float sc = 4.f * p.dxinv;
    const float ca = (p.M.V + 1.f) * .5f, cb = (p.M.V - 1.f) * .5f;
    float Cn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        // (an explicit fused multiply-add: left to the compiler, WHICH of the two products is fused changes from build to
        // build with the scheduling of the code around it, and C with it in the last bit)
        for (int c = 0; c < 3; ++c) Cn[r * 3 + c] = fmaf(ca, sc * nC[r * 3 + c], cb * (sc * nC[c * 3 + r]));
    // four 16-byte stores per particle (in this order: it keeps the kernel at 126 VGPRs without scratch)
    const float xn = x + nv[0] * dt, yn = y + nv[1] * dt, zn = z + nv[2] * dt;
    // A face particle's position and velocity are only looked at from outside (downloads): CalcFemStateAndForce
    // replaces them by the means of the corners before anything in a substep reads them (the re-sort bins a face by
    // the same means), and takes C8 from the c8 plane.  Between the substeps of one mpm_run_substeps batch the two
    // records are therefore not written (DP::lean_g2p; the last substep of a batch and the phase-by-phase calls
    // write them): 36 instead of 68 bytes out per face.
    const bool is_face = i < (unsigned)p.Nf;
    const bool full = !(p.lean_g2p && is_face);
    // (in this order: another one costs the kernel its register budget)
    S.q[2][i] = make_float4(Cn[0], Cn[1], Cn[2], Cn[3]);
    S.q[3][i] = make_float4(Cn[4], Cn[5], Cn[6], Cn[7]);
    if (full) {
        S.q[0][i] = make_float4(xn, yn, zn, vol);
        S.q[1][i] = make_float4(nv[0], nv[1], nv[2], Cn[8]);
    }
    if (is_face) S.c8[i] = Cn[8];
    // Does the advected particle still fit this block's tile?  Vertices are tested where the next
    // P2G will find them; a face is re-centred on its corners by the FEM kernel first, which moves
    // it by O(dt * velocity spread inside the face), hence the 1/8-cell guard band.
    const float guard = .125f, top = (float)(TILE_W - 2) - guard;
    const float tx = xn * p.dxinv - .5f - (float)ox, ty = yn * p.dxinv - .5f - (float)oy,
                tz = zn * p.dxinv - .5f - (float)oz;
    return bits | (int)!(tx >= guard && tx < top && ty >= guard && ty < top && tz >= guard && tz < top);
}

__global__ __launch_bounds__(G2P_THREADS) __attribute__((amdgpu_waves_per_eu(MPM_G2P_WAVES, MPM_G2P_WAVES))) void k_g2p(DP p, float dt) {
    __shared__ float4 tile[TILE_N];
    const Ctl* ctl = p.ctl;
    if (p.gated && ctl->skip_this) {   // (not need_rebuild itself: this kernel raises it while it runs)
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&p.ctl->skipped, 1u);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.ctl->time_since_resort += dt;   // (see Ctl::quiet_time)
    const PSet& S = p.set[ctl->cur];
    const unsigned n_items = ctl->n_items;
    for (unsigned vq = blockIdx.x; vq < n_items * G2P_SPLIT; vq += gridDim.x) {
        const unsigned q = vq / G2P_SPLIT, part = vq % G2P_SPLIT;
        __syncthreads();  // everybody is done with the previous tile
        const int4 fa = p.item_flat[2 * q];
        const unsigned h = (unsigned)fa.y;
        if (p.halo_cls >= 0) {   // split gather around the halo exchange (uniform over the workgroup)
            int hx, hy, hz;
            block_coords((uint32_t)fa.z, hx, hy, hz);
            if (!halo_item_selected(p, hx)) continue;
        }
        // (bit 2: every wave of every workgroup, which perturbs the kernel; bit 12: wave 0 of eight workgroups spread over
        // the heaviest-first order, which does not)
        const bool prof = (diag_flags(p) & 4) != 0 || ((diag_flags(p) & 4096) != 0 && (blockIdx.x & 63u) == 5u && threadIdx.x < 64);
        unsigned long long t0 = 0, t1 = 0;
        if (prof) t0 = __builtin_readcyclecounter();
        const int4 rg = p.item_rng[q];   // the item's particles: face slots [x, y), vertex slots [z, w)
        // faces then vertices as one index space: a single copy of the (large) particle body
        const int nfb = rg.y - rg.x, total = nfb + (rg.w - rg.z);
        auto slot_of = [&](int u) { return (unsigned)(u < nfb ? rg.x + u : rg.z + (u - nfb)); };
        // the first positions are requested before the tile is staged, later ones one iteration
        // ahead, so the HBM latency of the particle stream hides behind LDS work
        int u = (int)(threadIdx.x + part * G2P_THREADS);
        int left = 0;
        unsigned i = slot_of(u < total ? u : 0);
        float4 pq = S.q[0][i];
        unsigned long long ts[2] = {0, 0};
        load_tile(p, h, tile, p.gv, G2P_THREADS, prof ? ts : nullptr);
        __syncthreads();
        if (prof) t1 = __builtin_readcyclecounter();
        if (prof && (threadIdx.x & 63) == 0 && (diag_flags(p) & 4096)) {
            atomicAdd(&p.dbgbuf[12], ts[0] - t0);   // item descriptor + neighbour table arrived
            atomicAdd(&p.dbgbuf[13], ts[1] - ts[0]);   // node values arrived
        }
        int bx, by, bz;
        block_coords((uint32_t)fa.z, bx, by, bz);
        const int ox = bx * 4 - FREE_ZONE, oy = by * 4 - FREE_ZONE, oz = bz * 4 - FREE_ZONE;
#pragma unroll 1
        for (; u < total; u += G2P_THREADS * G2P_SPLIT) {
            if (!MPM_G2P_PREFETCH) {
                i = slot_of(u);
                pq = S.q[0][i];
            }
            const unsigned ci = i;
            const float4 c = pq;
            const int un = u + G2P_THREADS * G2P_SPLIT;
            if (MPM_G2P_PREFETCH && un < total) {
                i = slot_of(un);
                pq = S.q[0][i];
            }
            left |= g2p_particle(p, S, tile, ci, c.x, c.y, c.z, c.w, ox, oy, oz, dt);
        }
        // a plain store: the flag only ever goes 0 -> 1 inside this kernel (same-address atomics
        // from thousands of waves would serialise at the memory side)
        if (__ballot(left & 1) && (threadIdx.x & 63) == 0) p.ctl->need_rebuild = 1;
        if (p.dist.on && __ballot(left & 2) && (threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_HALO);
        if (prof && (threadIdx.x & 63) == 0) {
            const unsigned long long t2 = __builtin_readcyclecounter();
            atomicAdd(&p.dbgbuf[8], t1 - t0);
            atomicAdd(&p.dbgbuf[9], t2 - t1);
            atomicAdd(&p.dbgbuf[10], (unsigned long long)((total + G2P_THREADS - 1) / G2P_THREADS));
            atomicAdd(&p.dbgbuf[11], 1ull);
        }
    }
}

// ---------------------------------------------------------------------------
// Multi-GPU halo: ranks tile the domain along x, each with its own engine and local grid.  After
// the local gather (k_grid<0>) a rank packs the raw node sums of its active blocks in the layers
// next to a cut, relabelled into the neighbour's block coordinates; the neighbour adds them to
// its own sums (k_halo_add) and both then run the same grid update (k_grid<2>) on shared blocks.
// Buffer: [0] count, [4..) block ids (cap), then cap * 64 float4.
// ---------------------------------------------------------------------------

// both zones / both received buffers of a chain rank in one launch each (blockIdx.y selects)
struct HaloZones {
    int lo[2], hi[2], shift[2];
    uint32_t* buf[2];
};
__global__ __launch_bounds__(256) void k_halo_pack2(DP p, HaloZones z, unsigned cap) {
    const int k = blockIdx.y;
    uint32_t* buf = z.buf[k];
    const int bx_lo = z.lo[k], bx_hi = z.hi[k], shift_bx = z.shift[k];
    const unsigned n_active = p.ctl->n_active;
    float4* data = reinterpret_cast<float4*>(buf) + halo_data_offset(cap);
    for (unsigned a = blockIdx.x * 4 + (threadIdx.x >> 6); a < n_active; a += gridDim.x * 4) {
        int bx, by, bz;
        block_coords(p.act_block[a], bx, by, bz);
        if (bx < bx_lo || bx > bx_hi) continue;   // wave-uniform
        const int nbx = bx + shift_bx;
        if (nbx < 0 || nbx >= p.nb) continue;
        unsigned slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(&buf[0], 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= cap) {
            if ((threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
            continue;
        }
        if ((threadIdx.x & 63) == 0) buf[halo_ids_offset() + slot] = block_id((uint32_t)nbx, (uint32_t)by, (uint32_t)bz);
        data[(size_t)slot * 64 + (threadIdx.x & 63)] = p.gv[(size_t)a * 64 + (threadIdx.x & 63)];
    }
}
struct HaloBufs {
    const uint32_t* buf[2];
};

// ---- DIRECT halo (peer-to-peer stores + sequence flags) ---------------------------------------------------------------
// k_grid<0> of substep s has stored the zone sums into the neighbours' receive buffers (its halo_pbuf point into peer
// memory).  This one-thread kernel, behind it on the same stream, tells the neighbours: the kernel boundary in front of
// it has made those stores complete; the system-scope fence and release store order the flag behind them for an observer
// on another device.  (Cross-device ordering cannot be observed on a box with one GPU: the protocol, not its memory
// model, is what the one-GPU tests exercise -- DESIGN.md section 5.)
// (cnt_* / hdr_*: the pack kernel counted its entries in words of THIS device's memory -- no returning atomic on peer
// memory --; the counts go into the neighbours' buffer headers here, in front of the flags)
__global__ void k_halo_signal(uint32_t* flag_a, uint32_t* flag_b, uint32_t seq, const uint32_t* cnt_a, uint32_t* hdr_a,
                              const uint32_t* cnt_b, uint32_t* hdr_b, unsigned cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (hdr_a) __hip_atomic_store(hdr_a, min(*cnt_a, cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (hdr_b) __hip_atomic_store(hdr_b, min(*cnt_b, cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    // (MI355X_MICROARCH.md, "Compiler hazard": the wait behind the write-back may be dropped when the wave's vmcnt is
    // provably empty -- the flag could then overtake the data; an inline-asm wait is invisible to that pass)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (flag_a) __hip_atomic_store(flag_a, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (flag_b) __hip_atomic_store(flag_b, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and this one, one wave, waits until both neighbours have said so for substep `seq` (sequence numbers only grow:
// signed difference), BOUNDED: after `timeout_ticks` of the 100 MHz wall clock it gives up and raises ERR_HALO, so that a
// neighbour that never arrives is an error code, not a hung device.  The kernel boundary behind it is the acquire for
// k_grid<2>, which then reads the received sums (fine-grained memory: not held in this device's L2).
__global__ void k_halo_wait(const uint32_t* flag_a, const uint32_t* flag_b, uint32_t seq, unsigned long long timeout_ticks, Ctl* ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (true) {
        const bool a = !flag_a || (int)(__hip_atomic_load(flag_a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0;
        const bool b = !flag_b || (int)(__hip_atomic_load(flag_b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0;
        if (a && b) return;
        if (wall_clock64() - t0 > timeout_ticks) {
            atomicOr(&ctl->error, ERR_HALO);
            return;
        }
        __builtin_amdgcn_s_sleep(20);
    }
}
// (a block lies in one zone only, so the two buffers of a launch touch disjoint cells)
__global__ __launch_bounds__(256) void k_halo_add2(DP p, HaloBufs b, unsigned cap) {
    const uint32_t* buf = b.buf[blockIdx.y];
    const unsigned n = min(buf[0], cap);
    const float4* data = reinterpret_cast<const float4*>(buf) + halo_data_offset(cap);
    for (unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const uint32_t id = buf[halo_ids_offset() + e];
        if (id >= p.nblocks) continue;
        const int a = p.lut_act[id];
        if (a < 0) continue;  // nothing of ours reaches that block
        if (p.halo_nz > 0) {  // with a split update only zone blocks are still raw sums
            int hx, hy, hz;
            block_coords(id, hx, hy, hz);
            if (!in_halo_zone(p, hx)) {
                if ((threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
                continue;
            }
        }
        const size_t g = (size_t)a * 64 + (threadIdx.x & 63);
        const float4 r = data[(size_t)e * 64 + (threadIdx.x & 63)];
        float4 q = p.gv[g];
        q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
        p.gv[g] = q;
    }
}

}  // namespace mpm
