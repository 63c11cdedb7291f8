// The four kernels of a contact-free substep.
//   k_fem     : CalcFemStateAndForce        (cuda_mpm_kernels.cuh:183-294)
//   k_vforce  : vertex forces as a gather over adjacent faces (replaces the 9
//               float atomics per face of :287-292)
//   k_p2g     : ParticleToGrid              (cuda_mpm_kernels.cuh:418-543)
//   k_grid    : touched-block sum + UpdateGrid (cuda_mpm_kernels.cuh:545-796)
//   k_g2p     : GridToParticle              (cuda_mpm_kernels.cuh:798-924)
#pragma once
#include "mpm_device.h"

namespace mpm {

// ---------------------------------------------------------------------------
// FEM: one thread per face particle (slot order => coalesced face arrays).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fem(DP p, float dt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Nf) return;
    const PSet& S = p.set[p.ctl->cur];
    const int s0 = p.fv[0][i], s1 = p.fv[1][i], s2 = p.fv[2][i];
    float x0[3], x1[3], x2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        x0[d] = S.x[d][s0];
        x1[d] = S.x[d][s1];
        x2[d] = S.x[d][s2];
        S.x[d][i] = (x0[d] + x1[d] + x2[d]) / 3.f;
        S.v[d][i] = (S.v[d][s0] + S.v[d][s1] + S.v[d][s2]) / 3.f;
    }
    float F[9], C[9], Dm[4];
#pragma unroll
    for (int d = 0; d < 9; ++d) {
        F[d] = S.F[d][i];
        C[d] = S.C[d][i];
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) Dm[d] = S.Dm[d][i];
    const float vol = S.vol[i];

    // normal column evolves with the affine velocity field (:216-226)
    float cF[9];
    cF[0] = F[0]; cF[1] = F[1];
    cF[2] = (1.f + dt * C[0]) * F[2] + dt * C[1] * F[5] + dt * C[2] * F[8];
    cF[3] = F[3]; cF[4] = F[4];
    cF[5] = dt * C[3] * F[2] + (1.f + dt * C[4]) * F[5] + dt * C[5] * F[8];
    cF[6] = F[6]; cF[7] = F[7];
    cF[8] = dt * C[6] * F[2] + dt * C[7] * F[5] + (1.f + dt * C[8]) * F[8];
    project_strain(p.M, cF);
    // in-plane columns from the deformed edges (:230-250)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float e0 = x1[d] - x0[d], e1 = x2[d] - x0[d];
        cF[d * 3 + 0] = e0 * Dm[0] + e1 * Dm[2];
        cF[d * 3 + 1] = e0 * Dm[1] + e1 * Dm[3];
    }
#pragma unroll
    for (int d = 0; d < 9; ++d) S.F[d][i] = cF[d];

    float P[9];
    cloth_dphi_dF(p.M, cF, P);
#pragma unroll
    for (int d = 0; d < 9; ++d) P[d] *= vol;
    // tau = (V P[:,2]) (x) F[:,2]  (:265-267), kept factored
    p.ab[0][i] = P[2]; p.ab[1][i] = P[5]; p.ab[2][i] = P[8];
    p.ab[3][i] = cF[2]; p.ab[4][i] = cF[5]; p.ab[5][i] = cF[8];
    // grad_N = Dm^-T [[-1,1,0],[-1,0,1]]  (:269-276)
    const float g00 = -Dm[0] - Dm[2], g01 = Dm[0], g02 = Dm[2];
    const float g10 = -Dm[1] - Dm[3], g11 = Dm[1], g12 = Dm[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float a = P[d * 3 + 0], b = P[d * 3 + 1];
        p.G[d * 3 + 0][i] = a * g00 + b * g10;
        p.G[d * 3 + 1][i] = a * g01 + b * g11;
        p.G[d * 3 + 2][i] = a * g02 + b * g12;
    }
}

// Vertex force = - sum over adjacent (face, corner) of G[:, corner], summed in
// ascending original face id (the order sequential atomics would produce).
__global__ __launch_bounds__(256) void k_vforce(DP p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.Nv) return;
    const int s = p.Nf + k;
    const PSet& S = p.set[p.ctl->cur];
    const int vo = S.pid[s] - p.Nf;
    const int e0 = p.adj_off[vo], e1 = p.adj_off[vo + 1];
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int fc = p.adj_fc[e];
        const int fs = p.imap[fc >> 2];
        const int c = fc & 3;
        f0 += -p.G[0 + c][fs];
        f1 += -p.G[3 + c][fs];
        f2 += -p.G[6 + c][fs];
    }
    p.f[0][s] = f0;
    p.f[1][s] = f1;
    p.f[2][s] = f2;
}

// ---------------------------------------------------------------------------
// P2G
// ---------------------------------------------------------------------------
template <int N>
MPM_DEV float row_shl(float v) {
    // lane i receives lane i+N of its 16-lane row; 0 past the row end
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xF, 0xF, true));
}
template <int N>
MPM_DEV int row_shr_i(int v, int fill) {
    // lane i receives lane i-N of its row; `fill` before the row start
    return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xF, 0xF, false);
}

struct Seg {
    float t1, t2, t4, t8;  // 1.0 where lane i+d continues this lane's run
    bool leader;
};

// Per-row (16 lanes) run structure of `key`: a run is a maximal sequence of
// consecutive lanes with equal key.  Wave64 adaptation of the idea at
// cuda_mpm_kernels.cuh:438-457, rebuilt for DPP row operations.
MPM_DEV Seg make_segments(int key) {
    const int lane = threadIdx.x & 63;
    const int prev = row_shr_i<1>(key, ~key);
    Seg s;
    s.leader = prev != key;
    const unsigned long long L = __ballot(s.leader);
    const int in_row = lane & 15;
    // leaders strictly above this lane, within the row
    const unsigned m = (unsigned)(((L >> lane) >> 1) & ((1u << (15 - in_row)) - 1u));
    const int interval = m ? __builtin_ctz(m) : (15 - in_row);
    s.t1 = interval >= 1 ? 1.f : 0.f;
    s.t2 = interval >= 2 ? 1.f : 0.f;
    s.t4 = interval >= 4 ? 1.f : 0.f;
    s.t8 = interval >= 8 ? 1.f : 0.f;
    return s;
}

// Sum of v over this lane's run, valid in the run's first lane.
MPM_DEV float run_sum(float v, const Seg& s) {
    v = fmaf(row_shl<1>(v), s.t1, v);
    v = fmaf(row_shl<2>(v), s.t2, v);
    v = fmaf(row_shl<4>(v), s.t4, v);
    v = fmaf(row_shl<8>(v), s.t8, v);
    return v;
}

MPM_DEV void lds_add(float* a, float v) { __hip_atomic_fetch_add(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Scatter one particle's 27 stencil contributions into the LDS tile.
// val(node) = w * (m, q + Bdx * (i,j,k)); all lanes of the wave must call this.
MPM_DEV void scatter_tile(float4* tile, bool active, int rx, int ry, int rz, const float* wx, const float* wy,
                          const float* wz, float m, const float* q, const float* Bdx) {
    const int key = active ? ((rx * 8 + ry) * 8 + rz) : (0x1000 | (int)(threadIdx.x & 63));
    const Seg seg = make_segments(key);
    const bool emit = seg.leader && active;
    float* base = reinterpret_cast<float*>(tile + ((rx * TILE_W + ry) * TILE_W + rz));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float a0 = q[0] + Bdx[0] * (float)i, a1 = q[1] + Bdx[3] * (float)i, a2 = q[2] + Bdx[6] * (float)i;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float wxy = wx[i] * wy[j];
            const float b0 = a0 + Bdx[1] * (float)j, b1 = a1 + Bdx[4] * (float)j, b2 = a2 + Bdx[7] * (float)j;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float w = wxy * wz[k];
                float v0 = w * (b0 + Bdx[2] * (float)k);
                float v1 = w * (b1 + Bdx[5] * (float)k);
                float v2 = w * (b2 + Bdx[8] * (float)k);
                float v3 = w * m;
                v0 = run_sum(v0, seg);
                v1 = run_sum(v1, seg);
                v2 = run_sum(v2, seg);
                v3 = run_sum(v3, seg);
                if (emit) {
                    float* n = base + 4 * ((i * TILE_W + j) * TILE_W + k);
                    lds_add(n + 0, v0);
                    lds_add(n + 1, v1);
                    lds_add(n + 2, v2);
                    lds_add(n + 3, v3);
                }
            }
        }
    }
}

struct Stencil {
    int rx, ry, rz;      // base cell relative to the tile origin (block origin - FREE_ZONE)
    float fx[3];
    float wx[3], wy[3], wz[3];
    unsigned mask27;     // neighbour blocks reached by the 3^3 stencil
    bool soft_out, hard_out;
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
    const int lo_s = FREE_ZONE - SOFT_ZONE, hi_s = FREE_ZONE + 3 + SOFT_ZONE;
    s.soft_out = rx < lo_s || ry < lo_s || rz < lo_s || rx > hi_s || ry > hi_s || rz > hi_s;
    const int hi_h = TILE_W - 3;
    s.hard_out = rx < 0 || ry < 0 || rz < 0 || rx > hi_h || ry > hi_h || rz > hi_h;
    rx = min(max(rx, 0), hi_h);
    ry = min(max(ry, 0), hi_h);
    rz = min(max(rz, 0), hi_h);
    s.rx = rx; s.ry = ry; s.rz = rz;
    // block offset (-1,0,1) of the first and last stencil node per axis, as bit sets
    auto bits = [](int r) -> unsigned {
        const int lo = (r - FREE_ZONE + 4) >> 2, hi2 = (r + 2 - FREE_ZONE + 4) >> 2;  // 0..2
        return (1u << lo) | (1u << hi2);
    };
    const unsigned mx = bits(rx), my = bits(ry), mz = bits(rz);
    unsigned m27 = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (((mx >> a) & 1u) && ((my >> b) & 1u)) m27 |= mz << (a * 9 + b * 3);
    s.mask27 = m27;
    return s;
}

__global__ __launch_bounds__(512) void k_p2g(DP p, float dt) {
    __shared__ float4 tile[TILE_N];
    __shared__ unsigned s_mask;
    Ctl* ctl = p.ctl;
    const PSet& S = p.set[ctl->cur];
    const unsigned n_home = ctl->n_home;
    const int tid = threadIdx.x;
    const float gdt = p.M.gravity * dt;
    const float sdt = -dt * p.Dinv;
    for (unsigned h = blockIdx.x; h < n_home; h += gridDim.x) {
        for (int n = tid; n < TILE_N; n += 512) tile[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid == 0) s_mask = 0;
        __syncthreads();
        int bx, by, bz;
        block_coords(p.home_block[h], bx, by, bz);
        const int ox = bx * 4 - FREE_ZONE, oy = by * 4 - FREE_ZONE, oz = bz * 4 - FREE_ZONE;
        const int4 rg = p.home_range[h];
        unsigned mymask = 0;
        bool soft = false, hard = false;
        // face particles: stress term, no force
        for (int base = rg.x; base < rg.y; base += 512) {
            const int i = base + tid;
            const bool act = i < rg.y;
            const int ii = act ? i : rg.y - 1;
            const Stencil st = make_stencil(p, S.x[0][ii], S.x[1][ii], S.x[2][ii], ox, oy, oz);
            const float m = S.vol[ii] * p.M.density;
            float q[3], B[9];
            const float a0 = p.ab[0][ii], a1 = p.ab[1][ii], a2 = p.ab[2][ii];
            const float b0 = p.ab[3][ii], b1 = p.ab[4][ii], b2 = p.ab[5][ii];
            const float av[3] = {a0, a1, a2}, bv[3] = {b0, b1, b2};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) B[r * 3 + c] = sdt * (av[r] * bv[c]) + S.C[r * 3 + c][ii] * m;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float qq = S.v[r][ii] * m;
                if (r == p.M.gravity_axis) qq += m * gdt;
                qq -= (B[r * 3] * st.fx[0] + B[r * 3 + 1] * st.fx[1] + B[r * 3 + 2] * st.fx[2]) * p.dx;
                q[r] = qq;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) B[r] *= p.dx;
            if (act) {
                mymask |= st.mask27;
                soft |= st.soft_out;
                hard |= st.hard_out;
            }
            scatter_tile(tile, act, st.rx, st.ry, st.rz, st.wx, st.wy, st.wz, m, q, B);
        }
        // vertex particles: force term, no stress
        for (int base = rg.z; base < rg.w; base += 512) {
            const int i = base + tid;
            const bool act = i < rg.w;
            const int ii = act ? i : rg.w - 1;
            const Stencil st = make_stencil(p, S.x[0][ii], S.x[1][ii], S.x[2][ii], ox, oy, oz);
            const float m = S.vol[ii] * p.M.density;
            float q[3], B[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) B[r] = S.C[r][ii] * m;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float qq = S.v[r][ii] * m;
                if (r == p.M.gravity_axis) qq += m * gdt;
                qq += p.f[r][ii] * dt;
                qq -= (B[r * 3] * st.fx[0] + B[r * 3 + 1] * st.fx[1] + B[r * 3 + 2] * st.fx[2]) * p.dx;
                q[r] = qq;
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) B[r] *= p.dx;
            if (act) {
                mymask |= st.mask27;
                soft |= st.soft_out;
                hard |= st.hard_out;
            }
            scatter_tile(tile, act, st.rx, st.ry, st.rz, st.wx, st.wy, st.wz, m, q, B);
        }
        if (mymask) atomicOr(&s_mask, mymask);
        if (__ballot(soft) && (tid & 63) == 0) atomicOr(&ctl->need_rebuild, 1);
        if (__ballot(hard) && (tid & 63) == 0) atomicOr(&ctl->error, ERR_DRIFT);
        __syncthreads();
        float4* out = p.slab + (size_t)h * TILE_N;
        for (int n = tid; n < TILE_N; n += 512) out[n] = tile[n];
        if (tid == 0) p.slab_mask[h] = s_mask;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Grid: per active block, sum the overlapping tiles (fixed order => the grid is
// a pure function of the slabs, no float atomics), then the explicit update.
// MODE 0: store raw sums (mvx,mvy,mvz,m) -- only used to expose the state
//         "after ParticleToGrid" to mpm_download_array.
// MODE 1: v = mv/m, walls, analytic colliders per mpm_bc, store v and v*.
// ---------------------------------------------------------------------------
MPM_DEV bool sphere_sdf(const float* pos, float cx, float cy, float cz, float r, float* n) {
    const float d0 = pos[0] - cx, d1 = pos[1] - cy, d2 = pos[2] - cz;
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float inv = 1.f / len;
    n[0] = d0 * inv; n[1] = d1 * inv; n[2] = d2 * inv;
    return len - r < 0.f;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_grid(DP p, int bc) {
    const Ctl* ctl = p.ctl;
    const unsigned n_active = ctl->n_active;
    const int tid = threadIdx.x;
    const int cell = tid & 63;
    const int cx = cell >> 4, cy = (cell >> 2) & 3, cz = cell & 3;
    for (unsigned a = blockIdx.x * 4 + (tid >> 6); a < n_active; a += gridDim.x * 4) {
        const int* nbr = p.act_nbr_home + (size_t)a * 27;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < 27; ++o) {
            const int h = nbr[o];   // wave-uniform
            if (h < 0) continue;
            // this block seen from the home block is at offset -o
            if (!((p.slab_mask[h] >> (26 - o)) & 1u)) continue;
            const int tx = cx - 4 * (o / 9 - 1) + FREE_ZONE;
            const int ty = cy - 4 * ((o / 3) % 3 - 1) + FREE_ZONE;
            const int tz = cz - 4 * (o % 3 - 1) + FREE_ZONE;
            if (tx < 0 || ty < 0 || tz < 0 || tx >= TILE_W || ty >= TILE_W || tz >= TILE_W) continue;
            const float4 t = p.slab[(size_t)h * TILE_N + (tx * TILE_W + ty) * TILE_W + tz];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const size_t gi = (size_t)a * 64 + cell;
        if (MODE == 0) {
            p.gv[gi] = s;
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
            if (bc >= 0) {
                const float pos[3] = {((float)gx + .5f) * p.dx, ((float)gy + .5f) * p.dx, ((float)gz + .5f) * p.dx};
                bool fixed = false, inside = false;
                float n[3] = {0.f, 0.f, 0.f};
                if (bc == 0) {
                    if (sphere_sdf(pos, .5f, .5f, .5f, .08f, n)) {
                        const float dn = -(n[0] * v[0] + n[1] * v[1] + n[2] * v[2]);
                        inside = dn > 0.f;
                    }
                } else if (bc == 1) {
                    fixed = true;
                    inside = sphere_sdf(pos, .38f, .38f, .75f, .04f, n) || sphere_sdf(pos, .38f, .62f, .75f, .04f, n);
                } else if (bc == 2) {
                    n[2] = 1.f;
                    inside = pos[2] - .11f < 0.f;
                } else if (bc == 3) {
                    fixed = true;
                    inside = sphere_sdf(pos, .3f, .3f, .5f, .02f, n) || sphere_sdf(pos, .7f, .3f, .5f, .02f, n) ||
                             sphere_sdf(pos, .3f, .7f, .5f, .02f, n) || sphere_sdf(pos, .7f, .7f, .5f, .02f, n);
                }
                if (inside) {
                    if (fixed) {
                        v[0] = v[1] = v[2] = 0.f;  // v += (v_collider - v), collider at rest
                    } else {
                        const float dv[3] = {-v[0], -v[1], -v[2]};
                        const float dn = n[0] * dv[0] + n[1] * dv[1] + n[2] * dv[2];
                        const float fr = p.M.sdf_friction;
                        const float frac = dn * (1.f - fr);
                        v[0] += dv[0] * fr + n[0] * frac;
                        v[1] += dv[1] * fr + n[1] * frac;
                        v[2] += dv[2] * fr + n[2] * frac;
                    }
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
MPM_DEV void load_tile(const DP& p, unsigned h, float4* tile, const float4* field, int nthreads) {
    const int* nbr = p.home_nbr_act + (size_t)h * 27;
    for (int n = threadIdx.x; n < TILE_N; n += nthreads) {
        const int tx = n / (TILE_W * TILE_W), ty = (n / TILE_W) % TILE_W, tz = n % TILE_W;
        const int qx = tx - FREE_ZONE + 4, qy = ty - FREE_ZONE + 4, qz = tz - FREE_ZONE + 4;  // >= 2
        const int o = (qx >> 2) * 9 + (qy >> 2) * 3 + (qz >> 2);
        const int a = nbr[o];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a >= 0) v = field[(size_t)a * 64 + ((qx & 3) << 4) + ((qy & 3) << 2) + (qz & 3)];
        tile[n] = v;
    }
}

MPM_DEV void g2p_particle(const DP& p, const PSet& S, const float4* tile, int i, int ox, int oy, int oz, float dt) {
    const float x = S.x[0][i], y = S.x[1][i], z = S.x[2][i];
    const Stencil st = make_stencil(p, x, y, z, ox, oy, oz);
    float nv[3] = {0.f, 0.f, 0.f}, nC[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4* base = tile + ((st.rx * TILE_W + st.ry) * TILE_W + st.rz);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 g = base[(a * TILE_W + b) * TILE_W + c];
                const float w = st.wx[a] * st.wy[b] * st.wz[c];
                const float d0 = (float)a - st.fx[0], d1 = (float)b - st.fx[1], d2 = (float)c - st.fx[2];
                const float wv0 = w * g.x, wv1 = w * g.y, wv2 = w * g.z;
                nv[0] += wv0; nv[1] += wv1; nv[2] += wv2;
                nC[0] += wv0 * d0; nC[1] += wv0 * d1; nC[2] += wv0 * d2;
                nC[3] += wv1 * d0; nC[4] += wv1 * d1; nC[5] += wv1 * d2;
                nC[6] += wv2 * d0; nC[7] += wv2 * d1; nC[8] += wv2 * d2;
            }
    const float sc = 4.f * p.dxinv;
    const float ca = (p.M.V + 1.f) * .5f, cb = (p.M.V - 1.f) * .5f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) S.C[r * 3 + c][i] = ca * (sc * nC[r * 3 + c]) + cb * (sc * nC[c * 3 + r]);
    S.v[0][i] = nv[0]; S.v[1][i] = nv[1]; S.v[2][i] = nv[2];
    S.x[0][i] = x + nv[0] * dt;
    S.x[1][i] = y + nv[1] * dt;
    S.x[2][i] = z + nv[2] * dt;
}

__global__ __launch_bounds__(512) void k_g2p(DP p, float dt) {
    __shared__ float4 tile[TILE_N];
    const Ctl* ctl = p.ctl;
    const PSet& S = p.set[ctl->cur];
    const unsigned n_home = ctl->n_home;
    for (unsigned h = blockIdx.x; h < n_home; h += gridDim.x) {
        __syncthreads();
        load_tile(p, h, tile, p.gv, 512);
        __syncthreads();
        int bx, by, bz;
        block_coords(p.home_block[h], bx, by, bz);
        const int ox = bx * 4 - FREE_ZONE, oy = by * 4 - FREE_ZONE, oz = bz * 4 - FREE_ZONE;
        const int4 rg = p.home_range[h];
        for (int i = rg.x + (int)threadIdx.x; i < rg.y; i += 512) g2p_particle(p, S, tile, i, ox, oy, oz, dt);
        for (int i = rg.z + (int)threadIdx.x; i < rg.w; i += 512) g2p_particle(p, S, tile, i, ox, oy, oz, dt);
    }
}

}  // namespace mpm
