// Small dense device math for the cloth-MPM kernels (gfx950).
// Everything is fully unrolled so that matrices live in VGPRs.
// The formulas restate math_tools.cuh of the reference (cited per function);
// the code is written for registers, not translated.
#pragma once
#include <hip/hip_runtime.h>

namespace mpm {

#define MPM_DEV __device__ __forceinline__

struct Material {
    float mu, lambda;        // Lame parameters (settings.h:114-115)
    float density, gamma, K, V, cF, sdf_friction, gravity, epsv;
    int gravity_axis, wall;
};

// Reciprocal, square root and reciprocal square root of the constitutive update, in three arithmetics selected by a
// template parameter FM that runs from the kernel (k_fem<FM>) down through every helper of this file:
//   FM = 0  correctly rounded division and square root (what the reference's expressions mean; an IEEE division is ten
//           vector instructions -- v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup --, sqrtf eight).  THE DEFAULT
//           since round 5: with it one substep agrees with a plain-C restatement of the reference at north_star's plain 1e-5
//           wherever float arithmetic can deliver that (tests/test_fast_math_gpu.py: both arithmetics on the same states; tests/helpers.py: NOISE_FLOOR / floor_decides); every other
//           user of these helpers (Finalize's k_init_faces, the 3x3 solve of the contact direction) always runs it.
//   FM = 1  the hardware approximations (1 ulp each) followed by ONE Newton step in fused multiply-adds: 3 - 4
//           instructions instead of 8 - 10, results within ~0.6 ulp; with 19 divisions and 10 square roots per face
//           k_fem drops from ~910 to ~630 vector instructions per face, 2.4 us per substep at 1M particles.  The
//           caller's choice: mpm_set_fast_math(h, 1) / MPM_FAST_MATH=1.
//   FM = 2  the bare approximations (measurements only: the stress -- a difference of nearly equal numbers,
//           2 mu (F - R) -- comes out 6x noisier than with correctly rounded operations, tests/test_precision_gpu.py).
template <int FM>
MPM_DEV float f_rcp(float x) {
    if (FM == 0) return 1.f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    if (FM == 2) return r;
    return fmaf(fmaf(-x, r, 1.f), r, r);                 // r + r (1 - x r)
}
template <int FM>
MPM_DEV float f_rsqrt(float x) {
    if (FM == 0) return 1.f / sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    if (FM == 2) return y;
    const float e = fmaf(-x * y, y, 1.f);                // 1 - x y^2
    return fmaf(.5f * y, e, y);                          // y + y e / 2
}
template <int FM>
MPM_DEV float f_sqrt(float x) {
    if (FM == 0) return sqrtf(x);
    if (FM == 2) return __builtin_amdgcn_sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    return x > 0.f ? fmaf(fmaf(-s, s, x), .5f * y, s) : 0.f;   // s + (x - s^2) y / 2
}

// ---- 2x2 ------------------------------------------------------------------
MPM_DEV float det2(const float* m) { return m[0] * m[3] - m[1] * m[2]; }

template <int FM = 0>
MPM_DEV void inv2(const float* m, float* o) {  // math_tools.cuh:142-149
    const float di = f_rcp<FM>(det2(m));
    o[0] = m[3] * di;
    o[1] = -m[1] * di;
    o[2] = -m[2] * di;
    o[3] = m[0] * di;
}

// Rotation factor of the polar decomposition of a 2x2 matrix and the symmetric
// factor S (math_tools.cuh:512-549).
template <int FM = 0>
MPM_DEV void polar2(const float* A, float* U, float* S) {
    U[0] = 1.f; U[1] = 0.f; U[2] = 0.f; U[3] = 1.f;
    S[0] = A[0]; S[1] = A[1]; S[2] = A[2]; S[3] = A[3];
    if (A[0] == 0.f && A[1] == 0.f && A[2] == 0.f && A[3] == 0.f) return;
    const float detA = det2(A);
    const float adet = fabsf(detA);
    float B0, B1, B2, B3;
    if (detA < 0.f) {
        B0 = A[0] - A[3]; B1 = A[1] + A[2]; B2 = A[2] + A[1]; B3 = A[3] - A[0];
    } else {
        B0 = A[0] + A[3]; B1 = A[1] - A[2]; B2 = A[2] - A[1]; B3 = A[3] + A[0];
    }
    const float k = f_rsqrt<FM>(fabsf(B0 * B3 - B1 * B2));
    U[0] = B0 * k; U[1] = B1 * k; U[2] = B2 * k; U[3] = B3 * k;
    S[0] = (A[0] * A[0] + A[2] * A[2] + adet) * k;
    S[1] = (A[0] * A[1] + A[2] * A[3]) * k;
    S[2] = S[1];
    S[3] = (A[1] * A[1] + A[3] * A[3] + adet) * k;
}

// Rotation R = U V^T of the 2x2 SVD (math_tools.cuh:551-597 feeding
// cuda_mpm_kernels.cuh:75-78).  Only U V^T is needed by the PK1 stress; the
// reference builds it through its Jacobi SVD, so the same branches are kept to
// reproduce its column/sign choices (they cancel in U V^T, but only up to
// rounding).
template <int FM = 0>
MPM_DEV void svd2_rotation(const float* A, float* R) {
    float P[4], S[4];
    polar2<FM>(A, P, S);
    float c, s, s1, s2;
    if (fabsf(S[1]) < 1e-5f) {
        c = 1.f; s = 0.f; s1 = S[0]; s2 = S[3];
    } else {
        const float tao = .5f * (S[0] - S[3]);
        const float w = f_sqrt<FM>(tao * tao + S[1] * S[1]);
        const float t = S[1] * f_rcp<FM>(tao > 0.f ? tao + w : tao - w);
        c = f_rsqrt<FM>(t * t + 1.f);
        s = -t * c;
        s1 = c * c * S[0] - 2.f * c * s * S[1] + s * s * S[3];
        s2 = s * s * S[0] + 2.f * c * s * S[1] + c * c * S[3];
    }
    float V[4];
    if (s1 < s2) {
        V[0] = -s; V[1] = c; V[2] = -c; V[3] = -s;
    } else {
        V[0] = c; V[1] = s; V[2] = -s; V[3] = c;
    }
    // U = P V ; R = U V^T
    float U[4];
    U[0] = P[0] * V[0] + P[1] * V[2];
    U[1] = P[0] * V[1] + P[1] * V[3];
    U[2] = P[2] * V[0] + P[3] * V[2];
    U[3] = P[2] * V[1] + P[3] * V[3];
    R[0] = U[0] * V[0] + U[1] * V[1];
    R[1] = U[0] * V[2] + U[1] * V[3];
    R[2] = U[2] * V[0] + U[3] * V[1];
    R[3] = U[2] * V[2] + U[3] * V[3];
}

// 2D fixed-corotated first Piola-Kirchhoff stress (cuda_mpm_kernels.cuh:72-86)
template <int FM = 0>
MPM_DEV void pk1_fixed_corotated_2d(const Material& M, const float* F, float* P) {
    float R[4], Fi[4];
    svd2_rotation<FM>(F, R);
    const float J = det2(F);
    inv2<FM>(F, Fi);
    const float a = 2.f * M.mu, b = M.lambda * (J - 1.f) * J;
    P[0] = a * (F[0] - R[0]) + b * Fi[0];
    P[1] = a * (F[1] - R[1]) + b * Fi[2];
    P[2] = a * (F[2] - R[2]) + b * Fi[1];
    P[3] = a * (F[3] - R[3]) + b * Fi[3];
}

// ---- 3x3 ------------------------------------------------------------------
MPM_DEV void mul33(const float* a, const float* b, float* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
MPM_DEV void mul33T(const float* a, const float* b, float* c) {  // c = a b^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c[i * 3 + j] = a[i * 3] * b[j * 3] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
}
MPM_DEV void mulv3(const float* a, const float* x, float* y) {
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = a[i * 3] * x[0] + a[i * 3 + 1] * x[1] + a[i * 3 + 2] * x[2];
}
MPM_DEV float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

template <int FM = 0>
MPM_DEV void inv33(const float* m, float* o) {  // math_tools.cuh:113-134
    const float det = m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) +
                      m[6] * (m[1] * m[5] - m[4] * m[2]);
    const float di = f_rcp<FM>(det);
    o[0] = (m[4] * m[8] - m[5] * m[7]) * di;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * di;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * di;
    o[1] = (m[2] * m[7] - m[1] * m[8]) * di;
    o[4] = (m[0] * m[8] - m[2] * m[6]) * di;
    o[7] = (m[1] * m[6] - m[0] * m[7]) * di;
    o[2] = (m[1] * m[5] - m[2] * m[4]) * di;
    o[5] = (m[2] * m[3] - m[0] * m[5]) * di;
    o[8] = (m[0] * m[4] - m[1] * m[3]) * di;
}

// One Givens rotation zeroing the lower entry of column `col` between rows
// (ri, rk) of R (COLS wide), applied to the rows of Qt (3 wide) as well.
template <int COLS, int FM = 0>
MPM_DEV void givens_step(float* R, float* Qt, int ri, int rk, int col) {
    const float a = R[ri * COLS + col], b = R[rk * COLS + col];
    const float q2 = a * a + b * b;
    float c = 1.f, s = 0.f;
    if (q2 > 0.f) {
        const float t = f_rsqrt<FM>(q2);
        c = a * t;
        s = -b * t;
    }
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const float t1 = R[ri * COLS + j], t2 = R[rk * COLS + j];
        R[ri * COLS + j] = c * t1 - s * t2;
        R[rk * COLS + j] = s * t1 + c * t2;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float t1 = Qt[ri * 3 + j], t2 = Qt[rk * 3 + j];
        Qt[ri * 3 + j] = c * t1 - s * t2;
        Qt[rk * 3 + j] = s * t1 + c * t2;
    }
}

// QR of a 3xCOLS matrix by Givens rotations in the elimination order of
// math_tools.cuh:456-510 (column by column, bottom row upwards).
template <int COLS, int FM = 0>
MPM_DEV void givens_qr3(const float* A, float* Q, float* R) {
    float Qt[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
#pragma unroll
    for (int i = 0; i < 3 * COLS; ++i) R[i] = A[i];
    givens_step<COLS, FM>(R, Qt, 1, 2, 0);
    givens_step<COLS, FM>(R, Qt, 0, 1, 0);
    givens_step<COLS, FM>(R, Qt, 1, 2, 1);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Q[i * 3 + j] = Qt[j * 3 + i];
}

// Return mapping of the cloth model (cuda_mpm_kernels.cuh:146-181).
template <int FM = 0>
MPM_DEV void project_strain(const Material& M, float* F) {
    float Q[9], R[9];
    givens_qr3<3, FM>(F, Q, R);
    if (M.gamma == 0.f || R[8] > 1.f) {
        R[8] = fminf(R[8], 1.f);
        R[2] = 0.f;
        R[5] = 0.f;
    } else if (R[8] <= 0.f) {
        R[2] = 0.f;
        R[5] = 0.f;
        R[8] = fmaxf(R[8], -1.f);
    } else {
        const float rr = R[2] * R[2] + R[5] * R[5];
        const float gok = M.gamma / M.K;
        const float zz = M.cF * (R[8] - 1.f) * (R[8] - 1.f);
        const float f = gok * gok * rr - zz * zz;
        if (f > 0.f) {
            const float c = zz / (gok * sqrtf(rr));
            R[2] *= c;
            R[5] *= c;
        }
    }
    mul33(Q, R, F);
}

// dPsi/dF of the anisotropic cloth energy (cuda_mpm_kernels.cuh:88-144):
// in-plane fixed corotated on R[0:2,0:2], shear gamma, normal penalty K.
template <int FM = 0>
MPM_DEV void cloth_dphi_dF(const Material& M, const float* F, float* out) {
    float Q[9], R[9];
    givens_qr3<3, FM>(F, Q, R);
    const float Rh[4] = {R[0], R[1], R[3], R[4]};
    float P2[4];
    pk1_fixed_corotated_2d<FM>(M, Rh, P2);
    // Q * [P2 0; 0 0]
    float Pp[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Pp[i * 3 + 0] = Q[i * 3] * P2[0] + Q[i * 3 + 1] * P2[2];
        Pp[i * 3 + 1] = Q[i * 3] * P2[1] + Q[i * 3 + 1] * P2[3];
        Pp[i * 3 + 2] = 0.f;
    }
    const float gp = M.gamma;
    float fp = 0.f;
    if (R[8] < 1.f) fp = -M.K * (1.f - R[8]) * (1.f - R[8]);
    float A[9];
    A[0] = gp * R[2] * R[2];
    A[1] = gp * R[2] * R[5];
    A[2] = gp * R[8] * R[2];
    A[4] = gp * R[5] * R[5];
    A[5] = gp * R[8] * R[8];
    A[8] = fp * R[8];
    A[3] = A[1];
    A[6] = A[2];
    A[7] = A[5];
    float Ri[9], QA[9], Pn[9];
    inv33<FM>(R, Ri);
    mul33(Q, A, QA);
    mul33T(QA, Ri, Pn);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] = Pp[i] + Pn[i];
}

// Right-handed orthonormal frame whose row `2` is the unit vector u
// (math_tools.cuh:599-638 with axis_index = 2): rows are (v, w, u).
__host__ __device__ inline void frame_from_normal(const float* u, float* J) {   // (host too: mpm_contact_frame, a test entry)
    // i = index of the smallest |component|; written with selects so that no
    // register array is indexed dynamically.
    const float a0 = fabsf(u[0]), a1 = fabsf(u[1]), a2 = fabsf(u[2]);
    int i = 0;
    float am = a0;
    if (a1 < am) { i = 1; am = a1; }
    if (a2 < am) { i = 2; }
    const float ui = (i == 0) ? u[0] : (i == 1 ? u[1] : u[2]);
    const float mag = sqrtf(1.f - ui * ui);
    const float r = 1.f / mag;
    const float s = -r * ui;
    if (i == 0) {         // j = 1, k = 2
        J[0] = 0.f;        J[1] = -r * u[2];  J[2] = r * u[1];
        J[3] = mag;        J[4] = s * u[1];   J[5] = s * u[2];
    } else if (i == 1) {  // j = 2, k = 0
        J[0] = r * u[2];   J[1] = 0.f;        J[2] = -r * u[0];
        J[3] = s * u[0];   J[4] = mag;        J[5] = s * u[2];
    } else {              // j = 0, k = 1
        J[0] = -r * u[1];  J[1] = r * u[0];   J[2] = 0.f;
        J[3] = s * u[0];   J[4] = s * u[1];   J[5] = mag;
    }
    J[6] = u[0]; J[7] = u[1]; J[8] = u[2];
}

// ---- contact model (cuda_mpm_kernels.cuh:956-1040, 1425-1435) --------------
struct ContactParams {
    float dt, mu, k, d, epsv;
};

MPM_DEV void contact_grad_hess(const ContactParams& c, float phi0, const float* v0, const float* v, float* H,
                               float* g) {
    const float v_hat = fminf(phi0 / c.dt, 1.f / c.d);
    if (v0[2] > v_hat) {
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = 0.f;
        g[0] = g[1] = g[2] = 0.f;
        return;
    }
    const float yn = c.k * c.dt * (phi0 - c.dt * v[2]) * (1.f - c.d * v[2]);
    const float d2n = c.k * c.dt * (-c.dt - c.d * phi0 + 2.f * c.d * c.dt * v[2]);
    const float yn0 = fmaxf(c.k * c.dt * phi0 * (1.f - c.d * v0[2]), 0.f);
    const float ts = sqrtf(v[0] * v[0] + v[1] * v[1] + c.epsv * c.epsv);
    const float t0 = v[0] / ts, t1 = v[1] / ts;
    const float co = -c.mu * yn0 / ts;
    H[0] = co * (1.f - t0 * t0); H[1] = co * (-(t0 * t1)); H[2] = 0.f;
    H[3] = co * (-(t1 * t0));    H[4] = co * (1.f - t1 * t1); H[5] = 0.f;
    H[6] = 0.f; H[7] = 0.f; H[8] = d2n;
    g[0] = -c.mu * yn0 * t0;
    g[1] = -c.mu * yn0 * t1;
    g[2] = yn;
}

MPM_DEV float contact_cost(const ContactParams& c, float phi0, const float* v0, const float* v) {
    const float v_hat = fminf(phi0 / c.dt, 1.f / c.d);
    const float yn0 = fmaxf(c.k * c.dt * phi0 * (1.f - c.d * v0[2]), 0.f);
    const float lt = c.mu * yn0 * (sqrtf(v[0] * v[0] + v[1] * v[1] + c.epsv * c.epsv) - c.epsv);
    const float vn = fminf(v_hat, v[2]);
    const float a = c.k * c.d * c.dt * c.dt;
    const float b = -(c.k * c.dt * (c.dt + c.d * phi0));
    const float cc = c.k * c.dt * phi0;
    const float ln = -((1.f / 3.f) * a * vn * vn * vn + .5f * b * vn * vn + cc * vn);
    return lt + ln;
}

// ---- index maps (cuda_mpm_kernels.cuh:296-363), bit-exact ------------------
MPM_DEV uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
MPM_DEV uint32_t compact3(uint32_t v) {
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030C30C3u;
    v = (v ^ (v >> 4)) & 0x0300F00Fu;
    v = (v ^ (v >> 8)) & 0xFF0000FFu;
    v = (v ^ (v >> 16)) & 0x000003FFu;
    return v;
}
// Morton id of a 4^3 block, x in the highest bit of each triple (:317-323).
MPM_DEV uint32_t block_id(uint32_t bx, uint32_t by, uint32_t bz) {
    return spread3(bx) * 4u + spread3(by) * 2u + spread3(bz);
}
MPM_DEV void block_coords(uint32_t id, int& bx, int& by, int& bz) {
    bx = (int)compact3(id >> 2);
    by = (int)compact3(id >> 1);
    bz = (int)compact3(id);
}
// 6-bit in-block cell code (x major) and the full cell key (:334-341).
MPM_DEV uint32_t cell_key(uint32_t x, uint32_t y, uint32_t z) {
    return (block_id(x >> 2, y >> 2, z >> 2) << 6) | ((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u);
}

// Quadratic B-spline weights for fx in [0.5, 1.5) (:470-476).
MPM_DEV void bspline3(float fx, float* w) {
    w[0] = .5f * (1.5f - fx) * (1.5f - fx);
    w[1] = .75f - (fx - 1.f) * (fx - 1.f);
    w[2] = .5f * (fx - .5f) * (fx - .5f);
}

}  // namespace mpm
