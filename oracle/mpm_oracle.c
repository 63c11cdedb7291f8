/*
 * mpm_oracle.c -- CPU restatement of the g1n0st/drake GPU cloth-MPM substep.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The
 * shipped engine (drake_amd/csrc) never links or calls anything in here.
 *
 * What it is: a plain-C (C99 + optional OpenMP) restatement of the algorithm
 * that the reference implements as CUDA kernels in
 *   multibody/gpu_mpm/cuda_mpm_kernels.cuh, math_tools.cuh,
 *   cuda_mpm_solver.cu, cuda_mpm_model.cu, radix_sort.cuh
 * Each function cites the reference file:line it follows.  Data layouts are
 * the reference's (array-of-small-vectors, dense grid indexed by cell key).
 *
 * Pinning status: the reference is CUDA-only (needs nvcc, cuda_runtime.h and
 * Eigen, none of which exist in this image) so it cannot be built here, and
 * it ships no golden vectors or numeric unit tests.  The
 * restatement is pinned by the known answers that the survey captured from
 * the reference kernels (stored in
 * tests/golden/survey_known_answers.json) and by physical invariants.
 * Anything not covered by those known answers is "parity unpinned"; where no
 * vectors exist the restatement is pinned to the PUBLISHED MODEL instead
 * (tests/test_oracle_model.py: stress = gradient of an independently written
 * energy, vertex forces = -dE/dx, APIC transfer identities).
 *
 * Determinism (round 6): every float sum has a FIXED order, whatever the thread
 * count.  The vertex forces of CalcFemStateAndForce are computed per face in
 * parallel and added in face order by one thread; ParticleToGrid has an ordered
 * variant (orc_particle_to_grid_ordered: particle order, one thread) that
 * oracle.py uses by default; the contact solve is serial.  The reference's own
 * order is that of its float atomics, i.e. undefined: any fixed order is a
 * legitimate realisation of it, and a checker that is itself a random draw
 * forces statistical tolerances on the tests.  orc_particle_to_grid
 * (OpenMP atomics: order of arrival) and the coloured scatter stay for the
 * cpu_baseline leg and for the tests that compare the variants.
 */
#include <math.h>
#include <tgmath.h>   /* sqrt, fabs, fmin ... follow the type of `real` */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* Arithmetic type of the restatement.  The reference computes in float (settings.h:37, GpuT), and so does the
 * default build (libmpm_oracle.so), which is the parity oracle.  -DORC_REAL=double (make -C oracle f64:
 * libmpm_oracle_f64.so) evaluates the same formulas in double: tests/test_precision_gpu.py uses the distance
 * between the two builds as the yardstick for what float rounding alone does to every field, phase by phase and
 * over trajectories.  Parameters and collider tables cross the boundary as float in both builds. */
#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;
#define R(x) ((real)(x))

/* Runtime version of the compile-time constants in settings.h:36-127. */
typedef struct {
    int domain_bits;    /* settings.h:49  DOMAIN_BITS            */
    int wall;           /* settings.h:56  G_BOUNDARY_CONDITION   */
    int gravity_axis;   /* settings.h:118 GRAVITY_AXIS           */
    float youngs;       /* settings.h:73  */
    float poisson;      /* settings.h:77  */
    float density;      /* settings.h:81  */
    float gamma;        /* settings.h:85  */
    float K;            /* settings.h:92  */
    float V;            /* settings.h:99  */
    float cF;           /* settings.h:103 */
    float sdf_friction; /* settings.h:110 */
    float gravity;      /* settings.h:121 */
    float epsv;         /* settings.h:125 */
} orc_params;

ORC_API void orc_default_params(orc_params *p) {
    p->domain_bits = 7;
    p->wall = 3;
    p->gravity_axis = 2;
    p->youngs = R(400000.);
    p->poisson = R(.3);
    p->density = R(2000.);
    p->gamma = R(0.);
    p->K = R(100000.);
    p->V = R(.8);
    p->cF = R(0.);
    p->sdf_friction = R(.3);
    p->gravity = -R(9.8);
    p->epsv = R(1e-3);
}

ORC_API void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* settings.h:50,57-59 */
static inline real p_dxinv(const orc_params *p) { return (real)(1 << p->domain_bits); }
static inline real p_dx(const orc_params *p) { return R(1.) / p_dxinv(p); }
static inline real p_dinv(const orc_params *p) { return R(4.) * p_dxinv(p) * p_dxinv(p); }
/* settings.h:114-115 */
static inline real p_mu(const orc_params *p) { return p->youngs / (R(2.) * (R(1.) + p->poisson)); }
static inline real p_lambda(const orc_params *p) {
    return p->youngs * p->poisson / ((R(1.) + p->poisson) * (R(1.) - R(2.) * p->poisson));
}

/* CUDA real->uint32 conversion saturates; negative inputs give 0. */
static inline uint32_t f2u(real f) { return f > R(0.) ? (uint32_t)f : 0u; }

/* ------------------------------------------------------------------ */
/* small dense helpers (math_tools.cuh:11-149), row-major              */
/* ------------------------------------------------------------------ */

/* c[n,l] = a[n,m] * b[m,l]   (math_tools.cuh:11-24) */
static void mm(int n, int m, int l, const real *a, const real *b, real *c) {
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < l; ++k) {
            real s = R(0.);
            for (int j = 0; j < m; ++j) s += a[i * m + j] * b[j * l + k];
            c[i * l + k] = s;
        }
}
/* c[n,l] = a[n,m] * b[l,m]^T (math_tools.cuh:27-39) */
static void mmT(int n, int m, int l, const real *a, const real *b, real *c) {
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < l; ++k) {
            real s = R(0.);
            for (int j = 0; j < m; ++j) s += a[i * m + j] * b[k * m + j];
            c[i * l + k] = s;
        }
}
static real dotn(int n, const real *x, const real *y) {
    real s = R(0.);
    for (int i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
}
static real norm_sqr_n(int n, const real *x) { return dotn(n, x, x); }
static real norm_n(int n, const real *x) { return sqrt(norm_sqr_n(n, x)); }

/* math_tools.cuh:113-119 */
static real det3(const real *m) {
    return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[3] * (m[1] * m[8] - m[7] * m[2]) +
           m[6] * (m[1] * m[5] - m[4] * m[2]);
}
/* math_tools.cuh:121-134; "1. / det" is a double division rounded to real */
static void inv3(const real *m, real *o) {
    real di = (real)(1.0 / (double)det3(m));
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
/* math_tools.cuh:136-149 */
static real det2(const real *m) { return m[0] * m[3] - m[1] * m[2]; }
static void inv2(const real *m, real *o) {
    real di = (real)(1.0 / (double)det2(m));
    o[0] = m[3] * di;
    o[1] = -m[1] * di;
    o[2] = -m[2] * di;
    o[3] = m[0] * di;
}

/* Givens QR of A[n,m] -> Q[n,n], R[n,m]  (math_tools.cuh:456-510) */
static void givens_qr(int n, int m, const real *A, real *Q, real *R) {
    for (int i = 0; i < n * m; ++i) R[i] = A[i];
    for (int i = 0; i < n * n; ++i) Q[i] = R(0.);
    for (int i = 0; i < n; ++i) Q[i * n + i] = R(1.);
    for (int j = 0; j < m; ++j) {
        for (int i = n - 1; i > j; --i) {
            const int ri = i - 1, rk = i;
            const real a = R[ri * m + j], b = R[rk * m + j];
            const real d = a * a + b * b;
            real c = R(1.), s = R(0.);
            const real sq = sqrt(d);
            if (sq > R(0.)) {
                const real t = (real)(1.0 / (double)sq);
                c = a * t;
                s = -b * t;
            }
            for (int jj = 0; jj < m; ++jj) {
                const real t1 = R[ri * m + jj], t2 = R[rk * m + jj];
                R[ri * m + jj] = c * t1 - s * t2;
                R[rk * m + jj] = s * t1 + c * t2;
            }
            for (int jj = 0; jj < n; ++jj) {
                const real t1 = Q[ri * n + jj], t2 = Q[rk * n + jj];
                Q[ri * n + jj] = c * t1 - s * t2;
                Q[rk * n + jj] = s * t1 + c * t2;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const real t = Q[i * n + j];
            Q[i * n + j] = Q[j * n + i];
            Q[j * n + i] = t;
        }
}

/* math_tools.cuh:512-549 */
static void polar2(const real *A, real *U, real *P) {
    U[0] = R(1.); U[1] = R(0.); U[2] = R(0.); U[3] = R(1.);
    P[0] = A[0]; P[1] = A[1]; P[2] = A[2]; P[3] = A[3];
    if (A[0] == R(0.) && A[1] == R(0.) && A[2] == R(0.) && A[3] == R(0.)) return;
    const real detA = det2(A);
    const real adetA = fabs(detA);
    real B[4] = {A[0] + A[3], A[1] - A[2], A[2] - A[1], A[3] + A[0]};
    if (detA < R(0.)) {
        B[0] = A[0] - A[3];
        B[1] = A[1] + A[2];
        B[2] = A[2] + A[1];
        B[3] = A[3] - A[0];
    }
    const real adetB = fabs(det2(B));
    const real k = R(1.) / sqrt(adetB);
    U[0] = B[0] * k; U[1] = B[1] * k; U[2] = B[2] * k; U[3] = B[3] * k;
    P[0] = (A[0] * A[0] + A[2] * A[2] + adetA) * k;
    P[1] = (A[0] * A[1] + A[2] * A[3]) * k;
    P[2] = (A[0] * A[1] + A[2] * A[3]) * k;
    P[3] = (A[1] * A[1] + A[3] * A[3] + adetA) * k;
}

/* math_tools.cuh:551-597 */
static void svd2(const real *A, real *U, real *sig, real *V) {
    real R[4], S[4];
    polar2(A, R, S);
    real c, s, s1, s2;
    if (fabs(S[1]) < R(1e-5)) {
        c = R(1.); s = R(0.); s1 = S[0]; s2 = S[3];
    } else {
        const real tao = R(.5) * (S[0] - S[3]);
        const real w = sqrt(tao * tao + S[1] * S[1]);
        const real t = (tao > R(0.)) ? S[1] / (tao + w) : S[1] / (tao - w);
        c = R(1.) / sqrt(t * t + R(1.));
        s = -t * c;
        s1 = c * c * S[0] - R(2.) * c * s * S[1] + s * s * S[3];
        s2 = s * s * S[0] + R(2.) * c * s * S[1] + c * c * S[3];
    }
    if (s1 < s2) {
        const real t = s1; s1 = s2; s2 = t;
        V[0] = -s; V[1] = c; V[2] = -c; V[3] = -s;
    } else {
        V[0] = c; V[1] = s; V[2] = -s; V[3] = c;
    }
    mm(2, 2, 2, R, V, U);
    sig[0] = s1; sig[1] = R(0.); sig[2] = R(0.); sig[3] = s2;
}

/* math_tools.cuh:599-638 (Drake's RotationMatrix::MakeFromOneUnitVector) */
static void frame_from_unit(const real u[3], int axis, real *J) {
    int i = 0;
    if (fabs(u[1]) < fabs(u[i])) i = 1;
    if (fabs(u[2]) < fabs(u[i])) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    const real uu = u[i] * u[i];
    const real mag = (real)sqrt(1.0 - (double)uu);
    const real r = R(1.) / mag;
    const real s = -r * u[i];
    J[axis * 3 + 0] = u[0]; J[axis * 3 + 1] = u[1]; J[axis * 3 + 2] = u[2];
    real v[3] = {R(0.), R(0.), R(0.)};
    v[j] = -r * u[k];
    v[k] = r * u[j];
    const int vi = (axis + 1) % 3;
    J[vi * 3 + i] = R(0.); J[vi * 3 + j] = v[j]; J[vi * 3 + k] = v[k];
    real w[3];
    w[i] = mag; w[j] = s * u[j]; w[k] = s * u[k];
    const int wi = (axis + 2) % 3;
    J[wi * 3 + 0] = w[0]; J[wi * 3 + 1] = w[1]; J[wi * 3 + 2] = w[2];
}

/* ------------------------------------------------------------------ */
/* index maps (cuda_mpm_kernels.cuh:296-363)                           */
/* ------------------------------------------------------------------ */
static inline uint32_t expand_bits(uint32_t v) {           /* :306-313 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t contract_bits(uint32_t v) {         /* :296-304 */
    v &= 0x09249249u;
    v = (v ^ (v >> 2)) & 0x030C30C3u;
    v = (v ^ (v >> 4)) & 0x0300F00Fu;
    v = (v ^ (v >> 8)) & 0xFF0000FFu;
    v = (v ^ (v >> 16)) & 0x000003FFu;
    return v;
}
ORC_API uint32_t orc_morton_code(uint32_t x, uint32_t y, uint32_t z) { /* :317-323 */
    return expand_bits(x) * 4u + expand_bits(y) * 2u + expand_bits(z);
}
ORC_API uint32_t orc_cell_index(uint32_t x, uint32_t y, uint32_t z) {  /* :334-341 */
    const uint32_t hi = orc_morton_code(x >> 2, y >> 2, z >> 2);
    const uint32_t lo = ((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u);
    return (hi << 6) | lo;
}
ORC_API void orc_inverse_cell_index(uint32_t key, uint32_t *xyz) {     /* :343-363 */
    const uint32_t hi = key >> 6, lo = key & 63u;
    xyz[0] = (contract_bits(hi >> 2) << 2) | ((lo >> 4) & 3u);
    xyz[1] = (contract_bits(hi >> 1) << 2) | ((lo >> 2) & 3u);
    xyz[2] = (contract_bits(hi) << 2) | (lo & 3u);
}

/* compute_base_cell_node_index_kernel (:365-382) */
ORC_API void orc_compute_keys(const orc_params *p, size_t n, const real *pos, uint32_t *keys,
                              uint32_t *ids) {
    const real dxinv = p_dxinv(p);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const uint32_t xi = f2u(pos[i * 3 + 0] * dxinv - R(.5));
        const uint32_t yi = f2u(pos[i * 3 + 1] * dxinv - R(.5));
        const uint32_t zi = f2u(pos[i * 3 + 2] * dxinv - R(.5));
        keys[i] = orc_cell_index(xi, yi, zi);
        if (ids) ids[i] = (uint32_t)i;
    }
}

/* radix_sort (radix_sort.cuh:202-286) as called at cuda_mpm_solver.cu:49-58:
 * a stable LSD sort of (key,id) pairs on the low `nbits` bits of the key.  The
 * reference does it 2 bits at a time; stability makes the outcome independent
 * of the digit width, so this uses 8-bit digits. Both key arrays and both id
 * arrays hold the result on exit (radix_sort.cuh:284-285). */
ORC_API void orc_sort_pairs_low_bits(size_t n, uint32_t *keys_io, uint32_t *ids_io,
                                     uint32_t *keys_out, uint32_t *ids_out, int nbits) {
    uint32_t *ka = keys_io, *ia = ids_io, *kb = keys_out, *ib = ids_out;
    for (int shift = 0; shift < nbits; shift += 8) {
        const int w = (nbits - shift) < 8 ? (nbits - shift) : 8;
        const uint32_t mask = (1u << w) - 1u;
        size_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; ++i) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; ++i) {
            const size_t dst = cnt[(ka[i] >> shift) & mask]++;
            kb[dst] = ka[i];
            ib[dst] = ia[i];
        }
        uint32_t *t = ka; ka = kb; kb = t;
        t = ia; ia = ib; ib = t;
    }
    if (ka != keys_out) {
        memcpy(keys_out, ka, n * sizeof(uint32_t));
        memcpy(ids_out, ia, n * sizeof(uint32_t));
    } else {
        memcpy(keys_io, ka, n * sizeof(uint32_t));
        memcpy(ids_io, ia, n * sizeof(uint32_t));
    }
}

/* compute_sorted_state_kernel (:384-416) */
ORC_API void orc_compute_sorted_state(size_t n, const real *pos, const real *vel,
                                      const real *vol, const real *C, const int *pids,
                                      const uint32_t *sort_ids, real *npos, real *nvel,
                                      real *nvol, real *nC, int *npids, int *index_mappings) {
    for (size_t i = 0; i < n; ++i) {
        const uint32_t s = sort_ids[i];
        nvol[i] = vol[s];
        npids[i] = pids[s];
        index_mappings[npids[i]] = (int)i;
        for (int d = 0; d < 3; ++d) {
            npos[i * 3 + d] = pos[s * 3 + d];
            nvel[i * 3 + d] = vel[s * 3 + d];
        }
        for (int d = 0; d < 9; ++d) nC[i * 9 + d] = C[s * 9 + d];
    }
}

/* ------------------------------------------------------------------ */
/* cloth constitutive model (cuda_mpm_kernels.cuh:72-181)              */
/* ------------------------------------------------------------------ */
static void fixed_corotated_pk1_2d(const orc_params *p, const real *F, real *P) { /* :72-86 */
    real U[4], sig[4], V[4], R[4], Finv[4];
    svd2(F, U, sig, V);
    mmT(2, 2, 2, U, V, R);
    const real J = det2(F);
    inv2(F, Finv);
    const real mu = p_mu(p), la = p_lambda(p);
    P[0] = R(2.) * mu * (F[0] - R[0]) + la * (J - R(1.)) * J * Finv[0];
    P[1] = R(2.) * mu * (F[1] - R[1]) + la * (J - R(1.)) * J * Finv[2];
    P[2] = R(2.) * mu * (F[2] - R[2]) + la * (J - R(1.)) * J * Finv[1];
    P[3] = R(2.) * mu * (F[3] - R[3]) + la * (J - R(1.)) * J * Finv[3];
}

static void compute_dphi_dF(const orc_params *p, const real *F, real *out) {     /* :88-144 */
    real Q[9], R[9];
    givens_qr(3, 3, F, Q, R);
    const real Rhat[4] = {R[0], R[1], R[3], R[4]};
    real P2[4];
    fixed_corotated_pk1_2d(p, Rhat, P2);
    const real Phat[9] = {P2[0], P2[1], R(0.), P2[2], P2[3], R(0.), R(0.), R(0.), R(0.)};
    real Pplane[9];
    mm(3, 3, 3, Q, Phat, Pplane);
    const real gp = p->gamma;
    real fp = R(0.);
    if (R[8] < R(1.)) fp = -p->K * (R(1.) - R[8]) * (R(1.) - R[8]);
    real A[9];
    A[0] = gp * R[2] * R[2];
    A[1] = gp * R[2] * R[5];
    A[2] = gp * R[8] * R[2];
    A[4] = gp * R[5] * R[5];
    A[5] = gp * R[8] * R[8];
    A[8] = fp * R[8];
    A[3] = A[1];
    A[6] = A[2];
    A[7] = A[5];
    real Rinv[9], QA[9], Pn[9];
    inv3(R, Rinv);
    mm(3, 3, 3, Q, A, QA);
    mmT(3, 3, 3, QA, Rinv, Pn);
    for (int i = 0; i < 9; ++i) out[i] = Pplane[i] + Pn[i];
}

static void project_strain(const orc_params *p, real *F) {                        /* :146-181 */
    real Q[9], R[9];
    givens_qr(3, 3, F, Q, R);
    if (p->gamma == R(0.)) {
        R[8] = fmin(R[8], R(1.));
        R[2] = R(0.);
        R[5] = R(0.);
    } else if (R[8] > R(1.)) {
        R[8] = fmin(R[8], R(1.));
        R[2] = R(0.);
        R[5] = R(0.);
    } else if (R[8] <= R(0.)) {
        R[2] = R(0.);
        R[5] = R(0.);
        R[8] = fmax(R[8], -R(1.));
    } else {
        const real rr = R[2] * R[2] + R[5] * R[5];
        const real gok = p->gamma / p->K;
        const real zz = p->cF * (R[8] - R(1.)) * (R[8] - R(1.));
        const real f = (gok * gok) * rr - (zz * zz);
        if (f > R(0.)) {
            const real c = zz / (gok * sqrt(rr));
            R[2] *= c;
            R[5] *= c;
        }
    }
    mm(3, 3, 3, Q, R, F);
}

/* initialize_fem_state_kernel (:13-70); vertex volumes are summed in face order */
ORC_API void orc_initialize_fem_state(const orc_params *p, size_t n_faces, const int *indices,
                                      real *pos, real *vel, real *vol, real *F, real *DmInv) {
    const real dx = p_dx(p);
    for (size_t f = 0; f < n_faces; ++f) {
        const int v0 = indices[f * 3], v1 = indices[f * 3 + 1], v2 = indices[f * 3 + 2];
        for (int i = 0; i < 3; ++i) {
            pos[f * 3 + i] = (pos[v0 * 3 + i] + pos[v1 * 3 + i] + pos[v2 * 3 + i]) / R(3.);
            vel[f * 3 + i] = (vel[v0 * 3 + i] + vel[v1 * 3 + i] + vel[v2 * 3 + i]) / R(3.);
        }
        real D0[3], D1[3];
        for (int i = 0; i < 3; ++i) {
            D0[i] = pos[v1 * 3 + i] - pos[v0 * 3 + i];
            D1[i] = pos[v2 * 3 + i] - pos[v0 * 3 + i];
        }
        const real Ds[6] = {D0[0], D1[0], D0[1], D1[1], D0[2], D1[2]};
        real Q[9], R[6];
        givens_qr(3, 2, Ds, Q, R);
        const real Dm[4] = {R[0], R[1], R(0.), R[3]};
        inv2(Dm, &DmInv[f * 4]);
        for (int i = 0; i < 9; ++i) F[f * 9 + i] = Q[i];
        const real cx = D0[1] * D1[2] - D1[1] * D0[2];
        const real cy = D0[2] * D1[0] - D1[2] * D0[0];
        const real cz = D0[0] * D1[1] - D1[0] * D0[1];
        const real c[3] = {cx, cy, cz};
        const real v4 = norm_n(3, c) / R(8.) * dx;
        vol[f] += v4;
        vol[v0] += v4;
        vol[v1] += v4;
        vol[v2] += v4;
    }
}

/* calc_fem_state_and_force_kernel (:183-294).  forces/taus must be zeroed by
 * the caller (cuda_mpm_solver.cu:74-75).  Vertex forces are summed in face
 * order when single-threaded. */
ORC_API void orc_calc_fem_state_and_force(const orc_params *p, size_t n_faces, const int *indices,
                                          const int *imap, const real *vol, const real *Caff,
                                          const real *DmInv, real *pos, real *vel, real *Fdef,
                                          real *forces, real *taus, real dt) {
    /* the nine corner-force components of every face, added to the vertices in face order below */
    real *Gall = (real *)malloc((n_faces ? n_faces : 1) * 9 * sizeof(real));
#pragma omp parallel for schedule(static)
    for (long f = 0; f < (long)n_faces; ++f) {
        const int fp = imap[f];
        const int v0 = imap[indices[f * 3]], v1 = imap[indices[f * 3 + 1]],
                  v2 = imap[indices[f * 3 + 2]];
        for (int i = 0; i < 3; ++i) {
            pos[fp * 3 + i] = (pos[v0 * 3 + i] + pos[v1 * 3 + i] + pos[v2 * 3 + i]) / R(3.);
            vel[fp * 3 + i] = (vel[v0 * 3 + i] + vel[v1 * 3 + i] + vel[v2 * 3 + i]) / R(3.);
        }
        real *F = &Fdef[f * 9];
        const real *C = &Caff[fp * 9];
        real ctF[9];
        ctF[0] = F[0];
        ctF[1] = F[1];
        ctF[2] = (R(1.) + dt * C[0]) * F[2] + dt * C[1] * F[5] + dt * C[2] * F[8];
        ctF[3] = F[3];
        ctF[4] = F[4];
        ctF[5] = dt * C[3] * F[2] + (R(1.) + dt * C[4]) * F[5] + dt * C[5] * F[8];
        ctF[6] = F[6];
        ctF[7] = F[7];
        ctF[8] = dt * C[6] * F[2] + dt * C[7] * F[5] + (R(1.) + dt * C[8]) * F[8];
        project_strain(p, ctF);

        real ds[6];
        for (int i = 0; i < 3; ++i) {
            ds[i * 2 + 0] = pos[v1 * 3 + i] - pos[v0 * 3 + i];
            ds[i * 2 + 1] = pos[v2 * 3 + i] - pos[v0 * 3 + i];
        }
        const real *Dmi = &DmInv[f * 4];
        real tF[6];
        mm(3, 2, 2, ds, Dmi, tF);
        ctF[0] = tF[0]; ctF[1] = tF[1];
        ctF[3] = tF[2]; ctF[4] = tF[3];
        ctF[6] = tF[4]; ctF[7] = tF[5];
        for (int i = 0; i < 9; ++i) F[i] = ctF[i];

        real VP[9];
        compute_dphi_dF(p, ctF, VP);
        for (int i = 0; i < 9; ++i) VP[i] *= vol[fp];

        const real a[3] = {VP[2], VP[5], VP[8]};
        const real b[3] = {ctF[2], ctF[5], ctF[8]};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) taus[fp * 9 + i * 3 + j] = a[i] * b[j];

        const real gNh[6] = {-R(1.), R(1.), R(0.), -R(1.), R(0.), R(1.)};
        const real DmiT[4] = {Dmi[0], Dmi[2], Dmi[1], Dmi[3]};
        real gN[6];
        mm(2, 2, 3, DmiT, gNh, gN);
        const real VP01[6] = {VP[0], VP[1], VP[3], VP[4], VP[6], VP[7]};
        real G[9];
        mm(3, 2, 3, VP01, gN, G);
        for (int i = 0; i < 9; ++i) Gall[f * 9 + i] = G[i];
    }
    /* the reference's 9 float atomics per face (:283-291), in face order: the same bits for any thread count */
    for (size_t f = 0; f < n_faces; ++f) {
        const int v0 = imap[indices[f * 3]], v1 = imap[indices[f * 3 + 1]],
                  v2 = imap[indices[f * 3 + 2]];
        const real *G = &Gall[f * 9];
        for (int i = 0; i < 3; ++i) {
            forces[v0 * 3 + i] += -G[i * 3 + 0];
            forces[v1 * 3 + i] += -G[i * 3 + 1];
            forces[v2 * 3 + i] += -G[i * 3 + 2];
        }
    }
    free(Gall);
}

/* ------------------------------------------------------------------ */
/* transfers                                                           */
/* ------------------------------------------------------------------ */
static inline void bspline(real fx, real *w0, real *w1, real *w2) { /* :470-476 */
    *w0 = R(.5) * (R(1.5) - fx) * (R(1.5) - fx);
    *w1 = R(.75) - (fx - R(1.)) * (fx - R(1.));
    *w2 = R(.5) * (fx - R(.5)) * (fx - R(.5));
}

/* clean_grid_kernel (:585-602) */
ORC_API void orc_clean_grid(uint32_t touched_cells, const uint32_t *ids, uint32_t *flags,
                            real *gm, real *gmv) {
    for (uint32_t i = 0; i < touched_cells; ++i) {
        const uint32_t b = ids[i >> 6];
        const uint32_t c = (b << 6) | (i & 63u);
        flags[b] = 0;
        gm[c] = R(0.);
        gmv[c * 3] = gmv[c * 3 + 1] = gmv[c * 3 + 2] = R(0.);
    }
}

/* particle_to_grid_kernel (:418-543), with every particle its own segment
 * (the warp-level run reduction of :438-457,520-528 only regroups the same
 * additions).  Grid sums are accumulated in particle order when single
 * threaded. */
static void p2g_loop(const orc_params *p, size_t n, const real *pos, const real *vel,
                     const real *vol, const real *Caff, const real *forces,
                     const real *taus, uint32_t *flags, real *gm, real *gmv,
                     real dt, int ordered) {
    const real dxinv = p_dxinv(p), dx = p_dx(p), dinv = p_dinv(p);
    const int gax = p->gravity_axis;
#pragma omp parallel for schedule(static) if (!ordered)
    for (long q = 0; q < (long)n; ++q) {
        uint32_t base[3];
        real fx[3], w[3][3];
        for (int d = 0; d < 3; ++d) {
            base[d] = f2u(pos[q * 3 + d] * dxinv - R(.5));
            fx[d] = pos[q * 3 + d] * dxinv - (real)base[d];
            bspline(fx[d], &w[0][d], &w[1][d], &w[2][d]);
        }
        const real mass = vol[q] * p->density;
        const real *v = &vel[q * 3];
        real B[9];
        for (int i = 0; i < 9; ++i) B[i] = (-dt * dinv) * taus[q * 9 + i] + Caff[q * 9 + i] * mass;
        const real *frc = &forces[q * 3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    const real xr[3] = {((real)i - fx[0]) * dx, ((real)j - fx[1]) * dx,
                                         ((real)k - fx[2]) * dx};
                    const real wt = w[i][0] * w[j][1] * w[k][2];
                    real val[4];
                    val[0] = mass * wt;
                    val[1] = v[0] * val[0];
                    val[2] = v[1] * val[0];
                    val[3] = v[2] * val[0];
                    val[gax + 1] += val[0] * p->gravity * dt;
                    val[1] += (B[0] * xr[0] + B[1] * xr[1] + B[2] * xr[2]) * wt;
                    val[2] += (B[3] * xr[0] + B[4] * xr[1] + B[5] * xr[2]) * wt;
                    val[3] += (B[6] * xr[0] + B[7] * xr[1] + B[8] * xr[2]) * wt;
                    val[1] += frc[0] * dt * wt;
                    val[2] += frc[1] * dt * wt;
                    val[3] += frc[2] * dt * wt;
                    const uint32_t c = orc_cell_index(base[0] + i, base[1] + j, base[2] + k);
                    flags[c >> 6] = 1;
#pragma omp atomic
                    gm[c] += val[0];
#pragma omp atomic
                    gmv[c * 3 + 0] += val[1];
#pragma omp atomic
                    gmv[c * 3 + 1] += val[2];
#pragma omp atomic
                    gmv[c * 3 + 2] += val[3];
                }
    }
}

/* the sums in the order of arrival of the threads' atomics (the reference: float atomics, :520-528) */
ORC_API void orc_particle_to_grid(const orc_params *p, size_t n, const real *pos, const real *vel,
                                  const real *vol, const real *Caff, const real *forces,
                                  const real *taus, uint32_t *flags, real *gm, real *gmv,
                                  real dt) {
    p2g_loop(p, n, pos, vel, vol, Caff, forces, taus, flags, gm, gmv, dt, 0);
}

/* the same sums in PARTICLE ORDER (one thread): what the parity tests compare against -- the same bits on every run */
ORC_API void orc_particle_to_grid_ordered(const orc_params *p, size_t n, const real *pos,
                                          const real *vel, const real *vol, const real *Caff,
                                          const real *forces, const real *taus, uint32_t *flags,
                                          real *gm, real *gmv, real dt) {
    p2g_loop(p, n, pos, vel, vol, Caff, forces, taus, flags, gm, gmv, dt, 1);
}


/* ------------------------------------------------------------------ */
/* CPU-baseline variant of the scatter                                 */
/* ------------------------------------------------------------------ */
/* Same arithmetic as orc_particle_to_grid, organised for many host cores: particles are
 * bucketed by the 4^3 block of their base cell and the blocks are processed in 8 colours
 * (parity of the block coordinates).  A stencil reaches at most into the next block, so two
 * blocks of one colour never write the same node and no atomics are needed.  Used only by the
 * cpu_baseline leg of bench.py (the "reference CPU path timed beside" the GPU); the parity tests
 * use the plain function above.  Sums per node are accumulated in a different order, so results
 * agree with orc_particle_to_grid to rounding, not bitwise (checked in tests/test_oracle_kat.py). */
ORC_API void orc_particle_to_grid_colored(const orc_params *p, size_t n, const real *pos,
                                          const real *vel, const real *vol, const real *Caff,
                                          const real *forces, const real *taus, uint32_t *flags,
                                          real *gm, real *gmv, real dt) {
    const real dxinv = p_dxinv(p), dx = p_dx(p), dinv = p_dinv(p);
    const int gax = p->gravity_axis;
    const uint32_t nblocks = 1u << (3 * (p->domain_bits - 2));
    uint32_t *blk = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *start = (uint32_t *)calloc(nblocks + 1, sizeof(uint32_t));
    uint32_t *order = (uint32_t *)malloc(n * sizeof(uint32_t));
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)n; ++q) {
        const uint32_t bx = f2u(pos[q * 3] * dxinv - R(.5)), by = f2u(pos[q * 3 + 1] * dxinv - R(.5)),
                       bz = f2u(pos[q * 3 + 2] * dxinv - R(.5));
        blk[q] = orc_morton_code(bx >> 2, by >> 2, bz >> 2);
    }
    for (size_t q = 0; q < n; ++q) start[blk[q] + 1]++;
    for (uint32_t b = 0; b < nblocks; ++b) start[b + 1] += start[b];
    {
        uint32_t *fill = (uint32_t *)malloc(nblocks * sizeof(uint32_t));
        memcpy(fill, start, nblocks * sizeof(uint32_t));
        for (size_t q = 0; q < n; ++q) order[fill[blk[q]]++] = (uint32_t)q;
        free(fill);
    }
    for (int colour = 0; colour < 8; ++colour) {
#pragma omp parallel for schedule(dynamic, 4)
        for (long b = 0; b < (long)nblocks; ++b) {
            if (start[b + 1] == start[b]) continue;
            /* colour = parity bits of the block coordinates; Morton code has x,y,z in bits 2,1,0 */
            if ((int)(b & 7) != colour) continue;
            for (uint32_t k = start[b]; k < start[b + 1]; ++k) {
                const size_t q = order[k];
                uint32_t base[3];
                real fx[3], w[3][3];
                for (int d = 0; d < 3; ++d) {
                    base[d] = f2u(pos[q * 3 + d] * dxinv - R(.5));
                    fx[d] = pos[q * 3 + d] * dxinv - (real)base[d];
                    bspline(fx[d], &w[0][d], &w[1][d], &w[2][d]);
                }
                const real mass = vol[q] * p->density;
                const real *v = &vel[q * 3];
                real B[9];
                for (int i = 0; i < 9; ++i) B[i] = (-dt * dinv) * taus[q * 9 + i] + Caff[q * 9 + i] * mass;
                const real *frc = &forces[q * 3];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        for (int kk = 0; kk < 3; ++kk) {
                            const real xr[3] = {((real)i - fx[0]) * dx, ((real)j - fx[1]) * dx,
                                                 ((real)kk - fx[2]) * dx};
                            const real wt = w[i][0] * w[j][1] * w[kk][2];
                            real val[4];
                            val[0] = mass * wt;
                            val[1] = v[0] * val[0];
                            val[2] = v[1] * val[0];
                            val[3] = v[2] * val[0];
                            val[gax + 1] += val[0] * p->gravity * dt;
                            val[1] += (B[0] * xr[0] + B[1] * xr[1] + B[2] * xr[2]) * wt;
                            val[2] += (B[3] * xr[0] + B[4] * xr[1] + B[5] * xr[2]) * wt;
                            val[3] += (B[6] * xr[0] + B[7] * xr[1] + B[8] * xr[2]) * wt;
                            val[1] += frc[0] * dt * wt;
                            val[2] += frc[1] * dt * wt;
                            val[3] += frc[2] * dt * wt;
                            const uint32_t c = orc_cell_index(base[0] + i, base[1] + j, base[2] + kk);
                            flags[c >> 6] = 1;
                            gm[c] += val[0];
                            gmv[c * 3 + 0] += val[1];
                            gmv[c * 3 + 1] += val[2];
                            gmv[c * 3 + 2] += val[3];
                        }
            }
        }
    }
    free(blk);
    free(start);
    free(order);
}

/* gather_touched_grid_kernel (:545-583): the reference compacts flagged blocks
 * in a nondeterministic order; the restatement emits them in ascending id. */
ORC_API uint32_t orc_gather_touched(uint32_t n_blocks, const uint32_t *flags, uint32_t *ids) {
    uint32_t cnt = 0;
    for (uint32_t b = 0; b < n_blocks; ++b)
        if (flags[b]) ids[cnt++] = b;
    return cnt;
}

static int sphere_hit(const real *pos, real cx, real cy, real cz, real r, real *nrm,
                      real *dist) {
    const real d[3] = {pos[0] - cx, pos[1] - cy, pos[2] - cz};
    const real len = sqrt(norm_sqr_n(3, d));
    *dist = len - r;
    const real inv = (real)(1.0 / (double)len); /* normalize<3> (math_tools.cuh:77-84) */
    nrm[0] = d[0] * inv; nrm[1] = d[1] * inv; nrm[2] = d[2] * inv;
    return *dist < R(0.);
}

/* update_grid_kernel<T,BC> (:632-796) */
ORC_API void orc_update_grid(const orc_params *p, int bc, uint32_t touched_cells,
                             const uint32_t *ids, const real *gm, real *gmv, real *gvs) {
    const int N = 1 << p->domain_bits, wall = p->wall;
    const real dx = p_dx(p);
#pragma omp parallel for schedule(static)
    for (long t = 0; t < (long)touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | ((uint32_t)t & 63u);
        if (!(gm[c] > R(0.))) continue;
        real *gv = &gmv[c * 3];
        gv[0] /= gm[c];
        gv[1] /= gm[c];
        gv[2] /= gm[c];
        uint32_t xyz[3];
        orc_inverse_cell_index(c, xyz);
        for (int d = 0; d < 3; ++d) {
            if ((int)xyz[d] < wall && gv[d] < R(0.)) gv[d] = R(0.);
            if ((int)xyz[d] >= N - wall && gv[d] > R(0.)) gv[d] = R(0.);
        }
        const real pos[3] = {((real)xyz[0] + R(.5)) * dx, ((real)xyz[1] + R(.5)) * dx,
                              ((real)xyz[2] + R(.5)) * dx};
        int fixed = 0, inside = 0;
        real dist = R(0.), dv[3] = {R(0.), R(0.), R(0.)}, nrm[3] = {R(0.), R(0.), R(0.)}, dotnv = R(0.);
        if (bc == 0) {                                         /* :673-692 */
            if (sphere_hit(pos, R(.5), R(.5), R(.5), R(.08), nrm, &dist)) {
                dv[0] = -gv[0]; dv[1] = -gv[1]; dv[2] = -gv[2];
                dotnv = dotn(3, nrm, dv);
                if (dotnv > R(0.) || fixed) inside = 1;
            }
        } else if (bc == 1) {                                  /* :694-734 */
            fixed = 1;
            if (sphere_hit(pos, R(.38), R(.38), R(.75), R(.04), nrm, &dist)) {
                dv[0] = -gv[0]; dv[1] = -gv[1]; dv[2] = -gv[2];
                dotnv = dotn(3, nrm, dv);
                inside = 1;
            } else if (sphere_hit(pos, R(.38), R(.62), R(.75), R(.04), nrm, &dist)) {
                dv[0] = -gv[0]; dv[1] = -gv[1]; dv[2] = -gv[2];
                dotnv = dotn(3, nrm, dv);
                inside = 1;
            }
        } else if (bc == 2) {                                  /* :737-749 */
            nrm[0] = R(0.); nrm[1] = R(0.); nrm[2] = R(1.);
            dist = pos[2] - R(.11);
            if (dist < R(0.)) {
                inside = 1;
                dv[0] = -gv[0]; dv[1] = -gv[1]; dv[2] = -gv[2];
                dotnv = dotn(3, dv, nrm);
            }
        } else if (bc == 3) {                                  /* :752-774 */
            fixed = 1;
            const real span[2] = {R(.3), R(.7)};
            for (int s = 0; s < 4; ++s) {
                if (sphere_hit(pos, span[s % 2], span[s / 2], R(.5), R(.02), nrm, &dist)) {
                    dv[0] = -gv[0]; dv[1] = -gv[1]; dv[2] = -gv[2];
                    dotnv = dotn(3, nrm, dv);
                    inside = 1;
                    break;
                }
            }
        }
        if (inside) {                                          /* :777-788 */
            if (fixed) {
                gv[0] += dv[0]; gv[1] += dv[1]; gv[2] += dv[2];
            } else {
                const real fr = p->sdf_friction;
                const real frac = (real)((double)dotnv * (1.0 - (double)fr));
                gv[0] += dv[0] * fr + nrm[0] * frac;
                gv[1] += dv[1] * fr + nrm[1] * frac;
                gv[2] += dv[2] * fr + nrm[2] * frac;
            }
        }
        gvs[c * 3 + 0] = gv[0];
        gvs[c * 3 + 1] = gv[1];
        gvs[c * 3 + 2] = gv[2];
    }
}

/* The same update with the colliders as a runtime table: checker for the
 * engine's mpm_set_grid_colliders.  Layout and semantics of mpm_grid_collider_t (include/mpm_hip.h):
 * the first collider whose region contains the node decides; mode 0 fixed (:778-781), 1 slip while
 * approaching (:684-690), 2 slip whenever inside (:737-749); response :783-786.  With the preset
 * tables it must reproduce orc_update_grid(bc) bit for bit (tests/test_oracle_kat.py). */
typedef struct {
    int32_t shape, mode;
    real p[3], n[3], radius, v[3], friction;
} orc_grid_collider;

ORC_API void orc_update_grid_table(const orc_params *p, int n_col, const orc_grid_collider *cols,
                                   uint32_t touched_cells, const uint32_t *ids, const real *gm,
                                   real *gmv, real *gvs) {
    const int N = 1 << p->domain_bits, wall = p->wall;
    const real dx = p_dx(p);
#pragma omp parallel for schedule(static)
    for (long t = 0; t < (long)touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | ((uint32_t)t & 63u);
        if (!(gm[c] > R(0.))) continue;
        real *gv = &gmv[c * 3];
        gv[0] /= gm[c];
        gv[1] /= gm[c];
        gv[2] /= gm[c];
        uint32_t xyz[3];
        orc_inverse_cell_index(c, xyz);
        for (int d = 0; d < 3; ++d) {
            if ((int)xyz[d] < wall && gv[d] < R(0.)) gv[d] = R(0.);
            if ((int)xyz[d] >= N - wall && gv[d] > R(0.)) gv[d] = R(0.);
        }
        const real pos[3] = {((real)xyz[0] + R(.5)) * dx, ((real)xyz[1] + R(.5)) * dx,
                              ((real)xyz[2] + R(.5)) * dx};
        for (int k = 0; k < n_col; ++k) {
            const orc_grid_collider *cl = &cols[k];
            real nrm[3], dist;
            if (cl->shape == 0) {
                sphere_hit(pos, cl->p[0], cl->p[1], cl->p[2], cl->radius, nrm, &dist);
            } else {
                nrm[0] = cl->n[0]; nrm[1] = cl->n[1]; nrm[2] = cl->n[2];
                dist = nrm[0] * (pos[0] - cl->p[0]) + nrm[1] * (pos[1] - cl->p[1]) +
                       nrm[2] * (pos[2] - cl->p[2]);
            }
            if (!(dist < R(0.))) continue;
            const real dv[3] = {cl->v[0] - gv[0], cl->v[1] - gv[1], cl->v[2] - gv[2]};
            const real dotnv = dotn(3, nrm, dv);
            if (cl->mode == 0) {
                gv[0] += dv[0]; gv[1] += dv[1]; gv[2] += dv[2];
            } else if (cl->mode == 2 || dotnv > R(0.)) {
                const real fr = cl->friction;
                const real frac = (real)((double)dotnv * (1.0 - (double)fr));
                gv[0] += dv[0] * fr + nrm[0] * frac;
                gv[1] += dv[1] * fr + nrm[1] * frac;
                gv[2] += dv[2] * fr + nrm[2] * frac;
            }
            break;
        }
        gvs[c * 3 + 0] = gv[0];
        gvs[c * 3 + 1] = gv[1];
        gvs[c * 3 + 2] = gv[2];
    }
}

/* grid_to_particle_kernel<T,BLOCK,CONTACT_TRANSFER> (:798-924) */
ORC_API void orc_grid_to_particle(const orc_params *p, size_t n, real *pos, real *vel,
                                  real *Caff, const real *gm, const real *gv, real dt,
                                  int contact_transfer) {
    const real dxinv = p_dxinv(p);
    const real ca = (p->V + R(1.)) * R(.5), cb = (p->V - R(1.)) * R(.5);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)n; ++q) {
        uint32_t base[3];
        real fx[3], w[3][3];
        for (int d = 0; d < 3; ++d) {
            base[d] = f2u(pos[q * 3 + d] * dxinv - R(.5));
            fx[d] = pos[q * 3 + d] * dxinv - (real)base[d];
            bspline(fx[d], &w[0][d], &w[1][d], &w[2][d]);
        }
        real nv[3] = {R(0.), R(0.), R(0.)}, nC[9] = {R(0.)};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    const real xr[3] = {(real)i - fx[0], (real)j - fx[1], (real)k - fx[2]};
                    const uint32_t c = orc_cell_index(base[0] + i, base[1] + j, base[2] + k);
                    const real *g = &gv[c * 3];
                    const real wt = w[i][0] * w[j][1] * w[k][2];
                    if (contact_transfer) {
                        if (gm[c] > R(1e-7)) {
                            nv[0] += wt * g[0];
                            nv[1] += wt * g[1];
                            nv[2] += wt * g[2];
                        }
                    } else {
                        nv[0] += wt * g[0];
                        nv[1] += wt * g[1];
                        nv[2] += wt * g[2];
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b)
                                nC[a * 3 + b] += R(4.) * dxinv * wt * g[a] * xr[b];
                    }
                }
        vel[q * 3 + 0] = nv[0];
        vel[q * 3 + 1] = nv[1];
        vel[q * 3 + 2] = nv[2];
        if (!contact_transfer) {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    Caff[q * 9 + a * 3 + b] = ca * nC[a * 3 + b] + cb * nC[b * 3 + a];
            pos[q * 3 + 0] += nv[0] * dt;
            pos[q * 3 + 1] += nv[1] * dt;
            pos[q * 3 + 2] += nv[2] * dt;
        }
    }
}

/* ------------------------------------------------------------------ */
/* contact (cuda_mpm_kernels.cuh:926-1658, cuda_mpm_solver.cu:193-621) */
/* ------------------------------------------------------------------ */

/* initialize_contact_velocities (:926-938) */
ORC_API void orc_initialize_contact_velocities(size_t nk, real *cvel, const uint32_t *cid,
                                               const real *vel) {
    for (size_t k = 0; k < nk; ++k)
        for (int i = 0; i < 3; ++i) cvel[k * 3 + i] = vel[cid[k] * 3 + i];
}

/* compute_contact_grad_and_hess (:956-1040) */
static void contact_grad_hess(const orc_params *p, real phi0, real dt, real k, real d, real mu,
                              const real *v0, const real *vn, real *H, real *g) {
    const real v_hat = fmin(phi0 / dt, R(1.) / d);
    if (v0[2] > v_hat) {
        for (int i = 0; i < 9; ++i) H[i] = R(0.);
        for (int i = 0; i < 3; ++i) g[i] = R(0.);
        return;
    }
    const real yn = k * dt * (phi0 - dt * vn[2]) * (R(1.) - d * vn[2]);
    const real d2n = k * dt * (-dt - d * phi0 + R(2.) * d * dt * vn[2]);
    const real yn0 = fmax(k * dt * phi0 * (R(1.) - d * v0[2]), R(0.));
    const real ts = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + p->epsv * p->epsv);
    const real th[2] = {vn[0] / ts, vn[1] / ts};
    const real yt[2] = {-mu * yn0 * th[0], -mu * yn0 * th[1]};
    const real Pt[4] = {th[0] * th[0], th[0] * th[1], th[1] * th[0], th[1] * th[1]};
    const real Pp[4] = {R(1.) - Pt[0], -Pt[1], -Pt[2], R(1.) - Pt[3]};
    const real co = -mu * yn0 / ts;
    H[0] = co * Pp[0]; H[1] = co * Pp[1]; H[2] = R(0.);
    H[3] = co * Pp[2]; H[4] = co * Pp[3]; H[5] = R(0.);
    H[6] = R(0.); H[7] = R(0.); H[8] = d2n;
    g[0] = yt[0]; g[1] = yt[1]; g[2] = yn;
}

/* contact cost l(v) (:1425-1435) */
static real contact_cost(const orc_params *p, real phi0, real dt, real k, real d, real mu,
                          const real *v0, const real *v) {
    const real v_hat = fmin(phi0 / dt, R(1.) / d);
    const real yn0 = fmax(k * dt * phi0 * (R(1.) - d * v0[2]), R(0.));
    const real lt = mu * yn0 * (sqrt(v[0] * v[0] + v[1] * v[1] + p->epsv * p->epsv) - p->epsv);
    const real vn = fmin(v_hat, v[2]);
    const real a = k * d * dt * dt;
    const real b = -(k * dt * (dt + d * phi0));
    const real c = k * dt * phi0;
    const real third = (real)(1. / 3.), half = (real)(1. / 2.);
    const real ln = -(third * a * vn * vn * vn + half * b * vn * vn + c * vn);
    return lt + ln;
}

typedef struct {
    real base_w[3][3];
    uint32_t base[3];
} stencil_t;

static void make_stencil(const orc_params *p, const real *x, stencil_t *s) {
    const real dxinv = p_dxinv(p);
    for (int d = 0; d < 3; ++d) {
        s->base[d] = f2u(x[d] * dxinv - R(.5));
        const real fx = x[d] * dxinv - (real)s->base[d];
        bspline(fx, &s->base_w[0][d], &s->base_w[1][d], &s->base_w[2][d]);
    }
}

/* contact_particle_to_grid_kernel<T,32,JACOBI=true> (:1042-1215) */
ORC_API void orc_contact_p2g(const orc_params *p, size_t nk, const real *cpos, const real *cvel,
                             const real *vel, const real *vol, const uint32_t *cid,
                             const real *cdist, const real *cnormal, const real *crv,
                             real *gH, real *gG, real dt, real mu, real k, real d) {
    for (size_t q = 0; q < nk; ++q) {
        stencil_t st;
        make_stencil(p, &cpos[q * 3], &st);
        const real mass = vol[cid[q]] * p->density;
        const real *pv0 = &vel[cid[q] * 3];
        const real *pv = &cvel[q * 3];
        const real nh[3] = {-cnormal[q * 3], -cnormal[q * 3 + 1], -cnormal[q * 3 + 2]};
        const real phi0 = -cdist[q];
        const real v0r[3] = {pv0[0] - crv[q * 3], pv0[1] - crv[q * 3 + 1], pv0[2] - crv[q * 3 + 2]};
        const real vr[3] = {pv[0] - crv[q * 3], pv[1] - crv[q * 3 + 1], pv[2] - crv[q * 3 + 2]};
        real RWC[9], RCW[9];
        frame_from_unit(nh, 2, RWC);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) RCW[j * 3 + i] = RWC[i * 3 + j];
        real v0[3], vn[3];
        mm(3, 3, 1, RWC, v0r, v0);
        mm(3, 3, 1, RWC, vr, vn);
        real CH[9], CG[3];
        contact_grad_hess(p, phi0, dt, k, d, mu, v0, vn, CH, CG);
        real WH[9], WG[3], tmp[9];
        mm(3, 3, 1, RCW, CG, WG);
        mm(3, 3, 3, RCW, CH, tmp);
        mm(3, 3, 3, tmp, RWC, WH);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int kk = 0; kk < 3; ++kk) {
                    const real wt = st.base_w[i][0] * st.base_w[j][1] * st.base_w[kk][2];
                    const uint32_t c =
                        orc_cell_index(st.base[0] + i, st.base[1] + j, st.base[2] + kk);
                    for (int a = 0; a < 9; ++a) gH[c * 9 + a] += mass * wt * wt * WH[a];
                    for (int a = 0; a < 3; ++a) gG[c * 3 + a] += mass * wt * WG[a];
                }
    }
}

/* clean_grid_contact_kernel (:604-630) */
ORC_API void orc_clean_grid_contact(uint32_t touched_cells, const uint32_t *ids, real *gH,
                                    real *gG, real *gD, real *galpha, real *gE0, real *gE1) {
    for (uint32_t t = 0; t < touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | (t & 63u);
        galpha[c] = -R(1.);
        for (int i = 0; i < 3; ++i) gG[c * 3 + i] = R(0.), gD[c * 3 + i] = R(0.);
        for (int i = 0; i < 9; ++i) gH[c * 9 + i] = R(0.);
        gE0[c] = R(0.);
        gE1[c] = R(0.);
    }
}

/* update_grid_contact_coordinate_descent_kernel<T,JACOBI=true> (:1217-1274).
 * Returns the two global scalars through norm_dir / dofs. */
ORC_API void orc_contact_grid_direction(uint32_t touched_cells, const uint32_t *ids,
                                        const real *gm, const real *gvs, real *gH, real *gG,
                                        const real *gv, real *gD, real *galpha, real *gE0,
                                        real *gE1, real relax, real *norm_dir, uint32_t *dofs) {
    for (uint32_t t = 0; t < touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | (t & 63u);
        if (!(gm[c] > R(0.))) continue;
        if (!((double)norm_n(9, &gH[c * 9]) > 1e-7 || (double)norm_n(3, &gG[c * 3]) > 1e-7))
            continue;
        const real m = gm[c];
        real *H = &gH[c * 9], *G = &gG[c * 3], *D = &gD[c * 3];
        H[0] -= m; H[4] -= m; H[8] -= m;
        for (int i = 0; i < 3; ++i) G[i] -= m * (gv[c * 3 + i] - gvs[c * 3 + i]);
        real Hi[9];
        inv3(H, Hi);
        mm(3, 3, 1, Hi, G, D);
        *norm_dir += norm_sqr_n(3, D);
        *dofs += 1u;
        for (int i = 0; i < 3; ++i) D[i] *= relax;
        galpha[c] = R(1.);
        gE0[c] = R(0.);
        gE1[c] = R(0.);
    }
}

/* grid_to_particle_vdb_line_search_kernel<T,32,JACOBI=true,SOLVE_DF_DDF> with
 * global_line_search=true (:1276-1473).  E0 is accumulated only if eval_E0;
 * dE/d2E only if want_derivs. */
ORC_API void orc_contact_line_search_eval(const orc_params *p, size_t nk, const real *cpos,
                                          const real *vel, const real *vol, const uint32_t *cid,
                                          const real *cdist, const real *cnormal,
                                          const real *crv, const real *gv, const real *gD,
                                          real dt, real mu, real k, real d, int eval_E0,
                                          int want_derivs, real alpha, real *E0, real *E1,
                                          real *dE1, real *d2E1) {
    for (size_t q = 0; q < nk; ++q) {
        stencil_t st;
        make_stencil(p, &cpos[q * 3], &st);
        real ov[3] = {R(0.), R(0.), R(0.)}, nv[3] = {R(0.), R(0.), R(0.)}, gd[3] = {R(0.), R(0.), R(0.)};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int kk = 0; kk < 3; ++kk) {
                    const uint32_t c =
                        orc_cell_index(st.base[0] + i, st.base[1] + j, st.base[2] + kk);
                    const real wt = st.base_w[i][0] * st.base_w[j][1] * st.base_w[kk][2];
                    for (int a = 0; a < 3; ++a) {
                        ov[a] += wt * gv[c * 3 + a];
                        nv[a] += wt * (gv[c * 3 + a] - alpha * gD[c * 3 + a]);
                        gd[a] += wt * gD[c * 3 + a];
                    }
                }
        const real mass = vol[cid[q]] * p->density;
        const real *pv0 = &vel[cid[q] * 3];
        const real nh[3] = {-cnormal[q * 3], -cnormal[q * 3 + 1], -cnormal[q * 3 + 2]};
        const real phi0 = -cdist[q];
        real v0r[3], vor[3], vnr[3];
        for (int a = 0; a < 3; ++a) {
            v0r[a] = pv0[a] - crv[q * 3 + a];
            vor[a] = ov[a] - crv[q * 3 + a];
            vnr[a] = nv[a] - crv[q * 3 + a];
        }
        real RWC[9];
        frame_from_unit(nh, 2, RWC);
        real v0[3], vol_[3], vnl[3];
        mm(3, 3, 1, RWC, v0r, v0);
        mm(3, 3, 1, RWC, vor, vol_);
        mm(3, 3, 1, RWC, vnr, vnl);
        *E1 += mass * contact_cost(p, phi0, dt, k, d, mu, v0, vnl);
        if (want_derivs) {
            real CH[9], CG[3], Rd[3], tmp[3];
            contact_grad_hess(p, phi0, dt, k, d, mu, v0, vnl, CH, CG);
            mm(3, 3, 1, RWC, gd, Rd);
            *dE1 += mass * dotn(3, CG, Rd);
            mm(1, 3, 3, Rd, CH, tmp);
            *d2E1 += mass * dotn(3, tmp, Rd);
        }
        if (eval_E0) *E0 += mass * contact_cost(p, phi0, dt, k, d, mu, v0, vol_);
    }
}

/* update_global_energy_grid_kernel<T,JACOBI=true,SOLVE_DF_DDF> (:1536-1589) */
ORC_API void orc_contact_grid_energy(uint32_t touched_cells, const uint32_t *ids, const real *gm,
                                     const real *gvs, const real *gv, const real *gD,
                                     real alpha, int eval_E0, int want_derivs, real *E0,
                                     real *E1, real *dE1, real *d2E1) {
    for (uint32_t t = 0; t < touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | (t & 63u);
        if (!(gm[c] > R(0.))) continue;
        const real m = gm[c];
        real o[3], n[3];
        for (int a = 0; a < 3; ++a) {
            o[a] = gv[c * 3 + a] - gvs[c * 3 + a];
            n[a] = gv[c * 3 + a] - alpha * gD[c * 3 + a] - gvs[c * 3 + a];
        }
        if (eval_E0) *E0 += R(.5) * m * norm_sqr_n(3, o);
        *E1 += R(.5) * m * norm_sqr_n(3, n);
        if (want_derivs) {
            *dE1 += -m * dotn(3, n, &gD[c * 3]);
            *d2E1 += m * norm_sqr_n(3, &gD[c * 3]);
        }
    }
}

/* apply_global_line_search_grid_kernel<T,JACOBI=true> (:1591-1614) */
ORC_API void orc_contact_apply_alpha(uint32_t touched_cells, const uint32_t *ids, const real *gm,
                                     real *gv, const real *gD, real alpha) {
    for (uint32_t t = 0; t < touched_cells; ++t) {
        const uint32_t c = (ids[t >> 6] << 6) | (t & 63u);
        if (!(gm[c] > R(0.))) continue;
        for (int a = 0; a < 3; ++a) gv[c * 3 + a] -= alpha * gD[c * 3 + a];
    }
}

/* apply_contact_impulse_to_rigid_bodies (:1616-1658) */
ORC_API void orc_contact_rigid_impulse(const orc_params *p, size_t nk, const real *cpos,
                                       const real *cvel0, const real *cvel, const real *vol,
                                       const uint32_t *cid, const uint32_t *crid,
                                       const real *cpwb, real *tau, real *frc) {
    for (size_t q = 0; q < nk; ++q) {
        const real m = vol[cid[q]] * p->density;
        real l[3], r[3];
        for (int a = 0; a < 3; ++a) {
            const real dv = cvel[q * 3 + a] - cvel0[q * 3 + a];
            l[a] = m * -dv;
            r[a] = cpos[q * 3 + a] - cpwb[q * 3 + a];
        }
        const real h[3] = {r[1] * l[2] - l[1] * r[2], r[2] * l[0] - l[2] * r[0],
                            r[0] * l[1] - l[0] * r[1]};
        for (int a = 0; a < 3; ++a) {
            tau[crid[q] * 3 + a] += h[a];
            frc[crid[q] * 3 + a] += l[a];
        }
    }
}

typedef struct {
    real E, dE, d2E;
} ls_eval;

typedef struct {
    const orc_params *p;
    size_t nk;
    uint32_t tc;
    const uint32_t *ids;
    const real *cpos, *vel, *vol, *cdist, *cnormal, *crv, *gm, *gvs;
    const uint32_t *cid;
    real *gv, *gD;
    real dt, mu, k, d;
} ls_ctx;

/* the `line_search` lambda of cuda_mpm_solver.cu:323-365 */
static ls_eval ls_full(const ls_ctx *c, real alpha) {
    real E1 = R(0.), dE = R(0.), d2E = R(0.), E0 = R(0.);
    orc_contact_line_search_eval(c->p, c->nk, c->cpos, c->vel, c->vol, c->cid, c->cdist,
                                 c->cnormal, c->crv, c->gv, c->gD, c->dt, c->mu, c->k, c->d, 0, 1,
                                 alpha, &E0, &E1, &dE, &d2E);
    orc_contact_grid_energy(c->tc, c->ids, c->gm, c->gvs, c->gv, c->gD, alpha, 0, 1, &E0, &E1, &dE,
                            &d2E);
    ls_eval r = {E1, dE, d2E};
    return r;
}
/* ------------------------------------------------------------------ */
/* Newton-Raphson with bisection fallback on a bracketed root.
 *
 * One body, two instantiations:
 *   - double, Drake semantics: DoNewtonWithBisectionFallback,
 *     multibody/contact_solvers/newton_with_bisection.cc:14-119 with Bracket
 *     (newton_with_bisection.h:21-60: signbit comparison).  This is what the
 *     reference's own test vectors exercise
 *     (multibody/contact_solvers/test/newton_with_bisection_test.cc:55-225).
 *   - real, the clone inside UpdateContact's exact line search
 *     (cuda_mpm_solver.cu:383-471), which differs from the original in three
 *     ways, selected by `flags`:
 *       ORC_RF_SIGN3     bracket update compares a three-way sign (:367-369,
 *                        :431) instead of signbit,
 *       ORC_RF_NO_ENDS   no early return when an end of the bracket already
 *                        satisfies f_tol (Drake :28-32),
 *       ORC_RF_STEP_LAST when |f(root)| < f_tol the clone still takes the
 *                        Newton/bisection step and returns the UPDATED root
 *                        (:437-468), the original returns the evaluated one.
 * The clone has no iteration limit; max_evals bounds it here (the caller
 * passes a large number) so a bracket of two adjacent floats cannot spin.
 * fn(ctx, x, out): out[0] = carried value (energy; unused by Drake's cases),
 * out[1] = f(x), out[2] = f'(x).
 * Returns 0 on convergence, 1 when max_evals ran out (Drake throws there). */
#define ORC_RF_SIGN3 1
#define ORC_RF_NO_ENDS 2
#define ORC_RF_STEP_LAST 4
#define ORC_DEFINE_ROOTFIND(NAME, T, FABS, HALF, TWO)                                            \
    static int NAME(void (*fn)(void *, T, T *), void *ctx, T x_lo, T f_lo, T x_hi, T f_hi,       \
                    T guess, T x_tol, T f_tol, int max_evals, int flags, T *root_out,            \
                    int *evals_out, T *last3) {                                                  \
        *evals_out = 0;                                                                          \
        if (!(flags & ORC_RF_NO_ENDS)) {                                                         \
            if (FABS(f_lo) < f_tol) { *root_out = x_lo; return 0; }                              \
            if (FABS(f_hi) < f_tol) { *root_out = x_hi; return 0; }                              \
        }                                                                                        \
        T root = guess, mdx = x_lo - x_hi, mdx_prev = mdx;                                       \
        T e[3] = {0, 0, 0};                                                                      \
        for (int n = 1; n <= max_evals; ++n) {                                                   \
            fn(ctx, root, e);                                                                    \
            *evals_out = n;                                                                      \
            if (last3) { last3[0] = e[0]; last3[1] = e[1]; last3[2] = e[2]; }                    \
            const T f = e[1], df = e[2];                                                         \
            int differ;                                                                          \
            if (flags & ORC_RF_SIGN3)                                                            \
                differ = ((f > 0) - (f < 0)) != ((f_hi > 0) - (f_hi < 0));                       \
            else                                                                                 \
                differ = (signbit(f) != 0) != (signbit(f_hi) != 0);                              \
            if (differ) { x_lo = root; f_lo = f; } else { x_hi = root; f_hi = f; }               \
            int done = 0;                                                                        \
            if (FABS(f) < f_tol) {                                                               \
                if (!(flags & ORC_RF_STEP_LAST)) { *root_out = root; return 0; }                 \
                done = 1;                                                                        \
            }                                                                                    \
            const int slow = TWO * FABS(f) > FABS(mdx_prev * df);                                \
            mdx_prev = mdx;                                                                      \
            if (slow) {                                                                          \
                mdx = HALF * (x_lo - x_hi);                                                      \
                root = x_lo - mdx;                                                               \
            } else {                                                                             \
                mdx = f / df;                                                                    \
                const T x = root - mdx;                                                          \
                if (x_lo <= x && x <= x_hi) {                                                    \
                    root = x;                                                                    \
                } else {                                                                         \
                    mdx = HALF * (x_lo - x_hi);                                                  \
                    root = x_lo - mdx;                                                           \
                }                                                                                \
            }                                                                                    \
            if (FABS(mdx) < x_tol) done = 1;                                                     \
            if (done) { *root_out = root; return 0; }                                            \
        }                                                                                        \
        *root_out = root;                                                                        \
        return 1;                                                                                \
    }
ORC_DEFINE_ROOTFIND(rootfind_f32, real, fabs, R(.5), R(2.))   /* ("f32": the reference's float clone; `real` wide here) */
ORC_DEFINE_ROOTFIND(rootfind_f64, double, fabs, .5, 2.0)

/* test entry: the double instantiation behind a C callback (f, f') = fn(x) */
typedef void (*orc_fn64)(double x, double *f, double *df);
static void rf64_thunk(void *ctx, double x, double *out) {
    out[0] = 0;
    (*(orc_fn64 *)ctx)(x, &out[1], &out[2]);
}
ORC_API int orc_newton_bisect_f64(orc_fn64 fn, double x_lo, double x_hi, double guess, double x_tol,
                                  double f_tol, int max_evals, int flags, double *root,
                                  int *evals) {
    double f_lo, f_hi, d;
    fn(x_lo, &f_lo, &d);
    fn(x_hi, &f_hi, &d);
    return rootfind_f64(rf64_thunk, &fn, x_lo, f_lo, x_hi, f_hi, guess, x_tol, f_tol, max_evals,
                        flags, root, evals, NULL);
}
/* the real instantiation with the clone's flags, same callback type (values are rounded to
 * real on the way in and out): lets the tests run the reference's cases through the very code
 * path orc_update_contact uses */
static void rf32_thunk(void *ctx, real x, real *out) {
    double f, df;
    (*(orc_fn64 *)ctx)((double)x, &f, &df);
    out[0] = R(0.); out[1] = (real)f; out[2] = (real)df;
}
ORC_API int orc_newton_bisect_f32(orc_fn64 fn, real x_lo, real x_hi, real guess, real x_tol,
                                  real f_tol, int max_evals, int flags, real *root, int *evals) {
    double f_lo, f_hi, d;
    fn((double)x_lo, &f_lo, &d);
    fn((double)x_hi, &f_hi, &d);
    return rootfind_f32(rf32_thunk, &fn, x_lo, (real)f_lo, x_hi, (real)f_hi, guess, x_tol, f_tol,
                        max_evals, flags, root, evals, NULL);
}

static void ls_thunk(void *ctx, real alpha, real *out) {
    const ls_eval r = ls_full((const ls_ctx *)ctx, alpha);
    out[0] = r.E; out[1] = r.dE; out[2] = r.d2E;
}


/* diagnostics of the last Newton iteration of the last orc_update_contact call (single-threaded
 * test use): accepted alpha, E(0), E(alpha), sum |Dir|^2, DoFs, line-search evaluations */
static real g_last_diag[6];
ORC_API void orc_last_contact_diag(real *out6) { memcpy(out6, g_last_diag, sizeof(g_last_diag)); }
/* ... and the same six numbers of EVERY Newton iteration of that call, plus the residual the loop test of :274 saw
 * after it (cuda_mpm_solver.cu:472-528, 567-570: the host-side decisions of the reference, one row per iteration):
 * alpha, E(0), E(alpha), sum |Dir|^2, DoFs, line-search evaluations, residual.  What the JSON dump of :587-612 holds
 * per iteration for the exact search, for both searches. */
#define ORC_LOG_MAX 4096
static real g_it_log[ORC_LOG_MAX][7];
static int g_it_log_n;
ORC_API int orc_contact_iteration_log(real *out, int max_rows) {
    const int n = g_it_log_n < max_rows ? g_it_log_n : max_rows;
    if (out && n > 0) memcpy(out, g_it_log, (size_t)n * 7 * sizeof(real));
    return g_it_log_n;
}

/* jacobi_relax_coeff (cuda_mpm_solver.cu:239) is a constant, 0.3, in the reference.  Tests raise it to make
 * the Newton step overshoot, so that the backtracking loop (:472-528) has to halve it several times. */
static real g_relax = R(0.3);
ORC_API void orc_set_contact_relax(real r) { g_relax = r; }

/* GpuMpmSolver::UpdateContact (cuda_mpm_solver.cu:214-621), Jacobi branch.
 * Grid arrays are dense, indexed by cell key.  On exit gv holds the post
 * contact grid velocities, cvel/cvel0 the contact velocities, tau/frc the
 * accumulated rigid impulses.  Returns the Newton iteration count. */
ORC_API int orc_update_contact(const orc_params *p, size_t nk, const real *cpos, real *cvel,
                               real *cvel0, const real *vel, const real *vol,
                               const uint32_t *cid, const uint32_t *crid, const real *cdist,
                               const real *cnormal, const real *crv, const real *cpwb,
                               uint32_t touched_blocks, const uint32_t *ids, const real *gm,
                               real *gv, const real *gvs, real *gH, real *gG, real *gD,
                               real *galpha, real *gE0, real *gE1, real *tau, real *frc,
                               real dt, real mu, real k, real d, int exact_line_search,
                               int max_iters, real *residual_out, real *ls_avg_out,
                               real *energy_out) {
    g_it_log_n = 0;
    if (!nk) return 0;                                           /* :216-217 */
    const uint32_t tc = touched_blocks * 64u;
    const real kTol = R(1e-4), relax = g_relax;                   /* :236,239 */
    if (max_iters <= 0) max_iters = 2000;                        /* :234 */
    real norm_dir = R(1e10);
    int count = 0;
    long ls_total = 0;
    real last_energy = R(0.);
    size_t n3 = nk * 3;
    real *pos_tmp = (real *)malloc(n3 * sizeof(real));
    /* pre-contact velocity at contact points (:267-272) */
    memcpy(pos_tmp, cpos, n3 * sizeof(real));
    orc_grid_to_particle(p, nk, pos_tmp, cvel0, NULL, gm, gv, dt, 1);
    ls_ctx ctx = {p, nk, tc, ids, cpos, vel, vol, cdist, cnormal, crv, gm, gvs, cid, gv, gD,
                  dt, mu, k, d};
    while (norm_dir > kTol && count < max_iters) {               /* :274 */
        real nd = R(0.), E0 = R(0.), E1 = R(0.);
        uint32_t dofs = 0;
        if (tc > 0) orc_clean_grid_contact(tc, ids, gH, gG, gD, galpha, gE0, gE1);
        orc_contact_p2g(p, nk, cpos, cvel, vel, vol, cid, cdist, cnormal, crv, gH, gG, dt, mu, k,
                        d);
        orc_contact_grid_direction(tc, ids, gm, gvs, gH, gG, gv, gD, galpha, gE0, gE1, relax, &nd,
                                   &dofs);
        int ls_cnt = 0;
        real alpha = R(1.);
        if (exact_line_search) {                                 /* :383-471 */
            const real f_tol = R(1e-8);
            const real x_tol = f_tol * relax;
            real x_lo = R(0.), x_hi = R(1.), root = R(1.);
            ls_eval f_lo = ls_full(&ctx, R(0.)), f_hi = ls_full(&ctx, R(1.));
            E0 = f_lo.E;
            if (f_lo.dE < R(0.) && f_hi.dE < R(0.)) {                /* :395-398 */
                x_lo = R(1.);
                f_lo = f_hi;
            }
            real last[3] = {R(0.), R(0.), R(0.)};
            /* guess = x_upper (:399); the clone's three deviations from Drake's routine */
            rootfind_f32(ls_thunk, &ctx, x_lo, f_lo.dE, x_hi, f_hi.dE, x_hi, x_tol, f_tol, 200,
                         ORC_RF_SIGN3 | ORC_RF_NO_ENDS | ORC_RF_STEP_LAST, &root, &ls_cnt, last);
            E1 = last[0];
            alpha = root;
        } else {                                                 /* :472-528 */
            int done = 0;
            while (!done) {
                E1 = R(0.);
                real dE = R(0.), d2E = R(0.);
                orc_contact_line_search_eval(p, nk, cpos, vel, vol, cid, cdist, cnormal, crv, gv, gD,
                                             dt, mu, k, d, ls_cnt == 0, 0, alpha, &E0, &E1, &dE,
                                             &d2E);
                orc_contact_grid_energy(tc, ids, gm, gvs, gv, gD, alpha, ls_cnt == 0, 0, &E0, &E1,
                                        &dE, &d2E);
                if (E1 <= E0) {
                    done = 1;
                } else {
                    alpha /= R(2.);
                    if (alpha < R(1e-8)) done = 1; /* "Tiny Alpha" (:523-526) */
                }
                ls_cnt += 1;
            }
        }
        ls_total += ls_cnt;
        last_energy = E1;
        g_last_diag[0] = alpha; g_last_diag[1] = E0; g_last_diag[2] = E1; g_last_diag[3] = nd;
        g_last_diag[4] = (real)dofs; g_last_diag[5] = (real)ls_cnt;
        orc_contact_apply_alpha(tc, ids, gm, gv, gD, alpha);     /* :532-539 */
        memcpy(pos_tmp, cpos, n3 * sizeof(real));
        orc_grid_to_particle(p, nk, pos_tmp, cvel, NULL, gm, gv, dt, 1); /* :557-562 */
        norm_dir = sqrt(nd) / (real)(int)dofs;                 /* :567-569 */
        if (g_it_log_n < ORC_LOG_MAX) {
            real *row = g_it_log[g_it_log_n++];
            row[0] = alpha; row[1] = E0; row[2] = E1; row[3] = nd; row[4] = (real)dofs; row[5] = (real)ls_cnt;
            row[6] = norm_dir;
        }
        count += 1;
    }
    free(pos_tmp);
    orc_contact_rigid_impulse(p, nk, cpos, cvel0, cvel, vol, cid, crid, cpwb, tau, frc); /* :615 */
    if (residual_out) *residual_out = norm_dir;
    if (ls_avg_out) *ls_avg_out = count ? (real)ls_total / (real)count : R(0.);
    if (energy_out) *energy_out = last_energy;
    return count;
}

/* ------------------------------------------------------------------ */
/* thin wrappers so the known-answer tests can reach the static helpers */
/* ------------------------------------------------------------------ */
ORC_API void orc_kat_givens_qr33(const real *A, real *Q, real *R) { givens_qr(3, 3, A, Q, R); }
ORC_API void orc_kat_givens_qr32(const real *A, real *Q, real *R) { givens_qr(3, 2, A, Q, R); }
ORC_API void orc_kat_dphi_dF(const orc_params *p, const real *F, real *P) {
    compute_dphi_dF(p, F, P);
}
ORC_API void orc_kat_project_strain(const orc_params *p, real *F) { project_strain(p, F); }
ORC_API void orc_kat_svd2(const real *A, real *U, real *s, real *V) { svd2(A, U, s, V); }
ORC_API void orc_kat_pk1_2d(const orc_params *p, const real *F, real *P) {
    fixed_corotated_pk1_2d(p, F, P);
}
ORC_API void orc_kat_frame(const real *u, int axis, real *J) { frame_from_unit(u, axis, J); }
ORC_API void orc_kat_contact_grad_hess(const orc_params *p, real phi0, real dt, real k, real d,
                                       real mu, const real *v0, const real *vn, real *H,
                                       real *g) {
    contact_grad_hess(p, phi0, dt, k, d, mu, v0, vn, H, g);
}
ORC_API real orc_kat_contact_cost(const orc_params *p, real phi0, real dt, real k, real d,
                                   real mu, const real *v0, const real *v) {
    return contact_cost(p, phi0, dt, k, d, mu, v0, v);
}
