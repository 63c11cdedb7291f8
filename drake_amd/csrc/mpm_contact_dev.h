// Device buffers and kernels of the grid-space contact solve (UpdateContact,
// cuda_mpm_solver.cu:214-621): Jacobi-Newton on the grid velocities of the
// SAP-style cost  E(v) = sum_nodes 1/2 m |v - v*|^2 + sum_contacts m_p l(v_p).
//
// Differences in organisation from the reference (same arithmetic):
//   - the contact -> node scatter of Hessians/gradients (12 float atomics x 27
//     nodes per contact per iteration, cuda_mpm_kernels.cuh:1184-1212) is a
//     gather: contact positions are fixed during the solve, so a node -> contacts
//     adjacency (CSR) is built once per UpdateContact and every Newton iteration
//     sums each node's list in a fixed order;
//   - the backtracking line search evaluates all 28 step lengths 1, 1/2, ...,
//     2^-27 in one pass and picks the first acceptable one on the device, so an
//     iteration needs no host round trip (the reference syncs >= 3 times);
//   - global scalars are reduced per workgroup and then in a fixed order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpm_device.h"

namespace mpm {

constexpr int LS_CAND = 28;          // alpha = 2^-j, j = 0..27 (alpha < 1e-8 is accepted as is)
constexpr int CT_PART = LS_CAND + 4; // partial record: E1[28], E0, norm_dir, dofs, pad

struct ContactState {       // device-resident solver state
    int done;               // converged or iteration cap reached
    int iters;
    int ls_total;           // accumulated line-search evaluations (statistics)
    int pad;
    float alpha;            // step accepted in the current iteration
    float residual;         // sqrt(sum |Dir|^2) / DoFs
    float energy;
    float E0;
    float norm_dir_sq;      // sum |Dir|^2 (before relaxation) of the current iteration
    float dofs;
    float pad2[2];
    double scal[4];         // exact line search: E, dE, d2E at the probed alpha
};

struct ContactDev {
    int n;                  // contacts
    int max_iters;
    float dt, mu, k, d, epsv, relax, tol;
    const uint32_t* slot;   // internal particle slot
    const uint32_t* body;
    const float *dist, *normal, *pos, *rigid_v, *p_WB;
    float *vel, *vel0;
    int* cnode;             // [27][n] compact grid index (active slot * 64 + cell) or -1
    float* cfx;             // [3][n]
    float* cmass;           // [n]
    float* cHG;             // [12][n] world-frame mass-weighted Hessian (9) and gradient (3)
    int* node_start;        // [cells + 1] CSR over compact grid cells
    int* node_fill;
    int2* entries;          // (contact, weight bits)
    float4* gD;             // [cells] search direction (relaxed)
    double* part;           // [workgroups][CT_PART]
    int part_wg;            // number of partial records of one kind
    ContactState* st;
    float* body_tau;
    float* body_f;
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
    int* cnode = nullptr;
    float* cfx = nullptr;
    float* cmass = nullptr;
    float* cHG = nullptr;
    int2* entries = nullptr;
    int* node_start = nullptr;
    int* node_fill = nullptr;
    float4* gD = nullptr;
    double* part = nullptr;
    ContactState* st = nullptr;
    float* body_tau = nullptr;  // F_Bq_W_tau
    float* body_f = nullptr;    // F_Bq_W_f

    void release() {
        void* ptrs[] = {slot, body, dist, normal, pos, rigid_v, p_WB, vel, vel0, cnode, cfx, cmass, cHG,
                        entries, node_start, node_fill, gD, part, st, body_tau, body_f};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        *this = ContactBuffers();
    }
    int resize_bodies(size_t nb, hipStream_t s) {
        if (nb > cap_bodies) {
            if (body_tau) (void)hipFree(body_tau);
            if (body_f) (void)hipFree(body_f);
            body_tau = body_f = nullptr;
            if (hipMalloc((void**)&body_tau, nb * 12) != hipSuccess) return -2;
            if (hipMalloc((void**)&body_f, nb * 12) != hipSuccess) return -2;
            cap_bodies = nb;
        }
        n_bodies = nb;
        // reset at the beginning of each time step (cuda_mpm_model.cu:334-337)
        if (cap_bodies) {
            (void)hipMemsetAsync(body_tau, 0, cap_bodies * 12, s);
            (void)hipMemsetAsync(body_f, 0, cap_bodies * 12, s);
        }
        return 0;
    }
};

constexpr int CT_WG = 256;       // threads per workgroup of the contact kernels
constexpr int CT_MAX_WG = 1024;  // partial-sum records per kind

// ---- set-up (once per UpdateContact) -----------------------------------------

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

// stencil of every contact: node indices into the compact grid, fx, particle mass
__global__ __launch_bounds__(256) void k_ct_stencil(DP p, ContactDev c) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n) return;
    const PSet& S = p.set[p.ctl->cur];
    uint32_t b[3];
    const uint32_t hi = (uint32_t)((1 << p.bits) - 3);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float x = c.pos[k * 3 + d];
        b[d] = min(base_cell(x, p.dxinv), hi);
        c.cfx[d * c.n + k] = x * p.dxinv - (float)b[d];
    }
    c.cmass[k] = S.q[0][c.slot[k]].w * p.M.density;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const uint32_t x = b[0] + i, y = b[1] + j, z = b[2] + l;
                const int a = p.lut_act[block_id(x >> 2, y >> 2, z >> 2)];
                const int g = a < 0 ? -1 : a * 64 + (int)(((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u));
                c.cnode[(i * 9 + j * 3 + l) * c.n + k] = g;
                if (g >= 0) atomicAdd(&c.node_fill[g], 1);
            }
}

// exclusive scan of node_fill -> node_start over all compact cells (single workgroup)
__global__ __launch_bounds__(1024) void k_ct_scan(DP p, ContactDev c) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncell = (int)p.ctl->n_active * 64;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += 1024) {
        const int g = base + tid;
        const int v = g < ncell ? c.node_fill[g] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int pre = s_carry;
        for (int q = 0; q < w; ++q) pre += s_w[q];
        if (g < ncell) {
            c.node_start[g] = pre + inc - v;
            c.node_fill[g] = 0;
        }
        __syncthreads();
        if (tid == 1023) s_carry = pre + inc;
        __syncthreads();
    }
    if (tid == 0) c.node_start[ncell] = s_carry;
}

MPM_DEV float stencil_weight(const float* wx, const float* wy, const float* wz, int n) {
    return wx[n / 9] * wy[(n / 3) % 3] * wz[n % 3];
}

__global__ __launch_bounds__(256) void k_ct_fill(DP p, ContactDev c) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n) return;
    float wx[3], wy[3], wz[3];
    bspline3(c.cfx[k], wx);
    bspline3(c.cfx[c.n + k], wy);
    bspline3(c.cfx[2 * c.n + k], wz);
#pragma unroll
    for (int n = 0; n < 27; ++n) {
        const int g = c.cnode[n * c.n + k];
        if (g < 0) continue;
        const int at = c.node_start[g] + atomicAdd(&c.node_fill[g], 1);
        c.entries[at] = make_int2(k, __float_as_int(stencil_weight(wx, wy, wz, n)));
    }
}

// order every node's list by contact id so that the per-node sums are reproducible
__global__ __launch_bounds__(256) void k_ct_sort_lists(DP p, ContactDev c) {
    const int ncell = (int)p.ctl->n_active * 64;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ncell; g += gridDim.x * 256) {
        const int a = c.node_start[g], b = c.node_start[g + 1];
        for (int i = a + 1; i < b; ++i) {
            const int2 e = c.entries[i];
            int j = i - 1;
            while (j >= a && c.entries[j].x > e.x) {
                c.entries[j + 1] = c.entries[j];
                --j;
            }
            c.entries[j + 1] = e;
        }
    }
}

// ---- per contact helpers -----------------------------------------------------
struct ContactFrame {
    float R[9];     // rows: tangent 1, tangent 2, normal (world -> contact)
    float v0[3];    // lagged particle velocity relative to the body, contact frame
    float phi0, mass;
};

MPM_DEV ContactFrame contact_frame(const DP& p, const ContactDev& c, int k) {
    ContactFrame f;
    const PSet& S = p.set[p.ctl->cur];
    const float nh[3] = {-c.normal[k * 3], -c.normal[k * 3 + 1], -c.normal[k * 3 + 2]};
    frame_from_normal(nh, f.R);
    const float4 pv = S.q[1][c.slot[k]];
    const float v0r[3] = {pv.x - c.rigid_v[k * 3], pv.y - c.rigid_v[k * 3 + 1], pv.z - c.rigid_v[k * 3 + 2]};
    mulv3(f.R, v0r, f.v0);
    f.phi0 = -c.dist[k];
    f.mass = c.cmass[k];
    return f;
}

// grid_to_particle_kernel<CONTACT_TRANSFER=true> (cuda_mpm_kernels.cuh:860-866, 891-894):
// velocity at the contact point from nodes with m > 1e-7
MPM_DEV void gather_contact_velocity(const DP& p, const ContactDev& c, int k, float* v) {
    float wx[3], wy[3], wz[3];
    bspline3(c.cfx[k], wx);
    bspline3(c.cfx[c.n + k], wy);
    bspline3(c.cfx[2 * c.n + k], wz);
    v[0] = v[1] = v[2] = 0.f;
#pragma unroll
    for (int n = 0; n < 27; ++n) {
        const int g = c.cnode[n * c.n + k];
        if (g < 0) continue;
        const float4 q = p.gv[g];
        if (q.w > 1e-7f) {
            const float w = stencil_weight(wx, wy, wz, n);
            v[0] += w * q.x;
            v[1] += w * q.y;
            v[2] += w * q.z;
        }
    }
}

__global__ __launch_bounds__(256) void k_ct_gather_vel(DP p, ContactDev c, float* out, int check_done) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n) return;
    if (check_done && c.st->done) return;
    float v[3];
    gather_contact_velocity(p, c, k, v);
    out[k * 3] = v[0];
    out[k * 3 + 1] = v[1];
    out[k * 3 + 2] = v[2];
}

// ---- one Newton iteration ------------------------------------------------------

// C1: refresh contact_vel (from the 2nd iteration on) and evaluate the contact Hessian and
// gradient in the world frame (cuda_mpm_kernels.cuh:1107-1154)
__global__ __launch_bounds__(256) void k_ct_contact_grad(DP p, ContactDev c, int first) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n || c.st->done) return;
    float v[3];
    if (first) {
        v[0] = c.vel[k * 3]; v[1] = c.vel[k * 3 + 1]; v[2] = c.vel[k * 3 + 2];
    } else {
        gather_contact_velocity(p, c, k, v);
        c.vel[k * 3] = v[0]; c.vel[k * 3 + 1] = v[1]; c.vel[k * 3 + 2] = v[2];
    }
    const ContactFrame f = contact_frame(p, c, k);
    const float vr[3] = {v[0] - c.rigid_v[k * 3], v[1] - c.rigid_v[k * 3 + 1], v[2] - c.rigid_v[k * 3 + 2]};
    float vl[3];
    mulv3(f.R, vr, vl);
    const ContactParams cp = {c.dt, c.mu, c.k, c.d, c.epsv};
    float CH[9], CG[3];
    contact_grad_hess(cp, f.phi0, f.v0, vl, CH, CG);
    // world frame: R^T G, R^T H R
    float RT[9], tmp[9], WH[9], WG[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) RT[j * 3 + i] = f.R[i * 3 + j];
    mulv3(RT, CG, WG);
    mul33(RT, CH, tmp);
    mul33(tmp, f.R, WH);
#pragma unroll
    for (int a = 0; a < 9; ++a) c.cHG[a * c.n + k] = f.mass * WH[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) c.cHG[(9 + a) * c.n + k] = f.mass * WG[a];
}

MPM_DEV void wg_reduce_store(double* vals, int count, double* out) {
    // vals: per-thread values; reduces over the workgroup (CT_WG threads) into out[count]
    __shared__ double s_red[CT_WG / 64][CT_PART];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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

// G1: per node, gather the Hessian/gradient of its contacts, add the inertia term and solve for
// the Newton direction (cuda_mpm_kernels.cuh:1217-1274)
__global__ __launch_bounds__(CT_WG) void k_ct_node_dir(DP p, ContactDev c) {
    const int ncell = (int)p.ctl->n_active * 64;
    double acc[2] = {0, 0};
    if (!c.st->done) {
        for (int g = blockIdx.x * CT_WG + threadIdx.x; g < ncell; g += gridDim.x * CT_WG) {
            float4 D = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 q = p.gv[g];
            const int a = c.node_start[g], b = c.node_start[g + 1];
            if (q.w > 0.f && b > a) {
                float H[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, G[3] = {0.f, 0.f, 0.f};
                for (int e = a; e < b; ++e) {
                    const int2 en = c.entries[e];
                    const float w = __int_as_float(en.y);
#pragma unroll
                    for (int t = 0; t < 9; ++t) H[t] += w * w * c.cHG[t * c.n + en.x];
#pragma unroll
                    for (int t = 0; t < 3; ++t) G[t] += w * c.cHG[(9 + t) * c.n + en.x];
                }
                float hn = 0.f, gn = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) hn += H[t] * H[t];
#pragma unroll
                for (int t = 0; t < 3; ++t) gn += G[t] * G[t];
                if ((double)sqrtf(hn) > 1e-7 || (double)sqrtf(gn) > 1e-7) {
                    const float4 vs = p.gvs[g];
                    H[0] -= q.w; H[4] -= q.w; H[8] -= q.w;
                    G[0] -= q.w * (q.x - vs.x);
                    G[1] -= q.w * (q.y - vs.y);
                    G[2] -= q.w * (q.z - vs.z);
                    float Hi[9], d[3];
                    inv33(H, Hi);
                    mulv3(Hi, G, d);
                    acc[0] += (double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    acc[1] += 1.0;
                    D = make_float4(d[0] * c.relax, d[1] * c.relax, d[2] * c.relax, 1.f);
                }
            }
            c.gD[g] = D;
        }
    }
    wg_reduce_store(acc, 2, c.part + (size_t)blockIdx.x * CT_PART + LS_CAND + 1);
}

// C2: contact part of the line-search energies for every candidate step
// (cuda_mpm_kernels.cuh:1276-1473 with global_line_search = true)
__global__ __launch_bounds__(CT_WG) void k_ct_ls_contact(DP p, ContactDev c, int exact, float alpha_probe) {
    double acc[CT_PART];
#pragma unroll
    for (int q = 0; q < CT_PART; ++q) acc[q] = 0;
    const bool live = !c.st->done;
    for (int k = blockIdx.x * CT_WG + threadIdx.x; live && k < c.n; k += gridDim.x * CT_WG) {
        float wx[3], wy[3], wz[3];
        bspline3(c.cfx[k], wx);
        bspline3(c.cfx[c.n + k], wy);
        bspline3(c.cfx[2 * c.n + k], wz);
        float ov[3] = {0.f, 0.f, 0.f}, dd[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 27; ++n) {
            const int g = c.cnode[n * c.n + k];
            if (g < 0) continue;
            const float w = stencil_weight(wx, wy, wz, n);
            const float4 q = p.gv[g], D = c.gD[g];
            ov[0] += w * q.x; ov[1] += w * q.y; ov[2] += w * q.z;
            dd[0] += w * D.x; dd[1] += w * D.y; dd[2] += w * D.z;
        }
        const ContactFrame f = contact_frame(p, c, k);
        const ContactParams cp = {c.dt, c.mu, c.k, c.d, c.epsv};
        const float rv[3] = {c.rigid_v[k * 3], c.rigid_v[k * 3 + 1], c.rigid_v[k * 3 + 2]};
        float ovl[3], ddl[3];
        {
            const float t[3] = {ov[0] - rv[0], ov[1] - rv[1], ov[2] - rv[2]};
            mulv3(f.R, t, ovl);
            mulv3(f.R, dd, ddl);
        }
        if (!exact) {
            acc[LS_CAND] += (double)(f.mass * contact_cost(cp, f.phi0, f.v0, ovl));
            float al = 1.f;
#pragma unroll 4
            for (int j = 0; j < LS_CAND; ++j) {
                const float nv[3] = {ovl[0] - al * ddl[0], ovl[1] - al * ddl[1], ovl[2] - al * ddl[2]};
                acc[j] += (double)(f.mass * contact_cost(cp, f.phi0, f.v0, nv));
                al *= .5f;
            }
        } else {
            const float al = alpha_probe;
            const float nv[3] = {ovl[0] - al * ddl[0], ovl[1] - al * ddl[1], ovl[2] - al * ddl[2]};
            float CH[9], CG[3];
            contact_grad_hess(cp, f.phi0, f.v0, nv, CH, CG);
            float t[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) t[a] = ddl[0] * CH[a] + ddl[1] * CH[3 + a] + ddl[2] * CH[6 + a];
            acc[0] += (double)(f.mass * contact_cost(cp, f.phi0, f.v0, nv));
            acc[1] += (double)(f.mass * dot3(CG, ddl));
            acc[2] += (double)(f.mass * dot3(t, ddl));
        }
    }
    wg_reduce_store(acc, LS_CAND + 1, c.part + (size_t)(c.part_wg + blockIdx.x) * CT_PART);
}

// G2: inertia part of the line-search energies (cuda_mpm_kernels.cuh:1536-1589)
__global__ __launch_bounds__(CT_WG) void k_ct_ls_grid(DP p, ContactDev c, int exact, float alpha_probe) {
    double acc[CT_PART];
#pragma unroll
    for (int q = 0; q < CT_PART; ++q) acc[q] = 0;
    const int ncell = (int)p.ctl->n_active * 64;
    if (!c.st->done) {
        for (int g = blockIdx.x * CT_WG + threadIdx.x; g < ncell; g += gridDim.x * CT_WG) {
            const float4 q = p.gv[g];
            if (!(q.w > 0.f)) continue;
            const float4 vs = p.gvs[g], D = c.gD[g];
            const float o[3] = {q.x - vs.x, q.y - vs.y, q.z - vs.z};
            if (!exact) {
                acc[LS_CAND] += (double)(.5f * q.w * (o[0] * o[0] + o[1] * o[1] + o[2] * o[2]));
                float al = 1.f;
#pragma unroll 4
                for (int j = 0; j < LS_CAND; ++j) {
                    const float n0 = o[0] - al * D.x, n1 = o[1] - al * D.y, n2 = o[2] - al * D.z;
                    acc[j] += (double)(.5f * q.w * (n0 * n0 + n1 * n1 + n2 * n2));
                    al *= .5f;
                }
            } else {
                const float al = alpha_probe;
                const float n0 = o[0] - al * D.x, n1 = o[1] - al * D.y, n2 = o[2] - al * D.z;
                acc[0] += (double)(.5f * q.w * (n0 * n0 + n1 * n1 + n2 * n2));
                acc[1] += (double)(-q.w * (n0 * D.x + n1 * D.y + n2 * D.z));
                acc[2] += (double)(q.w * (D.x * D.x + D.y * D.y + D.z * D.z));
            }
        }
    }
    wg_reduce_store(acc, LS_CAND + 1, c.part + (size_t)(2 * c.part_wg + blockIdx.x) * CT_PART);
}

// S: fixed-order sum of the partial records, choice of the step, convergence test
// (cuda_mpm_solver.cu:472-528, 567-570)
__global__ __launch_bounds__(64) void k_ct_decide(ContactDev c, int n_dir_wg, int n_con_wg, int n_grid_wg, int exact) {
    ContactState* st = c.st;
    if (st->done) return;
    const int lane = threadIdx.x;
    // lanes 0..28: energies; lane 29: norm_dir; lane 30: dofs
    double v = 0;
    if (lane <= LS_CAND) {
        for (int w = 0; w < n_con_wg; ++w) v += c.part[(size_t)(c.part_wg + w) * CT_PART + lane];
        for (int w = 0; w < n_grid_wg; ++w) v += c.part[(size_t)(2 * c.part_wg + w) * CT_PART + lane];
    } else if (lane <= LS_CAND + 2) {
        for (int w = 0; w < n_dir_wg; ++w) v += c.part[(size_t)w * CT_PART + lane];
    }
    if (lane == LS_CAND + 1) st->norm_dir_sq = (float)v;
    if (lane == LS_CAND + 2) st->dofs = (float)v;
    if (exact) {
        if (lane < 3) st->scal[lane] = v;
        return;
    }
    const float e = (float)v;
    const float E0 = __shfl(e, LS_CAND);
    const unsigned long long ok = __ballot(lane < LS_CAND && e <= E0);
    int j = ok ? __builtin_ctzll(ok) : LS_CAND - 1;  // "Tiny Alpha": accept 2^-27 anyway
    const float Ej = __shfl(e, j);
    const float nd = __shfl(e, LS_CAND + 1), dofs = __shfl(e, LS_CAND + 2);
    if (lane == 0) {
        st->alpha = ldexpf(1.f, -j);
        st->energy = Ej;
        st->E0 = E0;
        st->ls_total += j + 1;
        st->iters += 1;
        st->residual = sqrtf(nd) / dofs;   // NaN when there is no DoF: the loop stops, as in the reference
        if (!(st->residual > c.tol) || st->iters >= c.max_iters) st->done = 2;  // finish after this update
    }
}

// G3: v -= alpha Dir (cuda_mpm_kernels.cuh:1591-1614)
__global__ __launch_bounds__(CT_WG) void k_ct_apply(DP p, ContactDev c) {
    ContactState* st = c.st;
    if (st->done == 1) return;
    const float al = st->alpha;
    const int ncell = (int)p.ctl->n_active * 64;
    for (int g = blockIdx.x * CT_WG + threadIdx.x; g < ncell; g += gridDim.x * CT_WG) {
        float4 q = p.gv[g];
        if (!(q.w > 0.f)) continue;
        const float4 D = c.gD[g];
        q.x -= al * D.x; q.y -= al * D.y; q.z -= al * D.z;
        p.gv[g] = q;
    }
}

// turns "finish after this update" (2) into "finished" (1) once the update has been applied
__global__ void k_ct_latch(ContactDev c) {
    if (threadIdx.x == 0 && c.st->done == 2) c.st->done = 1;
}

// apply_contact_impulse_to_rigid_bodies (cuda_mpm_kernels.cuh:1616-1658)
__global__ __launch_bounds__(256) void k_ct_impulse(ContactDev c) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= c.n) return;
    const float m = c.cmass[k];
    float l[3], r[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        l[a] = m * -(c.vel[k * 3 + a] - c.vel0[k * 3 + a]);
        r[a] = c.pos[k * 3 + a] - c.p_WB[k * 3 + a];
    }
    const float h[3] = {r[1] * l[2] - l[1] * r[2], r[2] * l[0] - l[2] * r[0], r[0] * l[1] - l[0] * r[1]};
    const uint32_t b = c.body[k];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        atomicAdd(&c.body_tau[b * 3 + a], h[a]);
        atomicAdd(&c.body_f[b * 3 + a], l[a]);
    }
}

}  // namespace mpm
