// Initialisation kernels and everything that moves state across the C ABI:
// slot-order gathers/scatters, dense-grid views, the reference's slot sort.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "mpm_host.h"
#include "mpm_sort.h"

namespace mpm {

// initialize_fem_state_kernel (cuda_mpm_kernels.cuh:13-70), faces in original
// order (slot == original id at this point).  The per-face quarter volume is
// parked in G3[face * 3].x for k_init_vertex_volumes.
__global__ __launch_bounds__(256) void k_init_faces(DP p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Nf) return;
    const PSet& S = p.set[0];
    const float4 f3 = S.fq[3][i];
    const int s0 = __float_as_int(f3.y), s1 = __float_as_int(f3.z), s2 = __float_as_int(f3.w);
    const float4 xa = S.q[0][s0], xb = S.q[0][s1], xc = S.q[0][s2];
    const float4 va = S.q[1][s0], vb = S.q[1][s1], vc = S.q[1][s2];
    const float D0[3] = {xb.x - xa.x, xb.y - xa.y, xb.z - xa.z};
    const float D1[3] = {xc.x - xa.x, xc.y - xa.y, xc.z - xa.z};
    const float Ds[6] = {D0[0], D1[0], D0[1], D1[1], D0[2], D1[2]};
    float Q[9], R[6];
    givens_qr3<2>(Ds, Q, R);
    const float Dmat[4] = {R[0], R[1], 0.f, R[3]};
    float Di[4];
    inv2(Dmat, Di);
    const float cx = D0[1] * D1[2] - D1[1] * D0[2];
    const float cy = D0[2] * D1[0] - D1[2] * D0[0];
    const float cz = D0[0] * D1[1] - D1[0] * D0[1];
    const float v4 = sqrtf(cx * cx + cy * cy + cz * cz) / 8.f * p.dx;
    S.q[0][i] = make_float4((xa.x + xb.x + xc.x) / 3.f, (xa.y + xb.y + xc.y) / 3.f, (xa.z + xb.z + xc.z) / 3.f, v4);
    S.q[1][i] = make_float4((va.x + vb.x + vc.x) / 3.f, (va.y + vb.y + vc.y) / 3.f, (va.z + vb.z + vc.z) / 3.f, 0.f);
    pack_F(Q, S.fq[0][i], S.fq[1][i], S.f8[i]);
    S.c8[i] = 0.f;
    S.fq[2][i] = make_float4(Di[0], Di[1], Di[3], v4);   // (Di[2] = -0 / det: Dm is upper triangular)
    S.fq[3][i] = f3;   // (.x: the corners' ranks around their vertices, see DP::VF)
    const_cast<float4*>(p.dm_orig)[i] = make_float4(Di[0], Di[1], Di[2], Di[3]);
    p.G3[(size_t)i * 3].x = v4;
}

// a vertex with more than eight adjacent faces: the mark in its row of DP::VF that sends it to the adjacency CSR (slot ==
// original id before the first sort; re-sorts write it again at the vertex's new slot)
__global__ __launch_bounds__(256) void k_init_vertex_adjacency(DP p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.Nv) return;
    const int e0 = p.adj_off[k], e1 = p.adj_off[k + 1];
    if (e1 - e0 > 8) p.VF[vf_entry((unsigned)k, 0) * 3] = __uint_as_float(VF_MARK);
}

// vertex volume = sum of the quarter volumes of its faces (ascending face id)
__global__ __launch_bounds__(256) void k_init_vertex_volumes(DP p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.Nv) return;
    float v = 0.f;
    for (int e = p.adj_off[k]; e < p.adj_off[k + 1]; ++e) v += p.G3[(size_t)(p.adj_fc[e] >> 2) * 3].x;
    p.set[0].q[0][p.Nf + k].w = v;
}

// ---- slot-order views -------------------------------------------------------
enum Field { F_POS, F_VEL, F_VOL, F_AFFINE, F_FORCE, F_DEFGRAD, F_DMINV };

template <int FIELD>
struct FieldInfo;
template <> struct FieldInfo<F_POS> { static constexpr int N = 3; };
template <> struct FieldInfo<F_VEL> { static constexpr int N = 3; };
template <> struct FieldInfo<F_VOL> { static constexpr int N = 1; };
template <> struct FieldInfo<F_AFFINE> { static constexpr int N = 9; };
template <> struct FieldInfo<F_FORCE> { static constexpr int N = 3; };
template <> struct FieldInfo<F_DEFGRAD> { static constexpr int N = 9; };
template <> struct FieldInfo<F_DMINV> { static constexpr int N = 4; };

template <int FIELD>
MPM_DEV void read_field(const DP& p, const PSet& S, int j, float* o) {
    if (FIELD == F_POS) { const float4 q = S.q[0][j]; o[0] = q.x; o[1] = q.y; o[2] = q.z; }
    if (FIELD == F_VEL) { const float4 q = S.q[1][j]; o[0] = q.x; o[1] = q.y; o[2] = q.z; }
    if (FIELD == F_VOL) o[0] = fabsf(S.q[0][j].w);   // (the sign marks ghost copies in a partitioned domain)
    if (FIELD == F_AFFINE) unpack_C(S.q[1][j], S.q[2][j], S.q[3][j], o);
    if (FIELD == F_FORCE) {
        const bool v = j >= p.Nf;
        o[0] = v ? p.f[0][j] : 0.f; o[1] = v ? p.f[1][j] : 0.f; o[2] = v ? p.f[2][j] : 0.f;
    }
    if (FIELD == F_DEFGRAD) {
        unpack_F(S.fq[0][j], S.fq[1][j], S.f8[j], o);
    }
    if (FIELD == F_DMINV) {   // as Finalize computed it, entry [2] (a signed zero) included
        if (p.dm_orig) {
            const float4 c = p.dm_orig[S.pid[j]];
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
        } else {   // (partitioned domain: the table of the whole scene is gone; entry [2] is a zero either way)
            const float4 c = S.fq[2][j];
            o[0] = c.x; o[1] = c.y; o[2] = 0.f; o[3] = c.z;
        }
    }
}

template <int FIELD>
MPM_DEV void write_field(const DP& p, const PSet& S, int j, const float* v) {
    if (FIELD == F_POS) { float4 q = S.q[0][j]; q.x = v[0]; q.y = v[1]; q.z = v[2]; S.q[0][j] = q; }
    if (FIELD == F_VEL) { float4 q = S.q[1][j]; q.x = v[0]; q.y = v[1]; q.z = v[2]; S.q[1][j] = q; }
    if (FIELD == F_VOL) {
        S.q[0][j].w = S.q[0][j].w < 0.f ? -v[0] : v[0];
        if (j < p.Nf) S.fq[2][j].w = fabsf(v[0]);   // (the copy k_fem reads)
    }
    if (FIELD == F_AFFINE) {
        S.q[2][j] = make_float4(v[0], v[1], v[2], v[3]);
        S.q[3][j] = make_float4(v[4], v[5], v[6], v[7]);
        S.q[1][j].w = v[8];
        if (j < p.Nf) S.c8[j] = v[8];
    }
    if (FIELD == F_DEFGRAD) {
        pack_F(v, S.fq[0][j], S.fq[1][j], S.f8[j]);
    }
}

// out[s*N .. ] = field of the particle in slot s; `order` maps slot -> original id (the API slot
// order, or the identity for views in original order)
template <int FIELD>
__global__ __launch_bounds__(256) void k_gather_field(DP p, float* out, int n, const int* order) {
    constexpr int N = FieldInfo<FIELD>::N;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const PSet& S = p.set[p.ctl->cur];
    float v[9];
    const int j = p.imap[order[s]];
    if (j >= 0) {
        read_field<FIELD>(p, S, j, v);
    } else {   // partitioned domain: the particle is not on this rank
#pragma unroll
        for (int c = 0; c < 9; ++c) v[c] = __int_as_float(0x7FC00000);
    }
#pragma unroll
    for (int c = 0; c < N; ++c) out[(size_t)s * N + c] = v[c];
}

template <int FIELD>
__global__ __launch_bounds__(256) void k_scatter_field(DP p, const float* in, int n, const int* order) {
    constexpr int N = FieldInfo<FIELD>::N;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const PSet& S = p.set[p.ctl->cur];
    float v[9];
#pragma unroll
    for (int c = 0; c < N; ++c) v[c] = in[(size_t)s * N + c];
    const int j = p.imap[order[s]];
    if (j >= 0) write_field<FIELD>(p, S, j, v);
}

// taus()[slot] = a (x) b for face particles, zero for vertices
__global__ __launch_bounds__(256) void k_gather_taus(DP p, float* out, const int* pids_api) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.NpG) return;
    const int j = p.imap[pids_api[s]];
    float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
    if (j >= 0 && j < p.Nf) {
        const float3 q = p.ta[j];
        const float4 r = p.set[p.ctl->cur].fq[0][j];
        a[0] = q.x; a[1] = q.y; a[2] = q.z;
        b[0] = r.x; b[1] = r.y; b[2] = r.z;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(size_t)s * 9 + r * 3 + c] = a[r] * b[c];
}

// current_sort_keys(): the reference's key of the particle in each slot
// (compute_base_cell_node_index_kernel, cuda_mpm_kernels.cuh:365-382)
__global__ __launch_bounds__(256) void k_slot_keys(DP p, uint32_t* out, const int* pids_api) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.NpG) return;
    const PSet& S = p.set[p.ctl->cur];
    const int j = p.imap[pids_api[s]];
    if (j < 0) {
        out[s] = 0xFFFFFFFFu;
        return;
    }
    const float4 x = S.q[0][j];
    out[s] = cell_key(base_cell(x.x, p.dxinv), base_cell(x.y, p.dxinv), base_cell(x.z, p.dxinv));
}

// dense views of the compact grid: out_m[key], out_v[3*key]
__global__ __launch_bounds__(256) void k_densify(DP p, const float4* field, float* out_m, float* out_v) {
    const unsigned n = p.ctl->n_active * 64u;
    for (unsigned g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
        const float4 v = field[g];
        const size_t key = (size_t)p.act_block[g >> 6] * 64 + (g & 63u);
        if (out_m) out_m[key] = v.w;
        if (out_v) {
            out_v[key * 3 + 0] = v.x;
            out_v[key * 3 + 1] = v.y;
            out_v[key * 3 + 2] = v.z;
        }
    }
}

// grid_touched_flags(): blocks reached by any particle stencil in the last P2G
__global__ __launch_bounds__(256) void k_touched_flags(DP p, uint32_t* flags) {
    const unsigned n = p.ctl->n_items * 27u;
    for (unsigned w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
        const unsigned item = w / 27, o = w % 27;
        if ((p.slab_mask[item] >> o) & 1u) {
            const int nb = neighbor_block(p.home_block[p.item_desc[item].x], (int)o, p.nb);
            if (nb >= 0) flags[nb] = 1u;
        }
    }
}

}  // namespace mpm

// identity "slot -> original id" map on the device (views in original order)
static int resolve_contact_count(mpm_engine* e);   // (mpm_contact.h: the pair count may still be on the device)

static int device_iota(mpm_engine* e, int** out) {
    if (!e->d_iota) {
        std::vector<int> iota(e->np);
        std::iota(iota.begin(), iota.end(), 0);
        HIP_TRY(hipMalloc((void**)&e->d_iota, e->np * 4));
        e->allocs.push_back(e->d_iota);
        e->alloc_bytes.push_back(e->np * 4);
        H2D(e, e->d_iota, iota.data(), e->np * 4);
    }
    *out = e->d_iota;
    return 0;
}

template <int FIELD>
static int gather_to_host(mpm_engine* e, float* out, size_t n, const int* order) {
    constexpr int N = FieldInfo<FIELD>::N;
    if (n == 0) return 0;
    if (int rc = e->stage(n * N * 4)) return rc;
    hipLaunchKernelGGL(k_gather_field<FIELD>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->dp,
                       (float*)e->d_stage, (int)n, order);
    HIP_TRY(hipMemcpyAsync(out, e->d_stage, n * N * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

template <int FIELD>
static int scatter_from_host(mpm_engine* e, const float* in, size_t n, const int* order) {
    constexpr int N = FieldInfo<FIELD>::N;
    if (n == 0) return 0;
    if (int rc = e->stage(n * N * 4)) return rc;
    HIP_TRY(hipMemcpyAsync(e->d_stage, in, n * N * 4, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_scatter_field<FIELD>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->dp,
                       (const float*)e->d_stage, (int)n, order);
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// Vertex positions in original vertex order (DumpCpuState, cuda_mpm_model.cu:244-265)
static int download_original_vertices(mpm_engine* e, float* out) {
    int* iota = nullptr;
    if (int rc = device_iota(e, &iota)) return rc;
    return gather_to_host<F_POS>(e, out, e->nv, iota + e->nf);
}

// flags_out (nblocks uint32, may be null) and the touched count
static int touched_flags(mpm_engine* e, uint32_t* flags_out, uint32_t* cnt_out) {
    const size_t nb = e->dp.nblocks;
    if (int rc = e->stage(nb * 4)) return rc;
    HIP_TRY(hipMemsetAsync(e->d_stage, 0, nb * 4, e->stream));
    hipLaunchKernelGGL(k_touched_flags, dim3(64), dim3(256), 0, e->stream, e->dp, (uint32_t*)e->d_stage);
    std::vector<uint32_t> tmp;
    uint32_t* dst = flags_out;
    if (!dst) {
        tmp.resize(nb);
        dst = tmp.data();
    }
    HIP_TRY(hipMemcpyAsync(dst, e->d_stage, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (cnt_out) {
        uint32_t c = 0;
        for (size_t b = 0; b < nb; ++b) c += dst[b] != 0;
        *cnt_out = c;
    }
    return 0;
}

static int download_grid(mpm_engine* e, int which, void* out, size_t bytes, size_t* written) {
    const DP& p = e->dp;
    const size_t cells = p.ncells;
    const bool scalar = which == MPM_ARR_GRID_MASSES;
    const size_t need = cells * (scalar ? 4 : 12);
    REQUIRE(bytes >= need, "output buffer too small");
    REQUIRE(e->grid_state >= 1, "grid arrays are undefined before ParticleToGrid");
    const float4* field = p.gv;
    if (which == MPM_ARR_GRID_V_STAR) {
        REQUIRE(e->grid_state == 2, "grid_v_star is undefined before UpdateGrid");
        field = p.gvs;
    } else if (which == MPM_ARR_GRID_DIR) {
        // the relaxed Newton direction of the last UpdateContact iteration (nodes without contacts: 0)
        if (int rc_n = resolve_contact_count(e)) return rc_n;
        REQUIRE(e->cb.n > 0 && e->cb.gD && e->grid_state == 2, "grid_Dir is only defined after UpdateContact");
        field = e->cb.gD;
    } else if (e->grid_state == 3) {
        // raw sums already gathered (multi-GPU path)
    } else if (e->grid_state == 1) {
        // state right after ParticleToGrid: raw sums (mass, momentum)
        hipLaunchKernelGGL(k_grid<0>, dim3(e->g_grid), dim3(256), 0, e->stream, p, GridColliders{});
    }
    if (int rc = e->stage(need)) return rc;
    HIP_TRY(hipMemsetAsync(e->d_stage, 0, need, e->stream));
    hipLaunchKernelGGL(k_densify, dim3(256), dim3(256), 0, e->stream, p, field, scalar ? (float*)e->d_stage : nullptr,
                       scalar ? nullptr : (float*)e->d_stage);
    HIP_TRY(hipMemcpyAsync(out, e->d_stage, need, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (written) *written = need;
    return 0;
}

static int download_array(mpm_engine* e, int which, void* out, size_t bytes, size_t* written) {
    const size_t np = e->np, nf = e->nf;
    const DP& p = e->dp;
    int rc = 0;
    auto need = [&](size_t n) -> int {
        if (bytes < n) return fail(MPM_ERR_INVALID, "output buffer too small");
        if (written) *written = n;
        return 0;
    };
    switch (which) {
        case MPM_ARR_POSITIONS:
            if ((rc = need(np * 12))) return rc;
            return gather_to_host<F_POS>(e, (float*)out, np, e->d_pids_api);
        case MPM_ARR_VELOCITIES:
            if ((rc = need(np * 12))) return rc;
            return gather_to_host<F_VEL>(e, (float*)out, np, e->d_pids_api);
        case MPM_ARR_VOLUMES:
            if ((rc = need(np * 4))) return rc;
            return gather_to_host<F_VOL>(e, (float*)out, np, e->d_pids_api);
        case MPM_ARR_AFFINE:
            if ((rc = need(np * 36))) return rc;
            return gather_to_host<F_AFFINE>(e, (float*)out, np, e->d_pids_api);
        case MPM_ARR_PIDS:
        case MPM_ARR_INDEX_MAPPINGS: {
            if ((rc = need(np * 4))) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            D2H(e, out, which == MPM_ARR_PIDS ? e->d_pids_api : e->d_apimap, np * 4);
            return 0;
        }
        case MPM_ARR_SORT_KEYS: {
            if ((rc = need(np * 4))) return rc;
            if ((rc = e->stage(np * 4))) return rc;
            hipLaunchKernelGGL(k_slot_keys, dim3(e->g_np), dim3(256), 0, e->stream, p, (uint32_t*)e->d_stage,
                               e->d_pids_api);
            HIP_TRY(hipMemcpyAsync(out, e->d_stage, np * 4, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            return 0;
        }
        case MPM_ARR_FORCES:
            if ((rc = need(np * 12))) return rc;
            return gather_to_host<F_FORCE>(e, (float*)out, np, e->d_pids_api);
        case MPM_ARR_TAUS: {
            if ((rc = need(np * 36))) return rc;
            if ((rc = e->stage(np * 36))) return rc;
            hipLaunchKernelGGL(k_gather_taus, dim3(e->g_np), dim3(256), 0, e->stream, p, (float*)e->d_stage,
                               e->d_pids_api);
            HIP_TRY(hipMemcpyAsync(out, e->d_stage, np * 36, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            return 0;
        }
        case MPM_ARR_DEFORMATION_GRADIENTS:
        case MPM_ARR_DM_INVERSES: {
            // face arrays stay in original face order in the reference
            const int nc = which == MPM_ARR_DM_INVERSES ? 4 : 9;
            if ((rc = need(nf * nc * 4))) return rc;
            int* iota = nullptr;
            if ((rc = device_iota(e, &iota))) return rc;
            return nc == 4 ? gather_to_host<F_DMINV>(e, (float*)out, nf, iota)
                           : gather_to_host<F_DEFGRAD>(e, (float*)out, nf, iota);
        }
        case MPM_ARR_INDICES: {
            if ((rc = need(nf * 12))) return rc;
            int* o = static_cast<int*>(out);
            for (size_t k = 0; k < nf * 3; ++k) o[k] = e->h_idx[k] + (int)nf;
            return 0;
        }
        case MPM_ARR_GRID_MASSES:
        case MPM_ARR_GRID_MOMENTUM:
        case MPM_ARR_GRID_V_STAR:
        case MPM_ARR_GRID_DIR:
            return download_grid(e, which, out, bytes, written);
        case MPM_ARR_GRID_TOUCHED_FLAGS: {
            if ((rc = need((size_t)p.nblocks * 4))) return rc;
            REQUIRE(e->grid_state >= 1, "touched flags are undefined before ParticleToGrid");
            return touched_flags(e, (uint32_t*)out, nullptr);
        }
        case MPM_ARR_GRID_TOUCHED_IDS: {
            REQUIRE(e->grid_state >= 1, "touched ids are undefined before ParticleToGrid");
            std::vector<uint32_t> fl(p.nblocks);
            if ((rc = touched_flags(e, fl.data(), nullptr))) return rc;
            std::vector<uint32_t> ids;
            for (uint32_t b = 0; b < p.nblocks; ++b)
                if (fl[b]) ids.push_back(b);
            if ((rc = need(ids.size() * 4))) return rc;
            std::copy(ids.begin(), ids.end(), static_cast<uint32_t*>(out));
            return 0;
        }
        case MPM_ARR_CONTACT_VEL:
        case MPM_ARR_CONTACT_VEL0: {
            if ((rc = resolve_contact_count(e))) return rc;
            if ((rc = need(e->cb.n * 12))) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->cb.n)
                D2H(e, out, which == MPM_ARR_CONTACT_VEL ? e->cb.vel : e->cb.vel0, e->cb.n * 12);
            return 0;
        }
        default:
            return fail(MPM_ERR_INVALID, "unknown array id");
    }
}

static int upload_state(mpm_engine* e, const float* pos, const float* vel, const float* affine, const float* volumes,
                        const float* Fdef) {
    int rc = 0;
    const size_t np = e->np;
    if (pos) {
        if ((rc = scatter_from_host<F_POS>(e, pos, np, e->d_pids_api))) return rc;
        // positions changed arbitrarily: force a re-sort before the next transfer
        int one = 1;
        H2D(e, &e->dp.ctl->need_rebuild, &one, sizeof(int));
    }
    if (vel && (rc = scatter_from_host<F_VEL>(e, vel, np, e->d_pids_api))) return rc;
    if (affine && (rc = scatter_from_host<F_AFFINE>(e, affine, np, e->d_pids_api))) return rc;
    if (volumes && (rc = scatter_from_host<F_VOL>(e, volumes, np, e->d_pids_api))) return rc;
    if (Fdef && e->nf) {
        int* iota = nullptr;
        if ((rc = device_iota(e, &iota))) return rc;
        if ((rc = scatter_from_host<F_DEFGRAD>(e, Fdef, e->nf, iota))) return rc;
    }
    return 0;
}

// RebuildMapping(state, sort=true): re-order the slot order by a stable sort on
// the low min(3*bits,16) bits of the key (cuda_mpm_solver.cu:47-68).  Slot order
// is pure bookkeeping for this engine (pids/index_mappings), the particle data
// itself does not move.
namespace mpm {
// sort input: the masked key of every slot and the slot itself
__global__ __launch_bounds__(256) void k_api_sort_keys(DP p, const int* pids_api, uint32_t mask, uint32_t* keys,
                                                       uint32_t* vals) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.NpG) return;
    const PSet& S = p.set[p.ctl->cur];
    const float4 x = S.q[0][p.imap[pids_api[s]]];
    keys[s] = cell_key(base_cell(x.x, p.dxinv), base_cell(x.y, p.dxinv), base_cell(x.z, p.dxinv)) & mask;
    vals[s] = (uint32_t)s;
}
// new slot s holds the particle of old slot order[s]
__global__ __launch_bounds__(256) void k_api_relabel(int n, const uint32_t* order, const int* pids_old, int* pids_new,
                                                     int* apimap) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int pid = pids_old[order[s]];
    pids_new[s] = pid;
    apimap[pid] = s;
}
}  // namespace mpm

static int api_sort(mpm_engine* e) {
    using namespace mpm;
    const size_t np = e->np;
    if (!e->d_sort_keys) {
        int rc = 0;
        if ((rc = e->dalloc(&e->d_sort_keys, np, false)) || (rc = e->dalloc(&e->d_sort_vals, np, false)) ||
            (rc = e->dalloc(&e->d_sort_keys2, np, false)) || (rc = e->dalloc(&e->d_sort_vals2, np, false)) ||
            (rc = e->dalloc(&e->d_sort_hist, sort_hist_ints(np), false)) || (rc = e->dalloc(&e->d_pids_api2, np, false)))
            return rc;
    }
    const int nbits = std::min(3 * e->bits, 16);
    const uint32_t mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL(k_api_sort_keys, dim3(e->g_np), dim3(256), 0, e->stream, e->dp, (const int*)e->d_pids_api, mask,
                       e->d_sort_keys, e->d_sort_vals);
    if (radix_sort_pairs(e->stream, e->d_sort_keys, e->d_sort_vals, e->d_sort_keys2, e->d_sort_vals2, e->d_sort_hist, np,
                         nbits))
        return fail(MPM_ERR_HIP, "slot sort failed");
    hipLaunchKernelGGL(k_api_relabel, dim3(e->g_np), dim3(256), 0, e->stream, (int)np, (const uint32_t*)e->d_sort_vals,
                       (const int*)e->d_pids_api, e->d_pids_api2, e->d_apimap);
    std::swap(e->d_pids_api, e->d_pids_api2);
    e->api_identity = false;
    HIP_TRY(hipGetLastError());
    return 0;
}
