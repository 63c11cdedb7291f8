// Initialisation kernels and everything that moves state across the C ABI:
// slot-order gathers/scatters, dense-grid views, the reference's slot sort.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "mpm_host.h"

namespace mpm {

// initialize_fem_state_kernel (cuda_mpm_kernels.cuh:13-70), faces in original
// order (slot == original id at this point).  The per-face quarter volume is
// parked in ab[0] for k_init_vertex_volumes.
__global__ __launch_bounds__(256) void k_init_faces(DP p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.Nf) return;
    const PSet& S = p.set[0];
    const int s0 = p.fv[0][i], s1 = p.fv[1][i], s2 = p.fv[2][i];
    float D0[3], D1[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float a = S.x[d][s0], b = S.x[d][s1], c = S.x[d][s2];
        S.x[d][i] = (a + b + c) / 3.f;
        S.v[d][i] = (S.v[d][s0] + S.v[d][s1] + S.v[d][s2]) / 3.f;
        D0[d] = b - a;
        D1[d] = c - a;
    }
    const float Ds[6] = {D0[0], D1[0], D0[1], D1[1], D0[2], D1[2]};
    float Q[9], R[6];
    givens_qr3<2>(Ds, Q, R);
    const float Dmat[4] = {R[0], R[1], 0.f, R[3]};
    float Di[4];
    inv2(Dmat, Di);
#pragma unroll
    for (int d = 0; d < 4; ++d) S.Dm[d][i] = Di[d];
#pragma unroll
    for (int d = 0; d < 9; ++d) S.F[d][i] = Q[d];
    const float cx = D0[1] * D1[2] - D1[1] * D0[2];
    const float cy = D0[2] * D1[0] - D1[2] * D0[0];
    const float cz = D0[0] * D1[1] - D1[0] * D0[1];
    const float v4 = sqrtf(cx * cx + cy * cy + cz * cz) / 8.f * p.dx;
    S.vol[i] = v4;
    p.ab[0][i] = v4;
}

// vertex volume = sum of the quarter volumes of its faces (ascending face id)
__global__ __launch_bounds__(256) void k_init_vertex_volumes(DP p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.Nv) return;
    float v = 0.f;
    for (int e = p.adj_off[k]; e < p.adj_off[k + 1]; ++e) v += p.ab[0][p.adj_fc[e] >> 2];
    p.set[0].vol[p.Nf + k] = v;
}

// ---- slot-order views -------------------------------------------------------
struct Planes {
    const float* q[9];
};
struct PlanesW {
    float* q[9];
};

// out[s*NC + c] = plane[c][ imap[ pids_api[s] ] ]; zero where the slot is outside [lo, hi)
template <int NC>
__global__ __launch_bounds__(256) void k_gather_slots(Planes pl, float* out, int n, const int* pids_api,
                                                      const int* imap, int lo, int hi) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int j = imap[pids_api[s]];
    const bool ok = j >= lo && j < hi;
#pragma unroll
    for (int c = 0; c < NC; ++c) out[(size_t)s * NC + c] = ok ? pl.q[c][j] : 0.f;
}

template <int NC>
__global__ __launch_bounds__(256) void k_scatter_slots(PlanesW pl, const float* in, int n, const int* pids_api,
                                                       const int* imap) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int j = imap[pids_api[s]];
#pragma unroll
    for (int c = 0; c < NC; ++c) pl.q[c][j] = in[(size_t)s * NC + c];
}

// taus()[slot] = a (x) b for face particles, zero for vertices
__global__ __launch_bounds__(256) void k_gather_taus(DP p, float* out, const int* pids_api) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= p.Np) return;
    const int j = p.imap[pids_api[s]];
    float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
    if (j < p.Nf) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            a[d] = p.ab[d][j];
            b[d] = p.ab[3 + d][j];
        }
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
    if (s >= p.Np) return;
    const PSet& S = p.set[p.ctl->cur];
    const int j = p.imap[pids_api[s]];
    out[s] = cell_key(base_cell(S.x[0][j], p.dxinv), base_cell(S.x[1][j], p.dxinv), base_cell(S.x[2][j], p.dxinv));
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
    const unsigned n = p.ctl->n_home * 27u;
    for (unsigned w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
        const unsigned h = w / 27, o = w % 27;
        if ((p.slab_mask[h] >> o) & 1u) {
            const int nb = neighbor_block(p.home_block[h], (int)o, p.nb);
            if (nb >= 0) flags[nb] = 1u;
        }
    }
}

}  // namespace mpm

static PSet current_set(mpm_engine* e, int* rc) {
    Ctl c{};
    hipError_t err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) err = hipMemcpy(&c, e->dp.ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
    *rc = err == hipSuccess ? 0 : fail(MPM_ERR_HIP, std::string("read control block: ") + hipGetErrorString(err));
    return e->dp.set[c.cur & 1];
}

template <int NC>
static int gather_to_host(mpm_engine* e, const float* const* planes, float* out, int lo, int hi) {
    const size_t n = e->np;
    if (int rc = e->stage(n * NC * 4)) return rc;
    Planes pl{};
    for (int c = 0; c < NC; ++c) pl.q[c] = planes[c];
    hipLaunchKernelGGL(k_gather_slots<NC>, dim3(e->g_np), dim3(256), 0, e->stream, pl, (float*)e->d_stage, (int)n,
                       e->d_pids_api, e->dp.imap, lo, hi);
    HIP_TRY(hipMemcpyAsync(out, e->d_stage, n * NC * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// Vertex positions in original vertex order (DumpCpuState, cuda_mpm_model.cu:244-265)
static int download_original_vertices(mpm_engine* e, float* out) {
    int rc = 0;
    const PSet S = current_set(e, &rc);
    if (rc) return rc;
    std::vector<int> iota(e->np);
    std::iota(iota.begin(), iota.end(), 0);
    // gather in original order = identity "slot -> original id" map
    if ((rc = e->stage(e->np * 12 + e->np * 4))) return rc;
    int* d_iota = reinterpret_cast<int*>(static_cast<char*>(e->d_stage) + e->np * 12);
    HIP_TRY(hipMemcpy(d_iota, iota.data(), e->np * 4, hipMemcpyHostToDevice));
    Planes pl{};
    for (int c = 0; c < 3; ++c) pl.q[c] = S.x[c];
    hipLaunchKernelGGL(k_gather_slots<3>, dim3(e->g_np), dim3(256), 0, e->stream, pl, (float*)e->d_stage, (int)e->np,
                       d_iota, e->dp.imap, 0, (int)e->np);
    HIP_TRY(hipMemcpyAsync(out, static_cast<char*>(e->d_stage) + e->nf * 12, e->nv * 12, hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
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
        REQUIRE(e->cb.n > 0 && e->grid_state == 2, "grid_Dir is only defined after UpdateContact");
        return fail(MPM_ERR_INVALID, "grid_Dir download not available");
    } else if (e->grid_state == 3) {
        // raw sums already gathered (multi-GPU path)
    } else if (e->grid_state == 1) {
        // state right after ParticleToGrid: raw sums (mass, momentum)
        hipLaunchKernelGGL(k_grid<0>, dim3(e->g_grid), dim3(256), 0, e->stream, p, -1);
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
        case MPM_ARR_VELOCITIES: {
            if ((rc = need(np * 12))) return rc;
            const PSet S = current_set(e, &rc);
            if (rc) return rc;
            const float* pl[3];
            for (int d = 0; d < 3; ++d) pl[d] = which == MPM_ARR_POSITIONS ? S.x[d] : S.v[d];
            return gather_to_host<3>(e, pl, (float*)out, 0, (int)np);
        }
        case MPM_ARR_VOLUMES: {
            if ((rc = need(np * 4))) return rc;
            const PSet S = current_set(e, &rc);
            if (rc) return rc;
            const float* pl[1] = {S.vol};
            return gather_to_host<1>(e, pl, (float*)out, 0, (int)np);
        }
        case MPM_ARR_AFFINE: {
            if ((rc = need(np * 36))) return rc;
            const PSet S = current_set(e, &rc);
            if (rc) return rc;
            const float* pl[9];
            for (int d = 0; d < 9; ++d) pl[d] = S.C[d];
            return gather_to_host<9>(e, pl, (float*)out, 0, (int)np);
        }
        case MPM_ARR_PIDS:
        case MPM_ARR_INDEX_MAPPINGS: {
            if ((rc = need(np * 4))) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            HIP_TRY(hipMemcpy(out, which == MPM_ARR_PIDS ? e->d_pids_api : e->d_apimap, np * 4, hipMemcpyDeviceToHost));
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
        case MPM_ARR_FORCES: {
            if ((rc = need(np * 12))) return rc;
            const float* pl[3] = {p.f[0], p.f[1], p.f[2]};
            return gather_to_host<3>(e, pl, (float*)out, (int)nf, (int)np);
        }
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
            const PSet S = current_set(e, &rc);
            if (rc) return rc;
            std::vector<int> iota(np);
            std::iota(iota.begin(), iota.end(), 0);
            if ((rc = e->stage(np * 36 + np * 4))) return rc;
            int* d_iota = reinterpret_cast<int*>(static_cast<char*>(e->d_stage) + np * 36);
            HIP_TRY(hipMemcpy(d_iota, iota.data(), np * 4, hipMemcpyHostToDevice));
            Planes pl{};
            for (int c = 0; c < nc; ++c) pl.q[c] = nc == 4 ? S.Dm[c] : S.F[c];
            if (nf == 0) return 0;
            if (nc == 4)
                hipLaunchKernelGGL(k_gather_slots<4>, dim3(e->g_nf), dim3(256), 0, e->stream, pl, (float*)e->d_stage,
                                   (int)nf, d_iota, p.imap, 0, (int)nf);
            else
                hipLaunchKernelGGL(k_gather_slots<9>, dim3(e->g_nf), dim3(256), 0, e->stream, pl, (float*)e->d_stage,
                                   (int)nf, d_iota, p.imap, 0, (int)nf);
            HIP_TRY(hipMemcpyAsync(out, e->d_stage, nf * nc * 4, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            return 0;
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
            if ((rc = need(e->cb.n * 12))) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->cb.n)
                HIP_TRY(hipMemcpy(out, which == MPM_ARR_CONTACT_VEL ? e->cb.vel : e->cb.vel0, e->cb.n * 12,
                                  hipMemcpyDeviceToHost));
            return 0;
        }
        default:
            return fail(MPM_ERR_INVALID, "unknown array id");
    }
}

static int upload_state(mpm_engine* e, const float* pos, const float* vel, const float* affine, const float* volumes,
                        const float* Fdef) {
    int rc = 0;
    const PSet S = current_set(e, &rc);
    if (rc) return rc;
    const size_t np = e->np;
    auto put = [&](const float* src, int nc, float* const* planes) -> int {
        if (int r = e->stage(np * nc * 4)) return r;
        HIP_TRY(hipMemcpyAsync(e->d_stage, src, np * nc * 4, hipMemcpyHostToDevice, e->stream));
        PlanesW pl{};
        for (int c = 0; c < nc; ++c) pl.q[c] = planes[c];
        if (nc == 3)
            hipLaunchKernelGGL(k_scatter_slots<3>, dim3(e->g_np), dim3(256), 0, e->stream, pl, (const float*)e->d_stage,
                               (int)np, e->d_pids_api, e->dp.imap);
        else if (nc == 9)
            hipLaunchKernelGGL(k_scatter_slots<9>, dim3(e->g_np), dim3(256), 0, e->stream, pl, (const float*)e->d_stage,
                               (int)np, e->d_pids_api, e->dp.imap);
        else
            hipLaunchKernelGGL(k_scatter_slots<1>, dim3(e->g_np), dim3(256), 0, e->stream, pl, (const float*)e->d_stage,
                               (int)np, e->d_pids_api, e->dp.imap);
        HIP_TRY(hipStreamSynchronize(e->stream));
        return 0;
    };
    if (pos) {
        float* pl[3] = {S.x[0], S.x[1], S.x[2]};
        if ((rc = put(pos, 3, pl))) return rc;
        // positions changed arbitrarily: force a re-sort before the next transfer
        int one = 1;
        HIP_TRY(hipMemcpy(&e->dp.ctl->need_rebuild, &one, sizeof(int), hipMemcpyHostToDevice));
    }
    if (vel) {
        float* pl[3] = {S.v[0], S.v[1], S.v[2]};
        if ((rc = put(vel, 3, pl))) return rc;
    }
    if (affine) {
        float* pl[9];
        for (int d = 0; d < 9; ++d) pl[d] = S.C[d];
        if ((rc = put(affine, 9, pl))) return rc;
    }
    if (volumes) {
        float* pl[1] = {S.vol};
        if ((rc = put(volumes, 1, pl))) return rc;
    }
    if (Fdef && e->nf) {
        // original face order: "slot -> original id" is the identity for this view
        const size_t nf = e->nf;
        std::vector<int> iota(nf);
        std::iota(iota.begin(), iota.end(), 0);
        if ((rc = e->stage(nf * 36 + nf * 4))) return rc;
        int* d_iota = reinterpret_cast<int*>(static_cast<char*>(e->d_stage) + nf * 36);
        HIP_TRY(hipMemcpy(d_iota, iota.data(), nf * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpyAsync(e->d_stage, Fdef, nf * 36, hipMemcpyHostToDevice, e->stream));
        PlanesW pl{};
        for (int c = 0; c < 9; ++c) pl.q[c] = S.F[c];
        hipLaunchKernelGGL(k_scatter_slots<9>, dim3(e->g_nf), dim3(256), 0, e->stream, pl, (const float*)e->d_stage,
                           (int)nf, d_iota, e->dp.imap);
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return 0;
}

// RebuildMapping(state, sort=true): re-order the slot order by a stable sort on
// the low min(3*bits,16) bits of the key (cuda_mpm_solver.cu:47-68).  Slot order
// is pure bookkeeping for this engine (pids/index_mappings), the particle data
// itself does not move.
static int api_sort(mpm_engine* e) {
    const size_t np = e->np;
    std::vector<uint32_t> keys(np);
    std::vector<int> pids(np);
    if (int rc = download_array(e, MPM_ARR_SORT_KEYS, keys.data(), np * 4, nullptr)) return rc;
    HIP_TRY(hipMemcpy(pids.data(), e->d_pids_api, np * 4, hipMemcpyDeviceToHost));
    const int nbits = std::min(3 * e->bits, 16);
    const uint32_t mask = nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
    std::vector<uint32_t> order(np);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return (keys[a] & mask) < (keys[b] & mask); });
    std::vector<int> npids(np), nmap(np);
    for (size_t s = 0; s < np; ++s) {
        npids[s] = pids[order[s]];
        nmap[npids[s]] = (int)s;
    }
    HIP_TRY(hipMemcpy(e->d_pids_api, npids.data(), np * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_apimap, nmap.data(), np * 4, hipMemcpyHostToDevice));
    e->api_identity = false;
    return 0;
}
