// Device-side pieces of the TEAM transport that the contact kernels themselves use (k_ct_keys, k_ct_decide): the layout of
// a rank's region, the bounded flag wait, and the push / collect of the line-search sums.  See mpm_team.h for the
// protocol and for the zone-exchange and status kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpm_device.h"

namespace mpm {

constexpr int TEAM_MAX = 8;          // ranks of one node
constexpr int TEAM_RED = 32;         // doubles per rank and reduction (ContactState::red)

struct TeamState {                   // device memory of THIS rank, zero at set-up, never reset afterwards
    unsigned z_seq;                  // zone exchanges this rank has completed (k_team_zone_wait)
    unsigned z_cur;                  // the exchange whose slots k_team_zone_add reads (its parity)
    unsigned r_seq;                  // reductions this rank has completed (k_team_status<1>, k_ct_decide phase 3)
    unsigned cnt[2];                 // entries of the zone pack in flight (towards the left / right neighbour)
    unsigned timeouts;               // waits that gave up (diagnostics)
};

struct TeamDev {                     // kernel argument
    int on;                          // 0: not a team solve (the struct is ignored)
    int rank, world;
    int left, right;                 // neighbour ranks or -1
    TeamState* ts;
    char* peer[TEAM_MAX];            // every rank's region as this rank addresses it (peer[rank]: its own)
    unsigned zone_cap;               // blocks per zone slot
    unsigned long long zone_bytes;   // bytes of one zone slot
    unsigned long long timeout_ticks;
    int lo[2], hi[2];                // x-block ranges of the two zones (left cut, right cut)
};

// ---- region layout (host and device) ----------------------------------------------------------------------------
__host__ __device__ inline size_t team_zone_slot_bytes(size_t cap) {
    const size_t b = (((4 + cap) * 4 + 15) / 16) * 16 + cap * 64 * 3 * 16;   // header + ids, then 3 float4 per cell
    return (b + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t team_region_bytes(size_t zone_bytes) {
    return 4 * zone_bytes + (size_t)2 * TEAM_MAX * TEAM_RED * 8 + (size_t)(2 + TEAM_MAX) * 64;
}
__host__ __device__ inline char* team_zone_slot(char* base, size_t zone_bytes, int side, int parity) {
    return base + (size_t)(side * 2 + parity) * zone_bytes;
}
__host__ __device__ inline double* team_sum_slot(char* base, size_t zone_bytes, int parity, int src) {
    return reinterpret_cast<double*>(base + 4 * zone_bytes) + ((size_t)parity * TEAM_MAX + src) * TEAM_RED;
}
__host__ __device__ inline uint32_t* team_zone_flag(char* base, size_t zone_bytes, int side) {
    return reinterpret_cast<uint32_t*>(base + 4 * zone_bytes + (size_t)2 * TEAM_MAX * TEAM_RED * 8 + (size_t)side * 64);
}
__host__ __device__ inline uint32_t* team_sum_flag(char* base, size_t zone_bytes, int src) {
    return reinterpret_cast<uint32_t*>(base + 4 * zone_bytes + (size_t)2 * TEAM_MAX * TEAM_RED * 8 + (size_t)(2 + src) * 64);
}

// bounded wait for `flag >= seq` (sequence numbers only grow: signed difference)
MPM_DEV bool team_wait_flag(const uint32_t* flag, uint32_t seq, unsigned long long t0, unsigned long long timeout_ticks) {
    while (true) {
        if ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0) return true;
        if (wall_clock64() - t0 > timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(10);
    }
}

// ---- sums over all ranks ------------------------------------------------------------------------------------------
// One wave: lane l < TEAM_RED holds this rank's l-th partial sum; it goes into the slot [parity][this rank] of EVERY
// rank's region (its own included: the consumer adds all slots alike), then the flags.
MPM_DEV void team_push_sums(const TeamDev& t, double v) {
    const int lane = threadIdx.x & 63;
    const uint32_t q = t.ts->r_seq;
    const int parity = (int)(q & 1u);
    for (int d = 0; d < t.world; ++d) {
        double* slot = team_sum_slot(t.peer[d], t.zone_bytes, parity, t.rank);
        if (lane < TEAM_RED)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot) + lane, (unsigned long long)__double_as_longlong(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();   // (one wave: the fence is the wave's, it covers every lane's stores)
    // (MI355X_MICROARCH.md, "Compiler hazard": the wait behind the write-back may be dropped when the wave's vmcnt is
    // provably empty -- the flag could then overtake the data; an inline-asm wait is invisible to that pass)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int d = 0; d < t.world; ++d)
        if (lane == 0) __hip_atomic_store(team_sum_flag(t.peer[d], t.zone_bytes, t.rank), q + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Wave 0 of a workgroup: waits (bounded) until every rank's sums of this reduction have arrived, adds them IN RANK ORDER
// (the same order on every rank: identical bits, identical decisions) and returns lane l's total.  *ok = false: a rank
// never arrived.  Moves r_seq on.
MPM_DEV double team_collect_sums(const TeamDev& t, Ctl* ctl, bool* ok_out) {
    const int lane = threadIdx.x & 63;
    TeamState* ts = t.ts;
    const uint32_t q = ts->r_seq;
    const int parity = (int)(q & 1u);
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    if (lane < t.world) ok = team_wait_flag(team_sum_flag(t.peer[t.rank], t.zone_bytes, lane), q + 1u, t0, t.timeout_ticks);
    ok = __ballot(!ok) == 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // (system scope: the loads below must not be served from before the flags)
    double v = 0.0;
    if (lane < TEAM_RED)
        for (int r = 0; r < t.world; ++r) {
            const unsigned long long bits =
                __hip_atomic_load(reinterpret_cast<unsigned long long*>(team_sum_slot(t.peer[t.rank], t.zone_bytes, parity, r)) + lane,
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v += __longlong_as_double((long long)bits);
        }
    if (lane == 0) {
        if (!ok) {
            atomicOr(&ctl->error, ERR_HALO);
            ts->timeouts += 1u;
        }
        ts->r_seq = q + 1u;
    }
    *ok_out = ok;
    return v;
}

}  // namespace mpm
