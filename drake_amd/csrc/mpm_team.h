// TEAM transport of the distributed contact solve (round 6): what the ranks of a partitioned domain
// exchange INSIDE a solve -- the per-node (H, G) sums of the zone blocks with the two neighbours, the line-search sums
// with everybody -- as stores into each other's memory plus sequence flags, on the engine's stream, with no collective
// library and no host in between.  The reference has one device (multibody/gpu_mpm/settings.h:40) and reads its global
// scalars back per iteration (cuda_mpm_solver.cu:318-319, 359-363, 515-517, 567-570); round 5's partitioned solve
// re-created those round trips ACROSS ranks (an all-reduce per line-search probe driven by the host).  Here the host
// only polls the mailbox, as on one GPU.
//
// Every rank owns ONE region of fine-grained device memory, mapped by every other rank of the node (HIP IPC, or the
// plain pointer for ranks of one process):
//   zone slots   [from the left | from the right] x [parity 0 | 1]   packed (H, G) sums of the neighbour's zone blocks
//   sum slots    [parity 0 | 1] x [source rank] x 32 doubles          every rank's partial line-search sums
//   flags        one 32-bit sequence number per zone side and per source rank, 64 bytes apart
// Protocol (the microarch guide's hand-off): data stores into the peer's slot; a kernel boundary (all waves' stores are
// complete) or, for a single wave, a system-scope fence; a system-scope RELEASE store of the sequence number into the
// peer's flag; the consumer ACQUIRES its own flags with system-scope loads until they carry the number, then reads.
// Two parities are enough (as for the direct halo): a rank can be at most one exchange ahead of a
// peer, because completing exchange q needs the peer's signal of q, which the peer sends after it has finished reading
// exchange q - 1.
//
// WHICH exchange a kernel belongs to is decided ON THE DEVICE (TeamState): the host enqueues iteration patterns
// speculatively, patterns behind the end of a solve skip themselves -- exchanges included --, and every rank takes the
// same decisions from the same global sums, so the ranks' counters stay equal although their hosts enqueue different
// numbers of idle patterns.  All sums are added IN RANK ORDER by every rank: identical bits everywhere, hence identical
// `E1 <= E0` decisions (cuda_mpm_solver.cu:518) without a collective.
//
// Every wait is BOUNDED (timeout_ticks of the 100 MHz wall clock): a peer that never arrives is MPM_ERR_HALO and a
// finished solve, never a hung device.
//
// What one GPU can check and what it cannot: the protocol, the indexing and the rank-order sums run here between
// processes that share the card and between the ranks of an in-process world; the ORDERING of peer stores across two
// devices over xGMI cannot be observed on one device -- the scopes are the guide's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpm_contact_dev.h"
#include "mpm_team_dev.h"

namespace mpm {

// gate of an exchange inside the iteration pattern: the kernels of a direction skip themselves when the solve is over or
// the line search of the previous direction is still running (k_ct_tile / k_ct_node_dir) -- and so must its exchange, on
// every rank alike (the state is identical on all ranks: it follows from the global sums)
MPM_DEV bool team_gate_closed(const ContactDev& c, int gate) {
    const ContactState* st = c.st;
    return gate == 1 ? (st->done != 0 || st->ls_phase != 0) : st->done != 0;
}

// ---- zone exchange of a per-node field (NV float4 per cell) with the two neighbours ---------------------------------
// pack: like k_zone_pack, but straight into the NEIGHBOUR's slot of this exchange's parity; entries are counted in a word
// of this rank's own memory (no returning atomic on peer memory), the count travels with the signal
template <int NV>
__global__ __launch_bounds__(256) void k_team_zone_pack(DP p, ContactDev c, TeamDev t, const float4* field, int gate) {
    if (team_gate_closed(c, gate)) return;
    const int side = blockIdx.y;                      // 0: my left zone -> the left neighbour, 1: right
    const int nbr = side == 0 ? t.left : t.right;
    if (nbr < 0) return;
    const int parity = (int)(t.ts->z_seq & 1u);
    // what goes to the left arrives "from the right" over there
    uint32_t* buf = reinterpret_cast<uint32_t*>(team_zone_slot(t.peer[nbr], t.zone_bytes, 1 - side, parity));
    float4* data = reinterpret_cast<float4*>(buf) + zone_data_offset(t.zone_cap);
    const unsigned n_active = p.ctl->n_active;
    for (unsigned a = blockIdx.x * 4 + (threadIdx.x >> 6); a < n_active; a += gridDim.x * 4) {
        int bx, by, bz;
        block_coords(p.act_block[a], bx, by, bz);
        if (bx < t.lo[side] || bx > t.hi[side]) continue;   // wave-uniform
        unsigned slot = 0;
        if ((threadIdx.x & 63) == 0) slot = atomicAdd(&t.ts->cnt[side], 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= t.zone_cap) {
            if ((threadIdx.x & 63) == 0) atomicOr(&p.ctl->error, ERR_CAPACITY);
            continue;
        }
        if ((threadIdx.x & 63) == 0) buf[4 + slot] = p.act_block[a];
        const size_t cell = (size_t)a * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int v = 0; v < NV; ++v) data[((size_t)slot * 64 + (threadIdx.x & 63)) * NV + v] = field[cell * NV + v];
    }
}
// one thread, behind the pack's kernel boundary: the counts into the neighbours' headers, then the flags
__global__ void k_team_zone_signal(ContactDev c, TeamDev t, int gate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (team_gate_closed(c, gate)) return;
    TeamState* ts = t.ts;
    const uint32_t seq = ts->z_seq + 1u;
    const int parity = (int)(ts->z_seq & 1u);
    for (int side = 0; side < 2; ++side) {
        const int nbr = side == 0 ? t.left : t.right;
        if (nbr < 0) continue;
        uint32_t* hdr = reinterpret_cast<uint32_t*>(team_zone_slot(t.peer[nbr], t.zone_bytes, 1 - side, parity));
        __hip_atomic_store(hdr, min(ts->cnt[side], t.zone_cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ts->cnt[side] = 0u;
    }
    __threadfence_system();
    // (MI355X_MICROARCH.md, "Compiler hazard": the wait behind the write-back may be dropped when the wave's vmcnt is
    // provably empty -- the flag could then overtake the data; an inline-asm wait is invisible to that pass)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int side = 0; side < 2; ++side) {
        const int nbr = side == 0 ? t.left : t.right;
        if (nbr < 0) continue;
        __hip_atomic_store(team_zone_flag(t.peer[nbr], t.zone_bytes, 1 - side), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// one thread: waits for both neighbours' signals of this exchange, then moves the counters on (the add kernel behind the
// kernel boundary reads z_cur: no kernel reads a counter that another thread of the same launch writes)
__global__ void k_team_zone_wait(ContactDev c, TeamDev t, int gate, Ctl* ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (team_gate_closed(c, gate)) return;
    TeamState* ts = t.ts;
    const uint32_t q = ts->z_seq;
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    if (t.left >= 0) ok &= team_wait_flag(team_zone_flag(t.peer[t.rank], t.zone_bytes, 0), q + 1u, t0, t.timeout_ticks);
    if (t.right >= 0) ok &= team_wait_flag(team_zone_flag(t.peer[t.rank], t.zone_bytes, 1), q + 1u, t0, t.timeout_ticks);
    if (!ok) {
        atomicOr(&ctl->error, ERR_HALO);
        ts->timeouts += 1u;
    }
    ts->z_cur = q;
    ts->z_seq = q + 1u;
}
template <int NV>
__global__ __launch_bounds__(256) void k_team_zone_add(DP p, ContactDev c, TeamDev t, float4* field, int gate) {
    if (team_gate_closed(c, gate)) return;
    const int side = blockIdx.y;
    if ((side == 0 ? t.left : t.right) < 0) return;
    const uint32_t* buf = reinterpret_cast<const uint32_t*>(team_zone_slot(t.peer[t.rank], t.zone_bytes, side, (int)(t.ts->z_cur & 1u)));
    const unsigned n = min(buf[0], t.zone_cap);
    const float4* data = reinterpret_cast<const float4*>(buf) + zone_data_offset(t.zone_cap);
    for (unsigned e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const uint32_t id = buf[4 + e];
        if (id >= p.nblocks) continue;
        const int a = p.lut_act[id];
        if (a < 0) continue;   // nothing of this rank reaches that block
        const size_t cell = (size_t)a * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 r = data[((size_t)e * 64 + (threadIdx.x & 63)) * NV + v];
            float4 q = field[cell * NV + v];
            q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w;
            field[cell * NV + v] = q;
        }
    }
}

// Status of the set-up, agreed by all ranks BEFORE anything of the solve runs: a rank that must refuse its solve (pair
// buffers overflowed, a guess of its host was wrong, a count it may not index with) makes every rank refuse -- a solve is
// a global affair, and a rank that ran on while a neighbour returned at once would wait for sums that never come --, and
// the solve is finished at once when NO rank has a contact (cuda_mpm_solver.cu:216-217, decided globally).
// PHASE 0 (one wave, behind k_ct_keys): push -- lane k < 8 carries 1 if this rank's code is k, lane 8 its contacts.
// PHASE 1 (one wave): collect, decide: the worst refusal code any rank reported; else finished if nobody has a contact.
template <int PHASE>
__global__ __launch_bounds__(64) void k_team_status(ContactDev c, TeamDev t, Ctl* ctl) {
    ContactState* st = c.st;
    const int lane = threadIdx.x & 63;
    if (PHASE == 0) {
        const int code = st->done >= CT_DONE_FAULT ? min(st->done, 7) : 0;
        const double v = lane < 8 ? (lane == code ? 1.0 : 0.0) : (lane == 8 ? (double)ct_count(c) : 0.0);
        team_push_sums(t, v);
        return;
    }
    bool ok = true;
    const double v = team_collect_sums(t, ctl, &ok);
    const unsigned long long refused = __ballot(lane >= CT_DONE_FAULT && lane < 8 && v > 0.0);
    const double total = __shfl(v, 8);
    if (lane != 0) return;
    if (!ok) {
        st->done = 1;          // a rank never arrived (MPM_ERR_HALO is raised): nothing is solved, nothing waits again
    } else if (refused) {
        st->done = 63 - __builtin_clzll(refused);   // the worst code: CORRUPT > GATED > STALE > FAULT
    } else {
        st->done = total > 0.0 ? 0 : 1;
    }
}

}  // namespace mpm

