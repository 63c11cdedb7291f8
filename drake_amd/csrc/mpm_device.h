// Device-side data model of the engine (see DESIGN.md "Data layout in HBM").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpm_math.h"

namespace mpm {

// A home block's LDS tile covers its own 4^3 cells plus a free zone of
// FREE_ZONE cells on every side, plus the 2 extra nodes a quadratic stencil
// reaches beyond its base cell: 4 + 2*FREE_ZONE + 2 nodes per axis.
constexpr int FREE_ZONE = 2;
constexpr int TILE_W = 4 + 2 * FREE_ZONE + 2;        // 10
constexpr int TILE_N = TILE_W * TILE_W * TILE_W;     // 1000 nodes (16 KB of float4)
// A particle may sit up to FREE_ZONE cells outside its home block (hard limit); it asks for a
// re-sort when its next substep could take it past that (soft_zone_exit in mpm_step.h).

constexpr unsigned ERR_DRIFT = 1u;      // particle outside the hard free zone
constexpr unsigned ERR_CAPACITY = 2u;   // home/active table overflow
constexpr unsigned ERR_DOMAIN = 4u;     // particle base cell outside the grid
constexpr unsigned ERR_RANGE = 8u;      // a P2G node sum left the range of the fixed-point tile (or was NaN)
constexpr unsigned ERR_HALO = 16u;      // partitioned domain: a particle's stencil left the zone shared with the neighbour
constexpr unsigned ERR_SLABS = 32u;     // the re-sort made more work items than the slab pool holds (Ctl::n_items_wanted): the host
                                        // grows the pool and repeats the re-sort (recover_slab_overflow, mpm_engine.hip)

// One of the two ping-pong particle sets.  Every particle is four 16-byte records in four
// planes (one coalesced dwordx4 access per plane and wave), slots [0,Nf) are face particles,
// [Nf,Np) vertex particles, each range sorted by cell key at the last rebuild:
//   q[0] = (x, y, z, vol)      q[1] = (vx, vy, vz, C8)
//          (partitioned domain: vol > 0 owned, vol < 0 ghost copy of a neighbour's particle with
//           volume |vol|, vol == 0 released: dropped by the next re-sort)
//   q[2] = (C0, C1, C2, C3)    q[3] = (C4, C5, C6, C7)          C row-major, as in the reference
// Face particles carry four more records and two scalars, indexed by face slot.  What CalcFemStateAndForce
// rewrites every substep (F) is kept apart from what only a re-sort moves (Dm^-1, volume, corners), so
// that k_fem reads and writes exactly the 36 bytes of F:
//   fq[0] = (F2, F5, F8, F0)  fq[1] = (F1, F3, F4, F6)  f8 = F7     (written by k_fem; pack_F below: the normal
//           column F[:,2] first, k_p2g reads it from here as the second factor of tau)
//   fq[2] = (Dm0, Dm1, Dm3, |vol|)   Dm^-1 of a QR cloth is upper triangular (cuda_mpm_kernels.cuh:13-70:
//           Dm = R of a Givens QR), Dm^-1[2] is +-0 and not stored; |vol| is a copy of |q[0].w|
//   fq[3] = (0, v0, v1, v2)   v* = slots of the three corner vertices (int bits)
//   c8    = C8 of the face particle as GridToParticle left it: k_fem reads it here instead of fetching
//           the 16-byte record q[1] for 4 useful bytes (q[1].w carries the same value for ParticleToGrid)
constexpr unsigned VF_MARK = 0x7FB0A5C3u;   // (a NaN pattern: see DP::VF)
// entry (rank j, vertex k) of DP::VF, in units of 12-byte entries: vertices in chunks of 32, a chunk's eight rank planes
// next to each other (8 x 384 bytes = 24 full 128-byte lines)
constexpr unsigned VF_CHUNK = 32;
__host__ __device__ inline size_t vf_entry(unsigned k, unsigned j) {
    return (size_t)(k / VF_CHUNK) * (8u * VF_CHUNK) + j * VF_CHUNK + (k % VF_CHUNK);
}
struct PSet {
    float4* q[4];
    int* pid;      // slot -> original particle id ([faces | verts] order of Finalize)
    float4* fq[4];
    float* f8;
    float* c8;
};

MPM_DEV void unpack_C(const float4& q1, const float4& q2, const float4& q3, float* C) {
    C[0] = q2.x; C[1] = q2.y; C[2] = q2.z; C[3] = q2.w;
    C[4] = q3.x; C[5] = q3.y; C[6] = q3.z; C[7] = q3.w;
    C[8] = q1.w;
}

// Device-resident control block; everything the host would otherwise have to
// read back between kernels.
struct Ctl {
    int cur;               // index of the current PSet
    int need_rebuild;      // raised by G2P when an advected particle no longer fits its block tile
    unsigned error;        // sticky ERR_* bits
    unsigned skipped;      // substeps that were enqueued without their re-sort launches and found need_rebuild set:
                           // their kernels returned at once, the host runs them again later (settle, mpm_engine.hip)
    int skip_this;         // k_grid's verdict for the G2P of the same substep (G2P itself raises need_rebuild)
    unsigned watch_hit;    // number of the last k_ct_watch launch that found a particle inside (or a hair's breadth from) a
                           // collider: substeps enqueued as contact-free on the strength of an earlier watch skip themselves
                           // (DP::gated bit 2, DP::watch_base)
    unsigned n_home;       // blocks owning particles (tiles)
    unsigned n_active;     // blocks whose nodes are updated
    unsigned n_items;      // work items of the tile kernels (home blocks, heavy ones split)
    unsigned rebuilds;
    unsigned ticket;
    // Active particles occupy slots [0, nfa) (faces) and [Nf, Nf + nva) (vertices); the other slots
    // are free.  A single-domain engine has nfa = Nf, nva = Nv for ever.  In a partitioned domain
    // (mpm_dist_init) a rank only holds the particles it owns plus ghost copies of the neighbours'
    // particles next to the cuts: migration appends add_f / add_v particles behind the active ones,
    // the next re-sort drops the released ones and merges the new ones.
    int nfa, nva;
    int add_f, add_v;
    int nfa_new, nva_new;   // counts after the re-sort in flight (k_rb_tables -> k_rb_finish)
    unsigned n_items_wanted;   // work items the last re-sort made, before the clamp to the slab pool (> n_items: ERR_SLABS)
    // Quiet time: how long after the last re-sort no particle can leave its tile if all of them keep moving
    // ballistically (velocity + gravity) -- k_rb_count's estimate, the minimum over the particles.  A hint for the
    // host (mpm_run_substeps launches no re-sort checks while it holds, see launch_substep); nothing depends on it
    // being right: a substep that finds a re-sort pending without its check launches skips itself (DP::gated).
    float quiet_time;          // seconds, as of the last re-sort (0: not estimated)
    float time_since_resort;   // seconds of substeps run since then (k_g2p adds its dt)
    // Partitioned domain: how long after the last migration no particle of this rank can have drifted further along x
    // than the bands allow (Dist::mig_delta) if all of them keep moving ballistically -- k_dist_classify's estimate, the
    // minimum over the particles this rank holds (k_dist_mig_reduce).  The ranks agree on the minimum and migrate again
    // when half of it has passed (mpm_chain_substeps, drake_amd/dist.py: DomainChain).
    float mig_quiet;
};

// Partitioned domain: ranks cut ONE domain into x slabs at block boundaries.  Every
// rank was finalised with the whole scene; mpm_dist_init shrinks it to the particles it owns -- base
// cell x in [own_lo, own_hi) -- plus ghost copies of the neighbours' particles within the ghost bands of a
// cut, with the mesh topology kept per slot by original id (DP::fg / vg) instead of scene-sized tables.  Ghosts take part in FEM and G2P (so they stay
// bit-identical to the owner's copy without communication) and contribute nothing to P2G; the node
// sums of the blocks within zone_cells of a cut are exchanged every substep.
struct Dist {
    int on;
    int rank, world;
    int own_lo, own_hi;        // owned x range in cells
    int nbr_lo, nbr_hi;        // outer ends of the left / right neighbour's ranges (cells)
    int has_left, has_right;
    // Ghost bands, in cells beyond a cut, measured on xi = x / dx - 0.5 (the coordinate whose floor is the base cell):
    // fractional widths, so that a thin zone still leaves room for drift between two migrations (mpm_dist_init).
    float ghost_w;             // ghost band for face particles
    float vert_w;              // ghost band for vertex particles: wider by the reach of a face (centroid -> corner), so
                               // that a ghost face always finds its corner vertices on the same rank
    float hyst;                // hysteresis (cells) of ownership and of band membership: a particle changes owner when it
                               // is this far beyond a cut, and a ghost that was inside a band stays until it is this far
                               // outside -- a cloth that vibrates about a cut or a band edge causes no traffic (and no
                               // re-sort: a migration that moves nothing forces none)
    int zone_cells;            // depth of the exchanged zone either side of a cut
    float mig_delta;           // drift along x (cells) that every held particle may accumulate between two migrations
    float mig_reach;           // only particles within this distance (cells) of a cut can matter before they have
                               // travelled there: the others' time estimate includes the way to this region
    unsigned* mig_min;         // 32 slots, 128 bytes apart: complement of the smallest time estimate (k_dist_classify)
    unsigned char* prev;       // [original id] bit 0 / 1: was in the left / right neighbour's ghost band at the last migration
};

// xi of a position: base cell = floor(xi) (clamped to the grid)
MPM_DEV float dist_xi(float x, float dxinv) { return x * dxinv - .5f; }
// a neighbour's particle (base cell bx outside [own_lo, own_hi)) that I hold as a ghost; `extra` widens the band (the
// hysteresis for a ghost that is already held)
MPM_DEV bool dist_in_my_band(const Dist& d, int bx, float xi, bool face, float extra = 0.f) {
    const float w = (face ? d.ghost_w : d.vert_w) + extra;
    return (d.has_left && bx < d.own_lo && xi >= (float)d.own_lo - w) || (d.has_right && bx >= d.own_hi && xi < (float)d.own_hi + w);
}
// bit 0 / 1: an owned particle at xi is inside the band in which the left / right neighbour keeps ghosts (the same
// comparison of the same float on both ranks: owner and ghost rank always agree); `old` = the bits of the last
// migration: a side that was set stays set until the particle is d.hyst beyond the band
MPM_DEV int dist_in_neighbour_bands(const Dist& d, float xi, bool face, int old = 0) {
    const float w = face ? d.ghost_w : d.vert_w;
    const float wl = w + ((old & 1) ? d.hyst : 0.f), wr = w + ((old & 2) ? d.hyst : 0.f);
    return ((d.has_left && xi < (float)d.own_lo + wl) ? 1 : 0) | ((d.has_right && xi >= (float)d.own_hi - wr) ? 2 : 0);
}

struct DP {
    // SLOT space: face particles live in slots [0, Nf), vertex particles in [Nf, Np).  In a single-domain engine that
    // is also the id space of Finalize ([faces | verts]); a partitioned engine (mpm_dist_init) shrinks its slot space to
    // what the rank holds (share + ghost bands + head room) while the ids stay those of the whole scene:
    int Np, Nf, Nv;
    int NpG, NfG;          // ID space: original ids [0, NfG) are faces, [NfG, NpG) vertices (= Nf, Np unless partitioned)
    int bits;              // grid is (1<<bits)^3 cells
    int nb;                // blocks per axis
    unsigned nblocks, ncells;
    unsigned capH, capA, capI;
    unsigned capS;         // slabs allocated (<= capI): grown by the host at synchronisation points, see slab_pool_grow
    int item_groups;       // a work item holds at most this many 64-particle wave groups ...
    int item_groups_small; // ... or this many when the engine holds fewer than item_small_below wave groups in all (a
    int item_small_below;  // partitioned rank with a small share): more, smaller items keep the CUs busy (k_rb_tables)
    int dbg;               // MPM_DBG environment variable (kernel ablation switches, 0 in production)
    float dx, dxinv, Dinv;
    unsigned q_stride, f_stride;   // distance (elements) between the planes PSet::q[0..3] and f[0..2] (one allocation each)
    int gated;             // bit 0: this substep was enqueued without the re-sort launches and returns at once when it finds a
                           // re-sort pending; bit 1: it returns at once when the slab pool has overflowed (every substep
                           // of mpm_run_substeps; the phase-by-phase calls cannot be repeated by the engine).  Either way it
                           // counts itself in Ctl::skipped and the host runs it again (settle, mpm_engine.hip)
    unsigned watch_base;   // gated bit 2 (mpm_run_coupled_substeps: a substep enqueued WITHOUT pair generation and contact solve, on
                           // the strength of a k_ct_watch that found no particle in any collider): it returns at once when a watch
                           // numbered >= watch_base has found one since (Ctl::watch_hit), counts itself in Ctl::skipped, and the
                           // host runs it again as a coupled substep
    int lean_resort;       // 1: the conditional re-sort is followed by CalcFemStateAndForce at once (a whole substep was
                           // enqueued): k_rb_finish does not move the x / v records of face particles; per launch
    int lean_g2p;          // 1: another substep of the same mpm_run_substeps batch follows, nobody can look at the state
                           // in between: k_g2p leaves the x / v records of the face particles alone (see g2p_particle),
                           // k_p2g keeps the vertex forces of its work items in LDS only; per substep
    int fem_fast;          // mpm_set_fast_math: k_fem<1> runs (the re-sort forms a face's centroid the way k_fem will)
    float anticip;         // re-sort: cells a particle is binned ahead per unit of velocity (0 = by position), see k_rb_count
    // fixed-point scales of the LDS tile accumulators (powers of two), see k_p2g
    double fix_m, fix_p, unfix_m, unfix_p;
    Material M;
    Ctl* ctl;
    unsigned long long* dbgbuf;  // 16 diagnostic counters (only written when dbg & 4)
    PSet set[2];
    // per-substep scratch
    // partitioned domain, per-rank topology (null otherwise): what a particle needs to find its mesh neighbours by
    // ORIGINAL id, kept per slot, moved by the re-sort and carried by the migration records -- no array of the size of
    // the whole scene's topology stays on a rank
    int4* fg[2];           // [set][face slot] original ids of the three corner vertices; .w = the corners' ranks (fq[3].x)
    int4* vg[2][2];        // [set][2][vertex slot - Nf] up to eight (original face id << 2 | corner), -1 = none
    float3* ta;            // faces: tau = a (x) b with a = vol*P[:,2] (12-byte records); b = F[:,2] is the first three
                           // floats of fq[0] (see pack_F)
    float3* G3;            // faces: G3[face slot * 3 + c] = force triple the face exerts on corner c (negated when
                           // applied); 12-byte records: 36 B written per face and one dwordx3 gather per adjacency
    // The same triples where the VERTEX looks for them, VF[vf_entry(vertex slot - Nf, j) * 3 ..],
    // j = the rank of the face among the faces around that vertex (ascending original face id: the order of the sum).
    // j is a property of the mesh: the three corners' ranks ride in fq[3].x (4 bits each, 15 = this corner's vertex has
    // more than 8 faces: the triple goes to G3 and the vertex walks the CSR).  k_fem scatters 3 x 12 bytes per face; a
    // vertex reads its eight triples in ONE round trip of loads that coalesce over the vertex lanes of a wave (through
    // va and G3 it was two dependent round trips and eight scattered gathers).  Layout: chunks of 32 vertices, in a
    // chunk eight planes by rank of 32 entries each (vf_entry): neighbouring faces write neighbouring entries of the
    // same plane, a vertex's eight entries are 384 bytes apart (immediate offsets of one address).  One 96-byte row per
    // vertex cost k_fem +6 us (a store instruction touching 64 rows), eight planes over the whole vertex range the
    // same +0.7 as this.  Entries past a vertex's valence hold zeros (written at every re-sort: adding -0 changes
    // nothing); VF_MARK in the first word of plane 0 = "walk the CSR".  A partitioned domain works the same way since
    // round 5: a face's three ranks are a property of the mesh, so they travel with the face (DP::fg[..].w); a face that
    // is not on this rank leaves a stale entry in its corner's row, which only a ghost vertex can have (its force is not
    // used) -- an OWNED vertex with a face missing is found by the re-sort that follows every migration (ERR_HALO).
    float* VF;
    float* f[3];           // vertices: internal force
    // topology (original ids)
    const int* idx_orig[3];  // face -> original particle ids of its corners
    const int* adj_off;      // vertex (original, 0-based) -> range in adj_fc
    const int* adj_fc;       // (original face id << 2) | corner
    int* imap;               // original particle id -> slot
    // rebuild scratch
    uint32_t* pkey;
    uint32_t* prank;
    uint32_t* src_of;      // sorted slot -> previous slot
    int* dst_of;           // previous slot -> sorted slot
    unsigned* tickets;     // 32 arrival counters, 128 bytes apart (k_rb_finish); next to each, a slot of the quiet-time
                           // minimum in the making (k_rb_count)
    uint32_t* halo_hdr[2]; // per launch: halo send buffers whose entry counters k_grid<0> resets (or null)
    // per launch: restrict k_grid<2> / k_g2p to the part of the grid that does (1) or does not (0)
    // depend on the halo exchange; -1 = everything.  Zones are x-block ranges [lo, hi].
    int halo_cls, halo_nz, halo_zlo[2], halo_zhi[2];
    // per launch of k_grid<0> in a chain substep: the zones whose raw node sums go straight into the send buffers
    // (what k_halo_pack2 does as a kernel of its own for the public mpm_halo_pack)
    int halo_pn, halo_plo[2], halo_phi[2], halo_pshift[2];
    unsigned halo_pcap;
    uint32_t* halo_pbuf[2];
    // ... and where their entry counters live: null = word 0 of the buffer itself (buffers in this device's memory: RCCL,
    // staged transports); the DIRECT transport packs into the NEIGHBOUR's memory and counts in a word of its own -- a
    // returning atomic on a peer's memory is not something every interconnect guarantees --, and
    // k_halo_signal stores the final count into the neighbour's header behind the kernel boundary
    uint32_t* halo_pcnt[2];
    int* cellcnt[2];       // [type][cell key]; zero outside a rebuild
    int* blkcnt[2];        // [type][block id]; zero outside a rebuild
    int* blkstart[2];
    int* lut_home;         // block id -> home slot or -1
    int* lut_act;          // block id -> active slot or -1
    unsigned* home_bits;   // bitmap of non-empty blocks, filled by k_rb_count, cleared by k_rb_tables
    // tables
    uint32_t* home_block;  // home slot -> block id
    int4* home_range;      // (face begin, face end, vertex begin, vertex end) slots
    int* home_nbr_act;     // [home][27] active slot of block + offset, or -1
    // Work items of the tile kernels (P2G, G2P): a home block, or a contiguous run of the wave
    // groups of a heavy one.  Every item has its own slab; k_grid sums all slabs of a block.
    int4* item_desc;       // [item] (home, first group, end group, 0)
    uint32_t* item_order;  // items, most groups first (static round-robin order)
    int4* item_flat;       // [position in item_order][2]: (item, home, home block id, groups), (first face slot, end face
                           // slot, first vertex slot, first group's index in home_groups) -- everything the tile kernels
                           // need to start an item in ONE load instead of a chain of four dependent ones
    int4* item_rng;        // [position in item_order] the item's own slots: (first face, end face, first vertex, end vertex)
                           // (from the wave groups, which k_rb_scatter lays out: written there, through item_pos)
    uint32_t* item_pos;    // [item] position in item_order
    int2* home_items;      // [home] (first item, item count)
    int* act_nbr_items;    // [active][27] first item | count << 24 of the neighbour home block, or -1
    int* home_ngroups;     // [home] number of P2G wave groups
    int4* home_groups;     // pool of (face begin, face end, vertex begin, vertex end) slot ranges, <= 64 particles each
    uint32_t* act_block;
    int* act_nbr_home;     // [active][27] home slot of block + offset, or -1
    // grid
    float4* slab;          // [home][TILE_N] (mvx, mvy, mvz, m) partial sums of one tile
    uint32_t* slab_mask;   // [item] 27-bit set of neighbour blocks reached by a stencil
    float4* gv;            // [active][64] (vx, vy, vz, m)  (momentum before the grid update)
    float4* gvs;           // [active][64] v* (velocity after the explicit update, before contact)
    const float4* dm_orig; // [original face id] Dm^-1 as computed by Finalize (a face that joins a rank needs it)
    Dist dist;
};

// idx-th active (or freshly appended) particle -> slot: faces first, then vertices
MPM_DEV int active_slot(const DP& p, int idx, int nf_in) { return idx < nf_in ? idx : p.Nf + (idx - nf_in); }

// Kernel ablation / cycle-counter switches exist only in the diagnostic build (-DMPM_DIAG=1,
// `python -m drake_amd._build --diag`); in the production build they fold to 0 and cost nothing.
#ifndef MPM_DIAG
#define MPM_DIAG 0
#endif
MPM_DEV int diag_flags(const DP& p) { return MPM_DIAG ? p.dbg : 0; }

// The deformation gradient of a face (row-major F[0..8]) in its three planes: fq[0] = (F2, F5, F8, F0), fq[1] = (F1, F3,
// F4, F6), f8 = F7.  The normal column F[:,2] comes first and together: it is the factor b of tau = a (x) b that
// k_p2g needs, which reads it from here instead of from a second copy (8 bytes per face less for k_fem to write).
MPM_DEV void pack_F(const float* F, float4& r0, float4& r1, float& r2) {
    r0 = make_float4(F[2], F[5], F[8], F[0]);
    r1 = make_float4(F[1], F[3], F[4], F[6]);
    r2 = F[7];
}
MPM_DEV void unpack_F(const float4& r0, const float4& r1, float r2, float* F) {
    F[2] = r0.x; F[5] = r0.y; F[8] = r0.z; F[0] = r0.w;
    F[1] = r1.x; F[3] = r1.y; F[4] = r1.z; F[6] = r1.w;
    F[7] = r2;
}

// A home block with n particles has ceil(n/64) wave groups; its list starts at this pool offset
// (blocks are laid out in slot order, so the offsets never overlap).
MPM_DEV size_t group_pool_offset(const DP& p, const int4& range, unsigned home) {
    return (size_t)((range.x + (range.z - p.Nf)) >> 6) + home;
}

MPM_DEV int off_index(int ox, int oy, int oz) { return (ox + 1) * 9 + (oy + 1) * 3 + (oz + 1); }

// id of the block at coords(b) + (ox,oy,oz), or -1 outside the grid
MPM_DEV int neighbor_block(uint32_t b, int o, int nb) {
    int bx, by, bz;
    block_coords(b, bx, by, bz);
    bx += o / 9 - 1;
    by += (o / 3) % 3 - 1;
    bz += o % 3 - 1;
    if (bx < 0 || by < 0 || bz < 0 || bx >= nb || by >= nb || bz >= nb) return -1;
    return (int)block_id((uint32_t)bx, (uint32_t)by, (uint32_t)bz);
}

// base cell of a position (float -> uint conversion saturates at 0 like the
// reference's CUDA cast, cuda_mpm_kernels.cuh:372-374)
// Workgroups are dealt to the 8 XCDs round robin (blockIdx % 8), each XCD with its own L2.  The
// streaming kernels gather mesh neighbours, which sit in nearby slots: give every XCD one
// contiguous eighth of the index range so that shared lines are fetched into one L2, not eight.
// Launch with a grid rounded up to a multiple of 8; chunks past the end are empty.
MPM_DEV unsigned xcd_chunk(unsigned b, unsigned grid) { return (b & 7u) * (grid >> 3) + (b >> 3); }
// The same when only the first n indices exist (a partitioned rank holds a fraction of the slots the launch was sized
// for): the eight XCDs share the ACTIVE range evenly -- with the plain mapping a rank that holds an eighth of the scene
// runs its whole FEM pass on one XCD.  Returns the 256-index chunk of workgroup b, or ~0u if it has none.  For n = the
// launch's full range (grid = chunks rounded up to a multiple of 8) this is xcd_chunk.
MPM_DEV unsigned xcd_chunk_active(unsigned b, unsigned n) {
    const unsigned per = (((n + 255u) >> 8) + 7u) >> 3;   // chunks per XCD
    const unsigned k = b >> 3;
    return k < per ? (b & 7u) * per + k : 0xFFFFFFFFu;
}

MPM_DEV bool in_halo_zone(const DP& p, int bx) {
    bool z = false;
    for (int k = 0; k < p.halo_nz; ++k) z |= bx >= p.halo_zlo[k] && bx <= p.halo_zhi[k];
    return z;
}
// a grid block is "boundary" if neighbours' sums are added to it; a work item if its tile (blocks
// bx-1 .. bx+1) touches such a block
MPM_DEV bool halo_block_selected(const DP& p, int bx) { return p.halo_cls < 0 || (int)in_halo_zone(p, bx) == p.halo_cls; }
MPM_DEV bool halo_item_selected(const DP& p, int bx) {
    return p.halo_cls < 0 || (int)(in_halo_zone(p, bx - 1) || in_halo_zone(p, bx) || in_halo_zone(p, bx + 1)) == p.halo_cls;
}

MPM_DEV uint32_t base_cell(float x, float dxinv) {
    const float t = x * dxinv - .5f;
    return t > 0.f ? (uint32_t)t : 0u;
}

}  // namespace mpm
