"""The halo chain (drake_amd/dist.py) over gloo on CPU with 2 and 3 ranks.  The engine is replaced
by a numpy stand-in that speaks the same halo buffer format, so this tests the neighbour
bookkeeping, the block relabelling and the send/recv pairing -- not the kernels."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 32  # blocks per axis (128^3 grid)


def _bid(bx, by, bz):  # any injective code works for the stand-in, as long as both sides agree
    return (bx * NB + by) * NB + bz


class FakeEngine:
    """Active blocks with 64x4 sums each; halo_pack / halo_add with the engine's buffer layout
    (uint32 count, ids at word 4, data at 16-byte aligned offset)."""

    def __init__(self, blocks):
        self.blocks = {b: v.copy() for b, v in blocks.items()}  # (bx,by,bz) -> float32[64,4]

    @staticmethod
    def halo_buffer_bytes(cap):
        return (((4 + cap) * 4 + 15) // 16) * 16 + cap * 64 * 16

    @staticmethod
    def _view(ptr, cap):
        n = FakeEngine.halo_buffer_bytes(cap)
        raw = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr))
        off = (((4 + cap) * 4 + 15) // 16) * 16
        return raw[:off].view(np.uint32), raw[off:].view(np.float32).reshape(cap, 64, 4)

    def halo_pack(self, lo, hi, shift, ptr, cap):
        words, data = self._view(ptr, cap)
        k = 0
        for (bx, by, bz), v in sorted(self.blocks.items()):
            if lo <= bx <= hi and 0 <= bx + shift < NB:
                words[4 + k] = _bid(bx + shift, by, bz)
                data[k] = v
                k += 1
        words[0] = k

    def halo_add(self, ptr, cap):
        words, data = self._view(ptr, cap)
        inv = {_bid(*b): b for b in self.blocks}
        for k in range(int(words[0])):
            b = inv.get(int(words[4 + k]))
            if b is not None:
                self.blocks[b] += data[k]


def _blocks_of(rank):
    """Rank's active blocks: patch spans local bx 8..23, active region 7..24 (one layer beyond)."""
    rng = np.random.default_rng(100 + rank)
    out = {}
    for bx in range(7, 25):
        for by in (10, 11):
            out[(bx, by, 16)] = rng.standard_normal((64, 4)).astype(np.float32)
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd.dist import HaloChain
    eng = FakeEngine(_blocks_of(rank))
    chain = HaloChain(eng, rank, world, cut_lo_block=8, cut_hi_block=24, pitch_blocks=16, zone_blocks=2,
                      capacity_blocks=64)
    chain.exchange()
    q.put((rank, {k: v for k, v in eng.blocks.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_chain_exchange_sums_shared_blocks(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    before = {r: _blocks_of(r) for r in range(world)}
    for r in range(world):
        for (bx, by, bz), v in got[r].items():
            want = before[r][(bx, by, bz)].copy()
            # the same physical block on the right neighbour is bx - 16 there, on the left one bx + 16
            if r + 1 < world and 22 <= bx <= 25 and (bx - 16, by, bz) in before[r + 1]:
                want += before[r + 1][(bx - 16, by, bz)]
            if r - 1 >= 0 and 6 <= bx <= 9 and (bx + 16, by, bz) in before[r - 1]:
                want += before[r - 1][(bx + 16, by, bz)]
            np.testing.assert_array_equal(v, want)
    # both sides of a cut agree on every shared block
    for r in range(world - 1):
        for bx in (23, 24):  # active on both sides (the neighbour holds 7..24)
            for by in (10, 11):
                np.testing.assert_array_equal(got[r][(bx, by, 16)], got[r + 1][(bx - 16, by, 16)])


# ---- DomainChain: one domain, particles that change hands --------------------------------------
REC = 112   # bytes per migration record (7 x 16, drake_amd/csrc/mpm_dist.h)


class FakeDomainEngine(FakeEngine):
    """1-D stand-in for a partitioned engine: particles are (gid, x in cells); rank r owns the cells
    [lo, hi).  Speaks the engine's migration buffer format (count at byte 0, records of 112 bytes
    from byte 16: original id, role, ..., x in the first float of the second 16-byte group)."""

    def __init__(self, n_particles, nb):
        super().__init__({})
        self.x = {g: 2.0 + g * (nb * 4 - 4.0) / n_particles for g in range(n_particles)}   # everybody has everything
        self.owned = set()
        self.calls = []

    def dist_init(self, rank, world, cuts, zone_blocks, ghost_cells, ghost_margin_cells):
        self.rank, self.world = rank, world
        self.lo = cuts[rank] * 4 if rank > 0 else -10 ** 9
        self.hi = cuts[rank + 1] * 4 if rank < world - 1 else 10 ** 9
        self.owned = {g for g, x in self.x.items() if self.lo <= x < self.hi}
        self.x = {g: self.x[g] for g in self.owned}

    @staticmethod
    def dist_migration_buffer_bytes(cap):
        return 16 + cap * REC

    def _buf(self, ptr, cap):
        raw = np.ctypeslib.as_array((ctypes.c_uint8 * (16 + cap * REC)).from_address(ptr))
        return raw[:4].view(np.uint32), raw[16:].reshape(cap, REC)

    def dist_migrate_pack(self, ptr_l, ptr_r, cap):
        self.calls.append("pack")
        for ptr, leaving in ((ptr_l, [g for g in self.owned if self.x[g] < self.lo]),
                             (ptr_r, [g for g in self.owned if self.x[g] >= self.hi])):
            cnt, recs = self._buf(ptr, cap)
            cnt[0] = len(leaving)
            for k, g in enumerate(sorted(leaving)):
                recs[k, :8].view(np.int32)[:] = (g, 1)
                recs[k, 16:20].view(np.float32)[0] = self.x[g]
                self.owned.discard(g)
                del self.x[g]

    def dist_migrate_apply(self, ptr_l, ptr_r, cap):
        self.calls.append("apply")
        for ptr in (ptr_l, ptr_r):
            if not ptr:
                continue
            cnt, recs = self._buf(ptr, cap)
            for k in range(int(cnt[0])):
                g, role = (int(v) for v in recs[k, :8].view(np.int32))
                assert role == 1 and g not in self.owned
                self.owned.add(g)
                self.x[g] = float(recs[k, 16:20].view(np.float32)[0])

    def halo_pack(self, lo, hi, shift, ptr, cap):
        assert shift == 0            # one domain: every rank uses global block coordinates
        self.calls.append(("halo", lo, hi))
        super().halo_pack(lo, hi, shift, ptr, cap)

    def substep_begin(self, dt):
        self.calls.append("begin")

    def substep_end(self, dt, bc):
        for g in self.x:
            self.x[g] += 0.7     # cells per substep, towards +x
        self.calls.append("end")


def _domain_worker(rank, world, cuts, port, q, migrate_every=3):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("MPM_MIG_SAFETY", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd.dist import DomainChain
    eng = FakeDomainEngine(300, cuts[-1])
    # adaptive cadence: every rank has its own estimate of how long the bands hold; the smallest one counts
    eng.dist_migration_quiet_time = lambda: (rank + 2) * 3.0e-3
    chain = DomainChain(eng, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=16,
                        migrate_every=migrate_every, migrate_capacity=128)
    start = set(eng.owned)
    for _ in range(12):
        chain.substep(1e-3, -1)
    q.put((rank, start, set(eng.owned), dict(eng.x), eng.calls, (chain.zone_lo, chain.zone_hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts,migrate_every", [(2, [0, 8, 16], 3), (3, [0, 6, 10, 16], 3), (3, [0, 6, 10, 16], 0)])
def test_domain_chain_migrates_particles_between_neighbours(world, cuts, migrate_every):
    """migrate_every = 0: the adaptive cadence -- a migration before the first substep (it yields the first estimate), then
    whenever half of the ranks' smallest estimate (6 ms here: rank 0's) has passed: before substeps 0, 3, 6, 9 of 1 ms."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + world + (os.getpid() % 200) + (40 if migrate_every == 0 else 0)
    procs = [ctx.Process(target=_domain_worker, args=(r, world, cuts, port, q, migrate_every)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every particle has exactly one owner before and after, and particles did move to the right
    for idx in (0, 1):
        allg = [g for r in range(world) for g in got[r][idx]]
        assert sorted(allg) == list(range(300))
    assert len(got[world - 1][1]) > len(got[world - 1][0]) and len(got[0][1]) < len(got[0][0])
    for r in range(world):
        start, owned, x, calls, zones = got[r]
        # ownership is current up to the last migration round (at most 2 substeps of 0.7 cells ago)
        lo = cuts[r] * 4 if r > 0 else -1e9
        hi = cuts[r + 1] * 4 if r < world - 1 else 1e9
        assert all(lo <= x[g] < hi + 3 * 0.7 for g in owned)
        # cadence: a migration round (pack, apply) before substeps 3, 6, 9 and no other (adaptive: also before substep 0)
        want = [3, 6, 9] if migrate_every else [0, 3, 6, 9]
        assert calls.count("pack") == len(want) and calls.count("apply") == len(want)
        begins = [i for i, c in enumerate(calls) if c == "begin"]
        packs = [i for i, c in enumerate(calls) if c == "pack"]
        assert [sum(1 for b in begins if b < pk) for pk in packs] == want
        # the zones straddle this rank's cuts in global block coordinates
        assert zones == ((cuts[r] - 2, cuts[r] + 1), (cuts[r + 1] - 2, cuts[r + 1] + 1))


def _nccl_shaped_worker(rank, world, cuts, port, q):
    """A DomainChain told that its transport is "nccl" must talk over ITS group only, never over the default group
    (an NCCL-only program has no gloo default group to fall back on).  RCCL itself needs one GPU per rank,
    so here the chain's group is a gloo subgroup standing in for it (device = cpu) and the DEFAULT group is poisoned:
    any collective or point-to-point call without an explicit group fails the test."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("MPM_MIG_SAFETY", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group(backend="gloo")
    from drake_amd import dist as mdist
    seen = []
    # (isend / irecv themselves cannot be wrapped: P2POp checks their identity; in this mode the chain uses them only
    # inside batch_isend_irecv, whose ops carry their group)
    for name in ("all_reduce", "batch_isend_irecv"):
        real = getattr(dist, name)

        def guarded(*a, _real=real, _name=name, **kw):
            if _name == "batch_isend_irecv":
                assert all(op.group is group for op in a[0]), "a point-to-point op outside the chain's group"
            else:
                assert kw.get("group") is group, f"dist.{_name} on the default group"
            seen.append(_name)
            return _real(*a, **kw)
        setattr(mdist.dist, name, guarded)
    eng = FakeDomainEngine(300, cuts[-1])
    eng.dist_migration_quiet_time = lambda: (rank + 2) * 3.0e-3
    chain = mdist.DomainChain(eng, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=16,
                              migrate_every=0, migrate_capacity=128, group=group, backend="nccl", split=False)
    assert not chain.staged
    for _ in range(8):
        chain.substep(1e-3, -1)
    q.put((rank, set(eng.owned), sorted(set(seen))))
    dist.barrier()
    dist.destroy_process_group()


def test_a_chain_on_an_nccl_group_never_touches_the_default_group():
    world, cuts = 2, [0, 8, 16]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 200)
    procs = [ctx.Process(target=_nccl_shaped_worker, args=(r, world, cuts, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(g for r in range(world) for g in got[r][0]) == list(range(300))
    for r in range(world):
        assert "all_reduce" in got[r][1] and "batch_isend_irecv" in got[r][1]


# ---- DomainChain.enable_team / coupled_substeps: the host logic of the device-resident coupled path ----------------------
class FakeTeamEngine(FakeDomainEngine):
    """Stand-in for the calls of the TEAM set-up and of mpm_run_coupled_substeps on a partitioned engine: what is checked
    here is the host logic around them -- all ranks or none switch to the team transport, the substeps between two
    migrations go into ONE call, the cadence is the chain's."""

    def __init__(self, n_particles, nb, fail_prepare=False):
        super().__init__(n_particles, nb)
        self.fail_prepare = fail_prepare

    def chain_init(self, uid, rank, world, cut_lo, cut_hi, pitch, zone, cap):
        assert uid is None and pitch == 0
        self.calls.append(("chain_init", cut_lo, cut_hi))

    def chain_direct_prepare(self):
        return bytes([self.rank]) * 64

    def team_prepare(self, zone_capacity_blocks):
        if self.fail_prepare:
            raise RuntimeError("no fine-grained memory on this rank")
        return bytes([100 + self.rank]) * 64, 0

    def chain_direct_connect(self, left, right):
        self.calls.append(("direct_connect", left[0] if left else None, right[0] if right else None))

    def team_connect(self, handles, local_bases=None):
        self.calls.append(("team_connect", tuple(h[0] for h in handles)))

    def chain_destroy(self):
        self.calls.append("chain_destroy")

    def run_coupled_substeps(self, k, dt, colliders, mu, stiffness, damping, mpm_bc, exact, max_iters):
        self.calls.append(("coupled", k))
        for _ in range(k):
            for g in self.x:
                self.x[g] += 0.7
        return [dict(iterations=1, contacts=0, nodes=0, residual=0.0, setup_reused=False) for _ in range(k)]


def _team_worker(rank, world, cuts, port, q, migrate_every, failing_rank):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("MPM_MIG_SAFETY", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from drake_amd.dist import DomainChain
    eng = FakeTeamEngine(300, cuts[-1], fail_prepare=rank == failing_rank)
    eng.dist_migration_quiet_time = lambda: (rank + 2) * 3.0e-3
    chain = DomainChain(eng, rank, world, cuts, zone_blocks=2, ghost_cells=2, ghost_margin_cells=2, capacity_blocks=16,
                        migrate_every=migrate_every, migrate_capacity=128)
    ok = chain.enable_team(64)
    res = []
    if ok:
        res += chain.coupled_substeps(5, 1e-3, [], 0.5, 1e5, 1e-3)
        res += chain.coupled_substeps(7, 1e-3, [], 0.5, 1e5, 1e-3)
    q.put((rank, ok, chain.team_error, set(eng.owned), eng.calls, len(res)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts,migrate_every,failing_rank", [(2, [0, 8, 16], 3, None), (3, [0, 6, 10, 16], 0, None),
                                                                     (3, [0, 6, 10, 16], 0, 1)])
def test_team_setup_is_all_or_none_and_coupled_substeps_batch_between_migrations(world, cuts, migrate_every, failing_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + world + (os.getpid() % 200) + (40 if migrate_every == 0 else 0) + (80 if failing_rank is not None else 0)
    procs = [ctx.Process(target=_team_worker, args=(r, world, cuts, port, q, migrate_every, failing_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=120)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if failing_rank is not None:
        # one rank could not set its region up: NO rank switches, every rank has released what it had prepared and names the reason
        for r in range(world):
            ok, why, owned, calls, n_res = got[r]
            assert not ok and "fine-grained" in why and "chain_destroy" in calls and n_res == 0
            assert not any(isinstance(c, tuple) and c[0] in ("team_connect", "coupled") for c in calls)
        return
    assert sorted(g for r in range(world) for g in got[r][2]) == list(range(300))   # one owner per particle
    for r in range(world):
        ok, why, owned, calls, n_res = got[r]
        assert ok and n_res == 12
        # every rank mapped its neighbours' halo regions and ALL ranks' team regions, by the handles that travelled
        dc = [c for c in calls if isinstance(c, tuple) and c[0] == "direct_connect"][0]
        assert dc[1:] == (r - 1 if r > 0 else None, r + 1 if r < world - 1 else None)
        tc = [c for c in calls if isinstance(c, tuple) and c[0] == "team_connect"][0]
        assert tc[1] == tuple(100 + k for k in range(world))
        # the 12 substeps in batches that end where a migration is due: a migration before substeps (0,) 3, 6, 9 as in the
        # contact-free chain (test above), and the call boundary at 5 splits a batch
        want = [3, 6, 9] if migrate_every else [0, 3, 6, 9]
        seq, done, packs_at = [], 0, []
        for c in calls:
            if c == "pack":
                packs_at.append(done)
            elif isinstance(c, tuple) and c[0] == "coupled":
                seq.append(c[1])
                done += c[1]
        assert done == 12 and packs_at == want, (packs_at, seq)
        assert seq == [3, 2, 1, 3, 3], seq
