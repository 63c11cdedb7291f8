"""Multi-GPU layer: ranks tile the domain along x.

Every rank owns one engine (one GPU) with its own local grid.  Rank r's local block coordinate
bx corresponds to bx - pitch_blocks on rank r+1: the patches are `pitch_blocks` apart, the local
grids overlap around each cut.  Per substep the only exchange is the raw node sums (mass,
momentum) of the active blocks in `zone_blocks` layers either side of a cut, sent to the
neighbour in ITS block coordinates; both sides then run the same grid update on the shared
blocks, so no second exchange is needed.

Transport: torch.distributed point-to-point.  With the "nccl" backend (= RCCL over xGMI) the
buffers are device tensors and the engine runs on torch's current stream, so nothing
synchronises with the host.  With "gloo" (CPU tests, or two ranks sharing one GPU in the GPU
test) the buffers are staged through host memory.

`HaloChain` (weak scaling): every rank owns its own cloth patches in its own local frame; particles
never change owner, a scene must keep each rank's particles within `zone_blocks` layers of its
patch.  `DomainChain` (strong scaling): the ranks cut ONE domain into x slabs (global coordinates,
pitch 0); every rank was finalised with the whole scene and keeps its slab's particles plus ghost
copies next to the cuts (mpm_dist_init); particles migrate between neighbours every few substeps.
"""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist


def strong_geometry(bits: int, world: int):
    """The partition bench.py --scaling strong uses for `world` ranks on a (1 << bits)^3 grid: the cloth stack spans
    the x blocks nb/4 .. 3nb/4, cut into slabs of equal width (the outermost ranks extend to the walls).  The zone is
    two blocks deep where a slab is at least four blocks wide, else one; the ghost bands come from the mesh (fractional
    widths that leave the most room for drift, mpm_dist_init) and migrations follow the measured x velocities."""
    nb = (1 << bits) // 4
    lo, hi = nb // 4, 3 * nb // 4
    if (hi - lo) % world or (hi - lo) // world < 2:
        raise ValueError(f"the cloth's {hi - lo} blocks do not split evenly into {world} slabs of at least 2 blocks")
    width = (hi - lo) // world
    cuts = [0] + [lo + width * r for r in range(1, world)] + [nb]
    return dict(cuts=cuts, zone_blocks=2 if width >= 4 else 1, ghost_cells=0, ghost_margin_cells=0, migrate_every=0)


def migration_safety() -> float:
    """Share of the ranks' common quiet-time estimate after which they migrate again (as the native chain's)."""
    return min(1.0, max(0.05, float(os.environ.get("MPM_MIG_SAFETY", "0.5"))))


class HaloChain:
    def __init__(self, engine, rank: int, world: int, cut_lo_block: int, cut_hi_block: int, pitch_blocks: int,
                 zone_blocks: int = 2, capacity_blocks: int = 512, device: torch.device | None = None,
                 group=None, split: bool | None = None, backend: str | None = None):
        """cut_lo_block / cut_hi_block: local x block index of the first block at/after the left /
        right cut plane (e.g. patch x in [0.25, 0.75] on 128^3 -> 8 and 24).
        backend: name of the transport instead of the process group's ("local": the ranks live in one process and a
        LocalWorld moves their buffers)."""
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self.cap = capacity_blocks
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.zone_lo = (cut_lo_block - zone_blocks, cut_lo_block + zone_blocks - 1)
        self.zone_hi = (cut_hi_block - zone_blocks, cut_hi_block + zone_blocks - 1)
        self.pitch = pitch_blocks
        self._fast_args = None
        self._ops = None
        if backend is None:
            backend = dist.get_backend(group) if world > 1 else "none"
        self.backend = backend
        self.staged = backend != "nccl"
        # split = update and gather what does not depend on the neighbours while the exchange is in
        # flight (mpm_substep_mid_halo).  Pays with an asynchronous transport (RCCL); with the staged
        # one it only exercises the code path.
        self.split = (not self.staged) if split is None else split
        self.device = device if device is not None else torch.device("cpu")
        nbytes = engine.halo_buffer_bytes(capacity_blocks)
        mk = lambda: torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.send = {n: mk() for n in (self.left, self.right) if n is not None}
        self.recv = {n: mk() for n in (self.left, self.right) if n is not None}
        # byte offsets of the buffer layout (mpm_halo_buffer_bytes): header 16, ids at 16, data 16-byte aligned behind
        # `capacity` ids; 1 KiB of node sums per block
        self._data_off = (((4 + capacity_blocks) * 4 + 15) // 16) * 16
        self.bytes_sent = {}      # per neighbour, of the last staged exchange (tests)
        self.blocks_sent = {}

    # -- one exchange: pack -> send/recv -> add ------------------------------------
    def _zones(self):
        z = []
        if self.left is not None:   # my left zone, relabelled into the left neighbour's coordinates
            z.append((self.zone_lo[0], self.zone_lo[1], +self.pitch, self.left))
        if self.right is not None:
            z.append((self.zone_hi[0], self.zone_hi[1], -self.pitch, self.right))
        return z

    def _start_transfer(self):
        """Posts the exchange; returns the handles to wait for (none for the staged transport,
        which is synchronous)."""
        if self.staged:
            self._exchange_staged()
            return []
        if self._ops is None:   # the buffers never change: build the op list once
            self._ops = []
            for n in self.send:
                self._ops.append(dist.P2POp(dist.isend, self.send[n], n, group=self.group))
                self._ops.append(dist.P2POp(dist.irecv, self.recv[n], n, group=self.group))
        return dist.batch_isend_irecv(self._ops)

    def _transfer(self):
        for w in self._start_transfer():
            w.wait()   # stream-ordered: the current stream waits, the host does not

    def exchange(self):
        if self.world == 1:
            return
        e = self.e
        for lo, hi, shift, n in self._zones():
            e.halo_pack(lo, hi, shift, self.send[n].data_ptr(), self.cap)
        self._transfer()
        for n in self.recv:
            e.halo_add(self.recv[n].data_ptr(), self.cap)

    def _compact(self, buf):
        """header | ids[count] | sums[count]: the count-sized message of a packed buffer (16 + 4 count + 1024 count bytes)"""
        count = int(buf[:4].cpu().view(torch.int32)[0])
        assert 0 <= count <= self.cap, (count, self.cap)
        return count, torch.cat([buf[:16], buf[16:16 + 4 * count], buf[self._data_off:self._data_off + 1024 * count]]).cpu()

    def _expand(self, msg, buf):
        count = int(msg[:4].view(torch.int32)[0])
        assert msg.numel() == 16 + 1028 * count and count <= self.cap
        dev = msg.to(buf.device)
        buf[:16 + 4 * count] = dev[:16 + 4 * count]
        buf[self._data_off:self._data_off + 1024 * count] = dev[16 + 4 * count:]

    def _exchange_staged(self):
        """Through host memory (gloo): the host knows the counts here, so only the filled part of every buffer
        travels -- 16 + 4 count + 1024 count bytes -- behind an 8-byte size message."""
        cuda = self.device.type == "cuda"
        if cuda:
            torch.cuda.synchronize()
        msgs, sizes_out, sizes_in, reqs = {}, {}, {}, []
        for n in self.send:
            self.blocks_sent[n], msgs[n] = self._compact(self.send[n])
            self.bytes_sent[n] = msgs[n].numel()
            sizes_out[n] = torch.tensor([msgs[n].numel()], dtype=torch.int64)
            sizes_in[n] = torch.zeros(1, dtype=torch.int64)
            reqs.append(dist.isend(sizes_out[n], n, group=self.group))
            reqs.append(dist.irecv(sizes_in[n], n, group=self.group))
        for r in reqs:
            r.wait()
        reqs, inbox = [], {}
        for n in self.send:
            inbox[n] = torch.empty(int(sizes_in[n][0]), dtype=torch.uint8)
            reqs.append(dist.isend(msgs[n], n, group=self.group))
            reqs.append(dist.irecv(inbox[n], n, group=self.group))
        for r in reqs:
            r.wait()
        for n in self.recv:
            self._expand(inbox[n], self.recv[n])
        if cuda:
            torch.cuda.synchronize()

    # -- one substep of the whole chain ----------------------------------------------
    def substep(self, dt: float, mpm_bc: int = -1):
        e = self.e
        if self.world > 1 and hasattr(e, "substep_begin_halo"):
            # two host calls per substep: (re-sort, FEM, P2G, gather, packs) | exchange | (adds, update, G2P)
            if self._fast_args is None:
                zones = self._zones()
                self._fast_args = (e.halo_zone_args([z[:3] for z in zones], [self.send[z[3]].data_ptr() for z in zones]),
                                   e.halo_buffer_args([self.recv[n].data_ptr() for n in self.recv]))
            e.substep_begin_halo(dt, self._fast_args[0], self.cap)
            works = self._start_transfer()
            if self.split:
                # runs on the engine's stream while RCCL moves the buffers on its own
                e.substep_mid_halo(dt, mpm_bc)
            for w in works:
                w.wait()
            e.substep_end_halo(dt, mpm_bc, self._fast_args[1], self.cap)
            return
        e.substep_begin(dt)
        self.exchange()
        e.substep_end(dt, mpm_bc)

    def run_substeps(self, n: int, dt: float, mpm_bc: int = -1):
        for _ in range(n):
            self.substep(dt, mpm_bc)


def attach_engine_to_torch_stream(engine):
    """Run the engine on torch's current stream so that RCCL transfers and kernels are ordered."""
    engine.set_stream(torch.cuda.current_stream().cuda_stream)


class DomainChain(HaloChain):
    """One domain, `world` x slabs cut at the block indices `cuts` (world + 1 ascending values).

    Per substep the same exchange as HaloChain (shift 0: all ranks use global block coordinates).
    Every `migrate_every` substeps, before the substep, the ranks swap migration records:
    mpm_dist_migrate_pack -> send/recv with both neighbours -> mpm_dist_migrate_apply."""

    def __init__(self, engine, rank: int, world: int, cuts, zone_blocks: int = 2, ghost_cells: int = 2,
                 ghost_margin_cells: int = 2, capacity_blocks: int = 512, migrate_every: int = 4,
                 migrate_capacity: int = 8192, device: torch.device | None = None, group=None,
                 split: bool | None = None, backend: str | None = None, headroom: float | None = None,
                 partitioned: bool = False):
        """migrate_every = 0: adaptive -- the ranks migrate when a share (MPM_MIG_SAFETY, default half) of the time has
        passed in which, by their common ballistic estimate, no particle drifts further along x than the bands allow
        (mpm_dist_migration_quiet_time).  partitioned: the engine has been through mpm_dist_init already."""
        assert len(cuts) == world + 1 and all(a < b for a, b in zip(cuts, cuts[1:]))
        if not partitioned:
            kw = {} if headroom is None else {"headroom": headroom}
            engine.dist_init(rank, world, list(cuts), zone_blocks, ghost_cells, ghost_margin_cells, **kw)
        super().__init__(engine, rank, world, cut_lo_block=cuts[rank], cut_hi_block=cuts[rank + 1], pitch_blocks=0,
                         zone_blocks=zone_blocks, capacity_blocks=capacity_blocks, device=device, group=group,
                         split=split, backend=backend)
        self._cuts, self._zone_blocks = list(cuts), zone_blocks
        self.team = False
        self.migrate_every, self.mig_cap = int(migrate_every), int(migrate_capacity)
        self.steps = 0
        self.migrations = 0
        self.mig_budget, self.mig_elapsed = 0.0, 0.0   # adaptive cadence: seconds until the next migration / since the last
        self.last_dt = 0.0
        nbytes = engine.dist_migration_buffer_bytes(self.mig_cap)
        mk = lambda: torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        # pack always fills a left and a right buffer; a missing neighbour's stays local
        self.mig_send = {"l": mk(), "r": mk()}
        self.mig_recv = {n: mk() for n in (self.left, self.right) if n is not None}
        self._mig_ops = None

    def install_contact_transport(self, zone_capacity_blocks: int = 512):
        """Gives the engine what the distributed contact solve (mpm_update_contact on a partitioned
        domain) needs when there is no native chain: a neighbour exchange of device buffers and an
        all-reduce of a few doubles, both over this chain's process group (staged through the host)."""
        import numpy as np
        e = self.e   # copies go through the engine (its stream, its HIP runtime): nothing here opens libamdhip64

        def exchange(sl, sr, rl, rr, nbytes):
            reqs, hosts = [], []
            for n, src, dst in ((self.left, sl, rl), (self.right, sr, rr)):
                if n is None:
                    continue
                out = torch.empty(nbytes, dtype=torch.uint8)
                e.memcpy_d2h(out.data_ptr(), src, nbytes)
                inn = torch.empty(nbytes, dtype=torch.uint8)
                reqs.append(dist.isend(out, n, group=self.group))
                reqs.append(dist.irecv(inn, n, group=self.group))
                hosts.append((inn, dst, out))
            for r in reqs:
                r.wait()
            for inn, dst, _ in hosts:
                e.memcpy_h2d(dst, inn.data_ptr(), nbytes)

        def allreduce(values: "np.ndarray"):
            t = torch.from_numpy(values)
            dist.all_reduce(t, group=self.group)

        self.e.dist_set_transport(exchange, allreduce, zone_capacity_blocks)

    def enable_team(self, zone_capacity_blocks: int = 512):
        """Device-resident coupled substeps on this partition (mpm_team.h): the per-substep halo over the DIRECT transport
        and the contact solve's exchanges over the TEAM transport -- peer stores + sequence flags on the engine's stream,
        regions mapped through HIP IPC handles that travel over this chain's process group.  Every rank calls it; returns
        False (and leaves every rank on the host-driven transports) when any rank could not set it up."""
        e = self.e
        ok, why, hd, ht = True, "", None, None
        try:
            e.chain_init(None, self.rank, self.world, self._cuts[self.rank], self._cuts[self.rank + 1], 0, self._zone_blocks, self.cap)
            hd = e.chain_direct_prepare()
            ht, _ = e.team_prepare(zone_capacity_blocks)
        except Exception as ex:  # noqa: BLE001  (e.g. no fine-grained memory: all ranks must fall back together)
            ok, why = False, str(ex)
        got = [None] * self.world
        dist.all_gather_object(got, (ok, why, hd, ht), group=self.group)
        if all(g[0] for g in got):
            try:
                e.chain_direct_connect(got[self.left][2] if self.left is not None else None,
                                       got[self.right][2] if self.right is not None else None)
                e.team_connect([g[3] for g in got])
            except Exception as ex:  # noqa: BLE001
                ok, why = False, str(ex)
        else:
            ok = False
            why = "; ".join(g[1] for g in got if not g[0])
        flags = [None] * self.world
        dist.all_gather_object(flags, ok, group=self.group)
        self.team = all(flags)
        self.team_error = why
        if not self.team:
            try:
                e.chain_destroy()
            except Exception:  # noqa: BLE001
                pass
        return self.team

    def _substeps_until_migration(self, n: int, dt: float) -> int:
        """how many of the next n substeps run before a migration is due (0: one is due now)"""
        k = 0
        steps, elapsed = self.steps, self.mig_elapsed
        while k < n:
            self.last_dt = dt
            if self.world > 1:
                due = (steps > 0 and steps % self.migrate_every == 0) if self.migrate_every > 0 else not (elapsed + dt <= self.mig_budget)
                if due and k == 0:
                    return 0
                if due:
                    break
            k += 1
            steps += 1
            elapsed += dt
        return k

    def coupled_substeps(self, n: int, dt: float, colliders, friction_mu, stiffness, damping, mpm_bc=-1, exact_line_search=False,
                         max_newton_iterations=0):
        """n coupled substeps of the partition (deformable_driver.h:240-258 per substep, on every rank): migrations when they
        are due, the substeps between two migrations in ONE call of mpm_run_coupled_substeps (device resident: enable_team).
        -> this rank's per-substep results"""
        assert getattr(self, "team", False), "enable_team() first"
        out, left = [], n
        while left > 0:
            k = self._substeps_until_migration(left, dt)
            if k == 0:
                self.migrate()
                if self.migrate_every > 0:      # (the fixed cadence counts substeps: this one runs now)
                    k = min(left, self.migrate_every)
                else:
                    k = max(1, self._substeps_until_migration(left, dt))
            out += self.e.run_coupled_substeps(k, dt, colliders, friction_mu, stiffness, damping, mpm_bc, exact_line_search,
                                               max_newton_iterations)
            self.steps += k
            self.mig_elapsed += k * dt
            left -= k
        return out

    def migrate(self):
        if self.world == 1:
            return
        e = self.e
        e.dist_migrate_pack(self.mig_send["l"].data_ptr(), self.mig_send["r"].data_ptr(), self.mig_cap)
        pairs = [(self.left, self.mig_send["l"]), (self.right, self.mig_send["r"])]
        pairs = [(n, b) for n, b in pairs if n is not None]
        cuda = self.device.type == "cuda"
        if cuda:
            torch.cuda.synchronize()
        # Two rounds, like mpm_chain_substeps: the record counts first, then exactly the records that exist (the
        # buffers are sized for the worst case: 9 MB for 65536 records; a migration of a cloth at rest moves none).
        rec = e.dist_migration_buffer_bytes(1) - 16
        size_out = {n: 16 + rec * min(int(b[:4].cpu().view(torch.int32)[0]), self.mig_cap) for n, b in pairs}
        # The record counts travel over THIS CHAIN'S group, as tensors of the kind that group carries (with the
        # "nccl" backend they used to go as CPU tensors through the DEFAULT group, which only works when that one happens
        # to be gloo, as in bench.py; a program whose only group is NCCL -- the usual torchrun setup -- raised).  Device
        # tensors in one batch for RCCL, CPU tensors for gloo / the in-process transport.
        size_in = {}
        cdev = self.device if self.backend == "nccl" else torch.device("cpu")
        outs = {n: torch.tensor([size_out[n]], dtype=torch.int64, device=cdev) for n, _ in pairs}
        for n, _ in pairs:
            size_in[n] = torch.zeros(1, dtype=torch.int64, device=cdev)
        if self.backend == "nccl":
            ops = []
            for n, _ in pairs:
                ops.append(dist.P2POp(dist.isend, outs[n], n, group=self.group))
                ops.append(dist.P2POp(dist.irecv, size_in[n], n, group=self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        else:
            reqs = []
            for n, _ in pairs:
                reqs.append(dist.isend(outs[n], n, group=self.group))
                reqs.append(dist.irecv(size_in[n], n, group=self.group))
            for r in reqs:
                r.wait()
        size_in = {n: int(t.cpu()[0]) for n, t in size_in.items()}
        self.mig_bytes_sent = dict(size_out)
        if self.staged:
            reqs, hosts = [], {}
            for n, b in pairs:
                hosts[n] = torch.empty(size_in[n], dtype=torch.uint8)
                out = b[:size_out[n]]
                reqs.append(dist.isend(out.cpu() if cuda else out, n, group=self.group))
                reqs.append(dist.irecv(hosts[n], n, group=self.group))
            for r in reqs:
                r.wait()
            for n in hosts:
                self.mig_recv[n][:size_in[n]].copy_(hosts[n])
            if cuda:
                torch.cuda.synchronize()
        else:
            ops = []
            for n, b in pairs:
                ops.append(dist.P2POp(dist.isend, b[:size_out[n]], n, group=self.group))
                ops.append(dist.P2POp(dist.irecv, self.mig_recv[n][:size_in[n]], n, group=self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        e.dist_migrate_apply(self.mig_recv[self.left].data_ptr() if self.left is not None else None,
                             self.mig_recv[self.right].data_ptr() if self.right is not None else None, self.mig_cap)
        self.migrations += 1
        if self.migrate_every == 0:
            # the ranks agree on the smallest estimate (apply has just synchronised the stream: the read-back is cheap)
            # (over this chain's group as well: a device tensor for RCCL)
            t = torch.tensor([e.dist_migration_quiet_time()], dtype=torch.float64,
                             device=self.device if self.backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            self.set_quiet_time(float(t.item()))

    def set_quiet_time(self, t_all: float):
        """The next migration is due after a share of the ranks' common estimate -- but the interval at most doubles from
        one migration to the next (4 substeps after the first): the estimate is ballistic, and a change of the
        velocities must be seen after at most as long as they have been watched (as mpm_chain_substeps does)."""
        t_est = migration_safety() * (t_all if t_all >= 0.0 else 0.0)   # (NaN -> 0)
        self.mig_budget = min(t_est, max(4.0 * self.last_dt, 2.0 * self.mig_elapsed))
        self.mig_elapsed = 0.0
        if hasattr(self.e, "dist_retune"):   # (band widths for the migrations to come; the same on every rank)
            self.e.dist_retune(t_all, self.last_dt)

    def migration_due(self, dt: float) -> bool:
        self.last_dt = dt
        if self.world == 1:
            return False
        if self.migrate_every > 0:
            return self.steps > 0 and self.steps % self.migrate_every == 0
        return not (self.mig_elapsed + dt <= self.mig_budget)

    def substep(self, dt: float, mpm_bc: int = -1):
        if self.migration_due(dt):
            self.migrate()
        self.steps += 1
        self.mig_elapsed += dt
        super().substep(dt, mpm_bc)


class LocalWorld:
    """`world` ranks of ONE partitioned domain inside one process, every rank with its own engine on the same GPU: the
    kernels, the bookkeeping and the decisions are exactly a DomainChain's, only the transport is a device-to-device
    copy.  All engines run on torch's current stream, so the copies are ordered with their kernels without host
    synchronisation.  For tests and rehearsals of partitions with more ranks than a box has GPUs (a GPU box admits few
    processes per card): one process, any number of ranks."""

    def __init__(self, engines, cuts, zone_blocks: int = 2, ghost_cells: int = 0, ghost_margin_cells: int = 0,
                 capacity_blocks: int = 512, migrate_every: int = 0, migrate_capacity: int = 8192,
                 device: torch.device | None = None, headroom: float | None = None):
        self.world = len(engines)
        assert len(cuts) == self.world + 1
        device = device if device is not None else torch.device("cuda", 0)
        # one stream of its own for all engines and all copies (the engines' own streams are non-blocking: nothing on
        # torch's default stream is ordered with them, and a null stream handle means "the engine's own" to mpm_set_stream)
        self.stream = torch.cuda.Stream(device)
        self.chains = []
        with torch.cuda.stream(self.stream):
            for r, e in enumerate(engines):
                e.set_stream(self.stream.cuda_stream)
                self.chains.append(DomainChain(e, r, self.world, cuts, zone_blocks, ghost_cells, ghost_margin_cells,
                                               capacity_blocks, migrate_every, migrate_capacity, device=device,
                                               backend="local", headroom=headroom))
        self.stream.synchronize()
        self.migrations = 0

    def _move(self, send_of, recv_of):
        for c in self.chains:
            for n in (c.left, c.right):
                if n is not None:
                    recv_of(self.chains[n])[c.rank].copy_(send_of(c, n), non_blocking=True)

    def migrate(self):
        with torch.cuda.stream(self.stream):
            self._migrate()

    def substep(self, dt: float, mpm_bc: int = -1):
        with torch.cuda.stream(self.stream):
            self._substep(dt, mpm_bc)

    def _migrate(self):
        for c in self.chains:
            c.e.dist_migrate_pack(c.mig_send["l"].data_ptr(), c.mig_send["r"].data_ptr(), c.mig_cap)
        self._move(lambda c, n: c.mig_send["l" if n == c.left else "r"], lambda c: c.mig_recv)
        for c in self.chains:
            c.e.dist_migrate_apply(c.mig_recv[c.left].data_ptr() if c.left is not None else None,
                                   c.mig_recv[c.right].data_ptr() if c.right is not None else None, c.mig_cap)
            c.migrations += 1
        self.migrations += 1
        if self.chains[0].migrate_every == 0:
            t = min(c.e.dist_migration_quiet_time() for c in self.chains)
            for c in self.chains:
                c.set_quiet_time(t)

    def _substep(self, dt: float, mpm_bc: int = -1):
        if self.world > 1 and all([c.migration_due(dt) for c in self.chains]):
            self._migrate()
        for c in self.chains:
            c.steps += 1
            c.mig_elapsed += dt
            if c._fast_args is None:
                zones = c._zones()
                c._fast_args = (c.e.halo_zone_args([z[:3] for z in zones], [c.send[z[3]].data_ptr() for z in zones]),
                                c.e.halo_buffer_args([c.recv[n].data_ptr() for n in c.recv]))
            c.e.substep_begin_halo(dt, c._fast_args[0], c.cap)
        self._move(lambda c, n: c.send[n], lambda c: c.recv)
        for c in self.chains:
            c.e.substep_end_halo(dt, mpm_bc, c._fast_args[1], c.cap)

    def run_substeps(self, n: int, dt: float, mpm_bc: int = -1):
        for _ in range(n):
            self.substep(dt, mpm_bc)

    def enable_team(self, zone_capacity_blocks: int = 512):
        """The direct halo and the TEAM transport between the ranks of this process (regions named by pointer): what
        DomainChain.enable_team sets up between processes through IPC handles."""
        with torch.cuda.stream(self.stream):
            for c in self.chains:
                c.e.chain_init(None, c.rank, self.world, c._cuts[c.rank], c._cuts[c.rank + 1], 0, c._zone_blocks, c.cap)
                c.e.chain_direct_prepare()
            halo = [c.e.chain_direct_base() for c in self.chains]
            regions = []
            for c in self.chains:
                c.e.chain_direct_connect_local(halo[c.left] if c.left is not None else None, halo[c.right] if c.right is not None else None)
                regions.append(c.e.team_prepare(zone_capacity_blocks)[1])
            for c in self.chains:
                c.e.team_connect(None, [None if r == c.rank else regions[r] for r in range(self.world)])
                c.team = True
        self.stream.synchronize()

    def coupled_substeps(self, n: int, dt: float, colliders, friction_mu, stiffness, damping, mpm_bc=-1, exact_line_search=False,
                         max_newton_iterations=0):
        """n coupled substeps of the whole world (mpm_world_coupled_substeps: every phase enqueued for all ranks in turn on
        the one stream), migrations when they are due.  -> per rank the list of per-substep results"""
        from drake_amd import GpuMpm
        engines = [c.e for c in self.chains]
        out = [[] for _ in engines]
        left = n
        with torch.cuda.stream(self.stream):
            while left > 0:
                ks = [c._substeps_until_migration(left, dt) for c in self.chains]
                k = min(ks)
                if k == 0:
                    self._migrate()
                    if self.chains[0].migrate_every > 0:
                        k = min(left, self.chains[0].migrate_every)
                    else:
                        k = max(1, min(c._substeps_until_migration(left, dt) for c in self.chains))
                res = GpuMpm.world_coupled_substeps(engines, k, dt, colliders, friction_mu, stiffness, damping, mpm_bc,
                                                    exact_line_search, max_newton_iterations)
                for i, r in enumerate(res):
                    out[i] += r
                for c in self.chains:
                    c.steps += k
                    c.mig_elapsed += k * dt
                left -= k
        return out

    def sync(self):
        for c in self.chains:
            c.e.gpu_sync()
