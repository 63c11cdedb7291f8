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
