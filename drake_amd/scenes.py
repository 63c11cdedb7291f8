"""Synthetic scenes for parity tests and the benchmark.

Cloth is the only body type the reference's MPM path supports.  The generators
build cloth sheets exactly like the reference's demo/test set-up
(examples/multibody/deformable/mpm_cloth_shared.h:62-99, cuda_mpm_test.cc:36-55):
a res x res vertex lattice with two triangles per quad.
"""
from __future__ import annotations

import numpy as np


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed: int, n: int, stream: int) -> np.ndarray:
    """Deterministic uniform floats in [-1, 1): integer hash -> float32, identical on every platform."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(
            0x9E3779B1)
        bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
    return (bits.astype(np.float32) / np.float32(1 << 23) - np.float32(1.0)).astype(np.float32)


def sheet_indices(res: int) -> np.ndarray:
    """Two triangles per quad, in the reference's order (cuda_mpm_test.cc:43-55)."""
    i, j = np.meshgrid(np.arange(res - 1), np.arange(res - 1), indexing="ij")
    p = lambda a, b: a * res + b
    tri = np.stack([p(i, j), p(i + 1, j), p(i, j + 1), p(i + 1, j + 1), p(i, j + 1), p(i + 1, j)], -1)
    return tri.reshape(-1).astype(np.int32)


def cloth_sheet(res: int, side: float, z: float, center=(0.5, 0.5)):
    xs = (center[0] - 0.5 * side) + (side / res) * np.arange(res, dtype=np.float64)
    ys = (center[1] - 0.5 * side) + (side / res) * np.arange(res, dtype=np.float64)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    pos = np.stack([X.reshape(-1), Y.reshape(-1), np.full(res * res, z)], -1).astype(np.float32)
    return pos, sheet_indices(res)


def cloth_stack(layers: int, res: int, domain_bits: int, z0: float = 0.6, side: float = 0.5, seed: int = 1234,
                jitter: float = 0.05, vel_amp: float = 0.01, center=(0.5, 0.5)):
    """`layers` horizontal sheets of res x res vertices, sheet k at z0 + k*dx/2.

    Returns a list of (pos, vel, indices) tuples, one per sheet, ready for add_qr_cloth."""
    dx = 1.0 / (1 << domain_bits)
    spacing = side / res
    out = []
    for k in range(layers):
        pos, idx = cloth_sheet(res, side, z0 + k * dx * 0.5, center)
        n = pos.shape[0]
        j = np.stack([hash_uniform(seed, n, 6 * k + c) for c in range(3)], -1)
        v = np.stack([hash_uniform(seed, n, 6 * k + 3 + c) for c in range(3)], -1)
        pos = (pos + np.float32(jitter * spacing) * j).astype(np.float32)
        vel = (np.float32(vel_amp) * v).astype(np.float32)
        out.append((pos, vel, idx))
    return out


CONFIGS = {
    # name: (domain_bits, layers, res)  -- SURVEY.md section 8d / BASELINE.json configs
    "plumbing_64k": (6, 8, 52),       # Np = 63,248
    "cloth_1m": (7, 16, 145),         # Np = 999,952
    "cloth_125k": (7, 2, 145),        # Np = 124,994 (the share of one of 8 ranks)
    "cloth_250k": (7, 4, 145),        # Np = 249,988
    "cloth_500k": (7, 8, 145),        # Np = 499,976 (scaling probe)
    "cloth_2m": (7, 32, 145),         # Np = 1,999,904 (scaling probe)
    "cloth_8m": (8, 32, 290),         # Np = 8,036,544
    "cloth_4m": (8, 16, 290),         # Np = 4,018,272
}


def particle_count(layers: int, res: int) -> tuple:
    nv = layers * res * res
    nf = 2 * layers * (res - 1) ** 2
    return nv, nf, nv + nf


def populate(engine, sheets):
    for pos, vel, idx in sheets:
        engine.add_qr_cloth(pos, vel, idx)
    engine.finalize()
    return engine
