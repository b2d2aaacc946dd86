"""Deterministic synthetic inputs (density tiles/maps, AF3 one-hot encodings).

No dataset or sample map is reachable offline (reference README.md:27-39), so tests and
bench.py use these; both sides of every parity test regenerate them from the seed, so
only outputs are stored as fixtures.
"""
from __future__ import annotations

import numpy as np

from .weights import hash_uniform


def synth_density(shape, seed: int) -> np.ndarray:
    """float32 density in [0, 1) of the given shape (hash of (seed, index): host independent)."""
    n = int(np.prod(shape))
    u = hash_uniform("density", n, seed)
    return ((u + 1.0) * 0.5).astype(np.float32).reshape(shape)


def synth_af(shape, seed: int, p: float = 1e-3) -> np.ndarray:
    """float32 {0,1} encodings [24, *shape] with occupancy p (real atom rasters are ~1e-3,
    reference utils/preprocessing.py:288-298)."""
    n = 24 * int(np.prod(shape))
    u = hash_uniform("af3", n, seed)
    return ((u + 1.0) * 0.5 < p).astype(np.float32).reshape((24, *shape))


def synth_map_fast(n: int, seed: int) -> np.ndarray:
    """Large benchmark maps: default_rng(seed).random((n,n,n), float32)."""
    return np.random.default_rng(seed).random((n, n, n), dtype=np.float32)
