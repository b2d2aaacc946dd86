"""Random-shape sweep of the bit-exact volume ops against the CPU oracle: tiler (gather), stitch round trip, cubic resample and
normaliser (reference utils/create_grids.py:129-176, utils/predict.py:439-512, utils/preprocessing.py:111-133 via oracle/volume_oracle.py).

The fixed-shape tests of test_gpu_volume.py pin these against goldens of the reference's own code; this sweep draws the shapes nobody
picked by hand - prime edges, one voxel, one over / under the tile edge, non-cubic boxes, tilings other than (48, 8), zoom factors that
shrink and stretch - and demands bit equality every time.  MICA_FUZZ_SECONDS sets the duration per family (default 4 s: a few dozen
cases in the regular run; profiles/r05_fuzz_volume.txt records a 120-s run)."""
import os
import time

import numpy as np
import pytest
import torch

from mica_amd.synth import synth_density
from oracle import volume_oracle as vo

pytestmark = pytest.mark.gpu
SECONDS = float(os.environ.get("MICA_FUZZ_SECONDS", "4"))
EDGES = [1, 2, 3, 5, 7, 8, 15, 16, 17, 31, 33, 47, 48, 49, 55, 56, 57, 63, 64, 65, 71, 96, 97, 100]


@pytest.fixture(scope="module")
def eng():
    from mica_amd.engine import Engine
    e = Engine(0, max_batch=1, tile_size=64)
    yield e
    e.close()


def _box(rng, max_vox):
    while True:
        d = tuple(int(rng.choice(EDGES)) for _ in range(3))
        if d[0] * d[1] * d[2] <= max_vox:
            return d


def test_sweep_gather_and_stitch(eng):
    """gather == the oracle's pad-and-slice tiler for every tile; stitch(gather(v)) == v; tilings (48, 8), (32, 16), (16, 8), (56, 4), (60, 2)."""
    from mica_amd.engine import Engine
    rng = np.random.default_rng(11)
    engines = {64: eng}
    t0, n = time.time(), 0
    try:
        while time.time() - t0 < SECONDS:
            grid, pad = [(48, 8), (32, 16), (16, 8), (56, 4), (60, 2), (48, 8)][int(rng.integers(0, 6))]
            W = grid + 2 * pad
            if W not in engines:
                engines[W] = Engine(0, max_batch=1, tile_size=W)
            e = engines[W]
            shape = _box(rng, 400000)
            vol = synth_density(shape, int(rng.integers(1, 1 << 30)))
            ref, idx = vo.tile_volume(vol, grid, pad)
            dv = torch.from_numpy(vol).cuda()
            got = e.gather_tiles(dv, grid, pad, 0, len(idx))
            assert np.array_equal(got.cpu().numpy()[:, 0], ref), (shape, grid, pad)
            back = torch.full_like(dv, -7.0)
            first = 0
            while first < len(idx):                              # stitched in ragged pieces
                cnt = int(min(len(idx) - first, rng.integers(1, 9)))
                e.stitch_tiles(got[first:first + cnt], back, grid, pad, first)
                first += cnt
            assert torch.equal(back, dv), (shape, grid, pad)
            n += 1
    finally:
        for W, e in engines.items():
            if W != 64:
                e.close()
    print(f"gather / stitch sweep: {n} random (shape, tiling) cases, all bit-exact")
    assert n > 0


def test_sweep_zoom_and_normalise(eng):
    """zoom_cubic == scipy.ndimage.zoom(order = 3) as the oracle restates it, and the normaliser (median of the positive voxels, 99.9th
    percentile, clip and divide) == numpy's arithmetic, bit for bit, on random boxes and voxel sizes."""
    from mica_amd.preprocessing import DataPreprocessor
    dp = DataPreprocessor("unused.mrc", "unused", quiet=True, engine=eng)
    rng = np.random.default_rng(12)
    t0, n = time.time(), 0
    while time.time() - t0 < SECONDS:
        shape = _box(rng, 60000)
        if min(shape) < 2:
            continue                                            # scipy's zoom of a one-voxel axis is a special case the reference never meets
        raw = (synth_density(shape, int(rng.integers(1, 1 << 30))) - np.float32(rng.uniform(0.1, 0.6))).astype(np.float32)
        voxel = tuple(float(rng.choice([0.5, 0.75, 0.8, 1.0, 1.0, 1.07, 1.25, 1.5, 2.0])) for _ in range(3))
        try:
            ref, med, pct = vo.normalise_map(raw, voxel_size=voxel)
        except Exception:
            continue                                            # e.g. no positive voxel after the shift: the reference fails too (tested elsewhere)
        got, gmed, gpct = dp.normalize_array(raw, voxel)
        assert got.shape == ref.shape and np.array_equal(got, ref), (shape, voxel)
        assert (gmed, gpct) == (med, pct), (shape, voxel)
        n += 1
    print(f"zoom + normalise sweep: {n} random (shape, voxel size) cases, all bit-exact")
    assert n > 0
