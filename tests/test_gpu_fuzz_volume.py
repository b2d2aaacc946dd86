"""Random-shape sweep of the bit-exact volume ops against the CPU oracle: tiler (gather), stitch round trip, cubic resample and
normaliser, AF3 rasteriser, the point kernels of the clustering front end (reference utils/create_grids.py:129-176, utils/predict.py:439-512,
utils/preprocessing.py:111-133, 172-178, 283-298, utils/modeler.py:767-858 via oracle/volume_oracle.py, af3_oracle.py, cluster_oracle.py).

The fixed-shape tests of test_gpu_volume.py pin these against goldens of the reference's own code; this sweep draws the shapes nobody
picked by hand - prime edges, one voxel, one over / under the tile edge, non-cubic boxes, tilings other than (48, 8), zoom factors that
shrink and stretch - and demands bit equality every time.  MICA_FUZZ_SECONDS sets the duration per family, MICA_FUZZ_SEED shifts the generators (default 4 s, seed 0: a few dozen
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
SEED = int(os.environ.get("MICA_FUZZ_SEED", "0"))           # added to every family's generator seed
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
    rng = np.random.default_rng(11 + SEED)
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
    rng = np.random.default_rng(12 + SEED)
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


def test_sweep_af3_rasteriser(eng):
    """mica_rasterise_atoms == the oracle's restatement of preprocessing.py:172-178, 283-298 on random boxes, origins, atom counts and
    atom clouds that hang over every face (atoms outside the box are dropped; exact halves round to even)."""
    from mica_amd import af3_encoding as ae
    from oracle import af3_oracle as ao
    rng = np.random.default_rng(13 + SEED)
    names_all = ["CA", "N", "C", "O", "CB", "CG", "HA", "OXT", "SD", "NZ"]
    res_all = ao.AMINO_ACIDS + ["MSE", "UNK"]
    t0, n = time.time(), 0
    while time.time() - t0 < SECONDS:
        e = int(rng.choice([4, 7, 16, 20, 33, 48, 64]))
        shape = (e, int(rng.choice([4, 9, 17, 40])), e) if rng.random() < 0.5 else (e, e, e)     # nz == nx: the reference's cross-axis clip stays in range
        nz, ny, nx = shape
        origin = tuple(float(v) for v in rng.choice([0.0, -3.25, 4.5, 1.125, 100.5, -7.0], size=3))
        na = int(rng.choice([1, 2, 17, 300, 4000]))
        coords = (rng.random((na, 3), dtype=np.float32) * np.array([nx + 6, ny + 6, nz + 6], np.float32) - 3.0).astype(np.float32)
        coords[: na // 8] = np.floor(coords[: na // 8]) + 0.5
        coords = (coords + np.array(origin, np.float32)).astype(np.float32)
        names = [names_all[i] for i in rng.integers(0, len(names_all), na)]
        res = [res_all[i] for i in rng.integers(0, len(res_all), na)]
        ref = ao.rasterise_atoms(coords, names, res, origin, shape)
        got = ae.rasterise(eng, coords, names, res, origin, shape).cpu().numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref), (shape, origin, na)
        n += 1
    print(f"AF3 rasteriser sweep: {n} random (box, origin, atoms) cases, all bit-exact")
    assert n > 0


def test_sweep_point_kernels(eng):
    """threshold -> gather -> sub-voxel refinement -> NMS (Solver.clustering, utils/modeler.py:767-858) against the reference's numpy statements
    (oracle/cluster_oracle.py) on random boxes, thresholds and radii."""
    from oracle import cluster_oracle as co
    rng = np.random.default_rng(14 + SEED)
    t0, n = time.time(), 0
    while time.time() - t0 < SECONDS:
        shape = _box(rng, 120000)
        if min(shape) < 3:
            continue                                            # no interior voxel: nothing to refine
        ca = rng.random(shape, dtype=np.float32) ** int(rng.integers(2, 6))
        bb = rng.random(shape, dtype=np.float32)
        aa = rng.random((20, *shape), dtype=np.float32)
        aa /= aa.sum(0, keepdims=True)
        thr = float(rng.choice([0.3, 0.5, 0.05, 0.9]))
        d_ca, d_bb, d_aa = (torch.from_numpy(a).cuda() for a in (ca, bb, aa))
        pts = co.threshold_points(ca, thr)
        idx = eng.threshold_points(d_ca, thr)
        lin = (pts[:, 0] * shape[1] + pts[:, 1]) * shape[2] + pts[:, 2]
        assert np.array_equal(idx.cpu().numpy(), lin), (shape, thr)
        if len(pts) == 0:
            n += 1
            continue
        assert np.array_equal(eng.gather_values(d_bb, idx).cpu().numpy(), co.gather(bb, pts))
        cands = pts[rng.permutation(len(pts))[:500]]
        rc, ra, kept = co.refine_candidates(ca, aa, cands)
        gc, ga, ok = eng.refine_candidates(d_ca, d_aa, torch.from_numpy(cands.astype(np.int32)).cuda())
        ok = ok.cpu().numpy()
        assert np.array_equal(np.nonzero(ok)[0], kept), (shape, thr)
        if len(kept):                                           # (the oracle returns shapeless empties when every candidate sits on a face)
            assert np.array_equal(gc.cpu().numpy()[ok], rc, equal_nan=True) and np.array_equal(ga.cpu().numpy()[ok], ra, equal_nan=True), (shape, thr)
        # greedy NMS over the candidates in score order
        score = rng.random(len(cands))
        order = np.argsort(-score)
        radius = float(rng.choice([2.5, 3.0, 9.0 ** 0.5, 9.0, 30.0]))
        pred = np.concatenate([score[order, None], cands[order].astype(np.float64)], axis=1)
        ref = np.array(co.nms(pred.copy(), 0.0, radius)).reshape(-1, 3)
        keep = eng.nms_points(torch.from_numpy(cands[order].astype(np.int32)).cuda(), shape, radius).cpu().numpy()
        assert np.array_equal(cands[order][keep], ref), (shape, thr, radius)
        n += 1
    print(f"point-kernel sweep: {n} random (box, threshold, radius) cases, all bit-exact")
    assert n > 0
