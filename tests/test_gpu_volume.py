"""Tiler / stitch / normaliser / predictor parity on the GPU (bit-exact for the copies and the select)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from mica_amd.synth import synth_af, synth_density
from oracle import volume_oracle as vo

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def eng(weights):
    from mica_amd.engine import Engine
    e = Engine(0, max_batch=2, tile_size=64)
    e.load_state_dict(weights)
    yield e
    e.close()


def test_gather_tiles_bit_exact_vs_oracle_and_golden(eng, golden_dir):
    tiler = json.load(open(os.path.join(golden_dir, "tiler.json")))
    for key, rec in tiler.items():
        shape = tuple(int(v) for v in key.split("x"))
        vol = synth_density(shape, rec["seed"])
        ref, idx = vo.tile_volume(vol, 48, 8)
        got = eng.gather_tiles(torch.from_numpy(vol).cuda(), 48, 8, 0, len(idx)).cpu().numpy()[:, 0]
        assert np.array_equal(got, ref) and sha(got) == rec["tiles_sha256"]


def test_gather_multichannel_ranges_and_other_tilings(eng):
    from mica_amd.engine import Engine
    vol = np.stack([synth_density((70, 33, 50), s) for s in (1, 2, 3)])
    e32 = Engine(0, max_batch=1, tile_size=64)
    for grid, pad in ((48, 8), (32, 16)):
        refs = [vo.tile_volume(v, grid, pad) for v in vol]
        T = len(refs[0][1])
        got = e32.gather_tiles(torch.from_numpy(vol).cuda(), grid, pad, 1, T - 1).cpu().numpy()
        for c in range(3):
            assert np.array_equal(got[:, c], refs[c][0][1:])
    e32.close()


def test_stitch_bit_exact_and_round_trip_full_size(eng):
    shape = (50, 70, 100)
    rec = np.stack([synth_density((12, 64, 64, 64), s) for s in (4, 5)], axis=1)    # [T=12, C=2, W,W,W]
    _, idx = vo.tile_volume(np.zeros(shape, np.float32), 48, 8)
    out = torch.zeros((2, *shape), device="cuda")
    eng.stitch_tiles(torch.from_numpy(rec).cuda(), out, 48, 8, 0)
    assert np.array_equal(out.cpu().numpy(), vo.stitch_volume(rec, idx, shape, 8))
    # size-independent property at a BASELINE-size map: gather -> stitch is the identity (256^3, 216 tiles)
    g = torch.Generator(device="cuda").manual_seed(1)
    vol = torch.rand((256, 256, 256), generator=g, device="cuda")
    back = torch.zeros_like(vol)
    T = int(eng.lib.mica_tile_count(256, 256, 256, 48))
    assert T == 216
    for first in range(0, T, 27):
        eng.stitch_tiles(eng.gather_tiles(vol, 48, 8, first, 27), back, 48, 8, first)
    assert torch.equal(back, vol)


def test_normalise_bit_exact_vs_golden(eng, golden_dir):
    norm = json.load(open(os.path.join(golden_dir, "normaliser.json")))
    for n in ("40", "64"):
        rec = norm[n]
        vol = (synth_density((int(n),) * 3, rec["seed"]) - 0.3) * 3.0
        t = torch.from_numpy(vol).cuda()
        med, pct = eng.normalise_map_(t)
        assert med == rec["median"] and pct == rec["percentile"]
        assert sha(t.cpu().numpy()) == rec["sha256"]
    # odd element count, +-inf, NaN (no spline prefilter on this entry point: nan_to_num semantics only)
    vol = (synth_density((11, 13, 9), 5) - 0.5)
    vol[0, 0, 0], vol[1, 1, 1], vol[2, 2, 2] = np.nan, np.inf, -np.inf
    x = np.nan_to_num(vol)
    med = np.median(x)
    m = (x > med) * (x - med)
    pct = np.percentile(m[m > 0], 99.9)
    ref = ((m < pct) * m + (m >= pct) * pct) / pct
    t = torch.from_numpy(vol).cuda()
    gm, gp = eng.normalise_map_(t)
    assert gm == float(med) and gp == float(pct) and np.array_equal(t.cpu().numpy(), ref.astype(np.float32))
    from mica_amd.engine import MicaHipError
    with pytest.raises(MicaHipError, match="No positive values"):
        eng.normalise_map_(torch.zeros((8, 8, 8), device="cuda"))


@pytest.mark.parametrize("case", ["f32_64", "f32_odd", "i16", "u16"])
def test_normalise_numpy_legacy_mode_vs_explicit_statement(eng, case):
    """MICA_NUMPY_LEGACY on the normaliser: the arithmetic of numpy 1.x (the reference's pinned 1.19.1) as written out in
    oracle/volume_oracle.py::normalise_map(numpy_legacy=True) - float64 percentile weights and weighted sum, float64 clip and
    division for a float32 map too.  Bit-exact against that statement; the default (numpy 2) mode stays what the reference run in
    this container produced, and the two modes differ by at most one float32 ulp.  PARITY UNPINNED for the legacy mode: no
    fixture of the reference covers numpy 1.x, the statement is restated from numpy 1.19's source."""
    rng = np.random.default_rng({"f32_64": 1, "f32_odd": 2, "i16": 3, "u16": 4}[case])
    if case.startswith("f32"):
        shape = (64, 64, 64) if case == "f32_64" else (31, 33, 35)
        vol = ((rng.random(shape, dtype=np.float32) - 0.3) * 3.0).astype(np.float32)
        data, mt = vol, 0
    else:
        shape = (24, 20, 28)
        data = rng.integers(-300, 4000, shape).astype(np.int16) if case == "i16" else rng.integers(0, 5000, shape).astype(np.uint16)
        vol, mt = data.astype(np.float32), 2 if case == "i16" else 3
    ref, med, pct = vo.normalise_map(data, numpy_legacy=True)
    t = torch.from_numpy(vol).cuda()
    gm, gp = eng.normalise_map_(t, map_type=mt, numpy_legacy=True)
    got = t.cpu().numpy()
    assert gm == med and gp == pct
    assert np.array_equal(got, ref)
    t2 = torch.from_numpy(vol).cuda()
    eng.normalise_map_(t2, map_type=mt)
    d = np.abs(t2.cpu().numpy().astype(np.float64) - got.astype(np.float64))
    assert d.max() <= np.spacing(np.float32(1.0))      # the two rule sets differ in the last bit at most


def test_normalise_large_map_properties(eng):
    """256^3: output in [0,1], zeros exactly where x <= median, idempotent ranks."""
    g = torch.Generator(device="cuda").manual_seed(2)
    vol = torch.randn((256, 256, 256), generator=g, device="cuda")
    ref_med = vol.flatten().sort().values[[256 ** 3 // 2 - 1, 256 ** 3 // 2]]
    x = vol.clone()
    med, pct = eng.normalise_map_(x)
    assert med == float((ref_med[0] + ref_med[1]) / 2)
    assert float(x.min()) == 0.0 and float(x.max()) == 1.0
    assert torch.equal(x == 0, vol <= med)
    assert abs(float((x == 1).float().mean()) - 0.5 * 1e-3) < 2e-5


def _scaled(got, ref):
    ref = ref.astype(np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), np.sqrt(np.mean(ref ** 2)))))


def test_predictor_end_to_end_vs_reference_golden(tmp_path, weights, golden_dir):
    """The reference's CryoEMPredictor.run_prediction on a 60x40x40 map (2 tiles) is the golden."""
    from mica_amd.predict import CryoEMPredictor
    g = np.load(os.path.join(golden_dir, "predictor_60x40x40.npz"))
    shape = tuple(int(v) for v in g["shape"])
    vol = synth_density(shape, int(g["seed"]))
    tiles, idx = vo.tile_volume(vol, 48, 8)
    gdir = tmp_path / "grids" / "normalized_map_grids"
    os.makedirs(gdir)
    for t, (i, j, k, di, dj, dk) in enumerate(idx):
        np.savez(gdir / f"normalized_map_grid_i{i}_j{j}_k{k}.npz", grid=tiles[t], i=i, j=j, k=k, di=di, dj=dj, dk=dk,
                 orig_shape=shape, grid_size=48, padding=8)
    ck = str(tmp_path / "ckpt.pth")
    torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in weights.items()}}, ck)
    pred = CryoEMPredictor(model_path=ck, grids_path=str(tmp_path / "grids") + "/", output_path=str(tmp_path / "out"),
                           save_output=True, device="cuda", quiet=True)
    ok, vols = pred.run_prediction()
    assert ok and set(vols) == {"backbone_probability", "carbon_alpha_probability", "amino_acid_prediction", "amino_acid_probability"}
    for k in vols:
        assert vols[k].dtype == np.float32
    assert vols["amino_acid_probability"].shape == (20, *shape) and vols["backbone_probability"].shape == shape
    assert np.abs(vols["backbone_probability"] - g["backbone_probability"]).max() < 1e-4
    assert np.abs(vols["carbon_alpha_probability"] - g["carbon_alpha_probability"]).max() < 1e-4
    sub = vols["amino_acid_probability"][:, ::2, ::2, ::2]
    assert np.abs(sub - g["amino_acid_probability_sub"]).max() < 1e-4
    top = np.sort(vols["amino_acid_probability"], axis=0)
    gap = top[-1] - top[-2]
    mism = vols["amino_acid_prediction"].astype(np.int64) != g["amino_acid_prediction"].astype(np.int64)
    assert not np.any(mism & (gap > 2e-4)) and mism.mean() < 1e-3
    assert os.path.exists(tmp_path / "out" / "results" / "grids" / "backbone_probability.npy")


def test_file_predictor_staged_uploads_over_many_runs(tmp_path, weights):
    """The file-based predictor uploads run r + 1 from one of two pinned staging pairs while run r computes; a pair is rewritten
    only after a host-side wait on the event of the upload that last read it (advisor, round 3).  Seven runs of two tiles each, every
    tile different (so a staging buffer rewritten too early would show), half of the tiles without atoms: equal to the disk-free
    pipeline on the same map bit for bit."""
    from mica_amd.af3_encoding import CHANNEL_NAMES
    from mica_amd.engine import Engine
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.predict import CryoEMPredictor
    shape = (330, 48, 90)                                   # 7 x 1 x 2 = 14 tiles
    vol = synth_density(shape, 71)
    af = synth_af(shape, 71, 2e-3)
    af[:, :, :, 45:] = 0
    tiles, idx = vo.tile_volume(vol, 48, 8)
    gdir = tmp_path / "grids" / "normalized_map_grids"
    os.makedirs(gdir)
    for t, (i, j, k, di, dj, dk) in enumerate(idx):
        np.savez(gdir / f"normalized_map_grid_i{i}_j{j}_k{k}.npz", grid=tiles[t], i=i, j=j, k=k, di=di, dj=dj, dk=dk,
                 orig_shape=shape, grid_size=48, padding=8)
    for c, name in enumerate(CHANNEL_NAMES):
        cdir = tmp_path / "grids" / "AF3_encoding_grids" / f"{name}_grids"
        os.makedirs(cdir)
        tl, _ = vo.tile_volume(af[c], 48, 8)
        for t, (i, j, k, di, dj, dk) in enumerate(idx):
            np.savez(cdir / f"{name}_grid_i{i}_j{j}_k{k}.npz", grid=tl[t], i=i, j=j, k=k, di=di, dj=dj, dk=dk, orig_shape=shape,
                     grid_size=48, padding=8)
    ck = str(tmp_path / "ckpt.pth")
    torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in weights.items()}}, ck)
    pred = CryoEMPredictor(model_path=ck, grids_path=str(tmp_path / "grids") + "/", output_path=str(tmp_path / "out"), save_output=False,
                           device="cuda", quiet=True, batch_size=2)
    ok, vols = pred.run_prediction()
    assert ok and pred.sample_count == 14
    e = Engine(0, max_batch=2, tile_size=64)
    e.load_state_dict(weights)
    mem = VolumePredictor(e, 48, 8, batch=2).predict_volume(torch.from_numpy(vol).cuda(), torch.from_numpy(af).cuda())
    for k in vols:
        assert np.array_equal(vols[k], mem[k].cpu().numpy()), k
    e.close()


def test_in_memory_pipeline_equals_tile_files_and_gating_is_per_tile(eng, weights):
    """VolumePredictor (no disk) on a map whose AF3 encodings touch only one of the two tiles."""
    from mica_amd.pipeline import VolumePredictor
    from oracle import model_oracle as mo
    shape = (60, 40, 40)
    vol = synth_density(shape, 41)
    af = np.zeros((24, *shape), np.float32)
    af[:, 56:60, 10:30, 10:30] = synth_af((4, 20, 20), 3, 0.05)       # tile 0's window ends at index 55: only tile 1 sees them
    vp = VolumePredictor(eng, 48, 8, batch=2)
    out = vp.predict_volume(torch.from_numpy(vol).cuda(), torch.from_numpy(af).cuda())
    tiles, idx = vo.tile_volume(vol, 48, 8)
    aft = np.stack([vo.tile_volume(a, 48, 8)[0] for a in af], axis=1)
    assert not aft[0].any() and aft[1].any()
    lb, lc, la = mo.mica_forward_per_tile(weights, torch.from_numpy(tiles[:, None]), torch.from_numpy(aft))
    pb, pc, pa, pp = mo.postprocess(lb, lc, la)
    ref_bb = vo.stitch_volume(pb.numpy(), idx, shape, 8)
    ref_aa = vo.stitch_volume(pa.numpy(), idx, shape, 8)
    assert np.abs(out["backbone_probability"].cpu().numpy() - ref_bb).max() < 1e-4
    assert np.abs(out["amino_acid_probability"].cpu().numpy() - ref_aa).max() < 1e-4
    assert out["amino_acid_prediction"].dtype == torch.float32


def test_gridcreator_and_preprocessor_mirrors(tmp_path, eng):
    from mica_amd import mrc
    from mica_amd.create_grids import GridCreator
    from mica_amd.preprocessing import DataPreprocessor
    raw = ((synth_density((40, 50, 60), 7) - 0.3) * 3.0).astype(np.float32)          # [nz,ny,nx]
    os.makedirs(tmp_path / "in" / "af3")
    mp = str(tmp_path / "in" / "emd.mrc")
    mrc.write_mrc(mp, raw, nxstart=5, nystart=6, nzstart=7)
    dp = DataPreprocessor(mp, str(tmp_path / "in" / "af3"), quiet=True)
    dp.resample_and_normalize_map()
    assert dp.normalized_map_path == str(tmp_path / "in" / "resampled_normalized_map.mrc")
    normed, hd = mrc.read_mrc(dp.normalized_map_path)
    ref, _, _ = vo.normalise_map(raw)
    assert np.array_equal(normed, ref) and (hd.nxstart, hd.nystart, hd.nzstart) == (5, 6, 7)
    gc = GridCreator(quiet=True)
    res = gc.create_normalized_map_grids(dp.normalized_map_path, str(tmp_path / "grids" / "normalized_map_grids"))
    vol, off = vo.transpose_axes(ref, 1, 2, 3, [7, 6, 5])
    tiles, idx = vo.tile_volume(vol, 48, 8)
    assert res["success"] and res["grid_count"] == len(idx) == 4 and res["offset"] == off == [5.0, 6.0, 7.0]
    assert gc.wait_for_files() == 0                                   # default "sync": nothing is left to wait for when the wrapper returns
    files = glob.glob(str(tmp_path / "grids" / "normalized_map_grids" / "*.npz"))
    assert len(files) == 4
    for t, (i, j, k, di, dj, dk) in enumerate(idx):
        d = np.load(str(tmp_path / "grids" / "normalized_map_grids" / f"normalized_map_grid_i{i}_j{j}_k{k}.npz"))
        assert np.array_equal(d["grid"], tiles[t]) and (int(d["di"]), int(d["dj"]), int(d["dk"])) == (di, dj, dk)
        assert tuple(d["orig_shape"]) == vol.shape and int(d["padding"]) == 8
    assert gc.create_normalized_map_grids(str(tmp_path / "missing.mrc"), str(tmp_path / "x"))["success"] is False


def test_mica_module_mirror(weights, golden_dir):
    from mica_amd.model import MICA
    g = np.load(os.path.join(golden_dir, "model_S8_af.npz"))
    m = MICA().to("cuda").eval()
    m.load_state_dict({"module." + k: v for k, v in weights.items()})
    x = torch.from_numpy(synth_density((1, 1, 8, 8, 8), int(g["seed"])))
    af = torch.from_numpy(synth_af((8, 8, 8), int(g["seed"]), float(g["afp"])))[None]
    bb, ca, aa = m(x, af)
    assert bb.shape == (1, 4, 8, 8, 8) and aa.shape == (1, 21, 8, 8, 8)
    assert _scaled(aa.cpu().numpy(), g["aa"]) < 1e-4


def test_streamed_maps_double_buffered_h2d(eng):
    """BASELINE configs[4] in miniature: independent maps streamed back to back equal one-at-a-time results."""
    from mica_amd.pipeline import VolumePredictor
    vp = VolumePredictor(eng, 48, 8, batch=2)
    maps = [synth_density((50, 40, 30), s) for s in (71, 72, 73)]
    afs = [None, synth_af((50, 40, 30), 72, 0.01), None]
    got = vp.predict_maps_streamed(maps, afs)
    assert len(got) == 3
    for m, a, g in zip(maps, afs, got):
        ref = vp.predict_volume(torch.from_numpy(m).cuda(), None if a is None else torch.from_numpy(a).cuda())
        for k in ref:
            assert g[k].dtype == np.float32 and np.array_equal(g[k], ref[k].cpu().numpy()), k


def test_sharded_single_rank_equals_plain(eng):
    """predict_volume_sharded with one rank (no process group) is the plain pipeline; the 2-rank rendezvous
    itself is covered by the gloo tests."""
    from mica_amd.pipeline import VolumePredictor
    vp = VolumePredictor(eng, 48, 8, batch=2)
    vol = torch.from_numpy(synth_density((60, 50, 40), 81)).cuda()
    a = vp.predict_volume(vol)
    b = vp.predict_volume_sharded(vol)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("shape,factors", [((6, 7, 5), (1.5, 1.25, 0.8)), ((40, 33, 27), (0.83, 0.83, 0.83)), ((24, 24, 24), (1.0, 1.0, 1.0)),
                                           ((4, 1, 8), (47.0, 1.0, 0.5)), ((31, 50, 18), (1.07, 1.318, 1.3)), ((64, 64, 64), (1.37, 1.37, 1.37))])
def test_zoom_cubic_bit_exact_vs_scipy(eng, shape, factors):
    """mica_zoom_cubic == scipy.ndimage.zoom(order=3), the reference's resampler (preprocessing.py:117), bit for bit."""
    from scipy.ndimage import zoom
    x = ((synth_density(shape, 19) - 0.3) * 2.5).astype(np.float32)
    ref = zoom(x, factors, order=3)
    got = eng.zoom_cubic(torch.from_numpy(x).cuda(), factors).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_preprocessor_with_anisotropic_voxels_and_nan(tmp_path, eng):
    """Full DataPreprocessor path with resampling: bit-exact against the oracle (scipy zoom + numpy normalise)."""
    from mica_amd import mrc
    from mica_amd.preprocessing import DataPreprocessor
    raw = ((synth_density((20, 24, 28), 63) - 0.3)).astype(np.float32)
    os.makedirs(tmp_path / "in" / "af3")
    mp = str(tmp_path / "in" / "emd.mrc")
    mrc.write_mrc(mp, raw, voxel_size=(1.5, 1.25, 0.8))
    dp = DataPreprocessor(mp, str(tmp_path / "in" / "af3"), quiet=True, engine=eng)
    dp.resample_and_normalize_map()
    got, hd = mrc.read_mrc(dp.normalized_map_path)
    _, hd0 = mrc.read_mrc(mp)
    ref, _, _ = vo.normalise_map(raw, voxel_size=hd0.voxel_size)       # zoom factors in (x,y,z) order on axes 0,1,2 (:112-117)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    assert np.allclose(hd.voxel_size, (1.0, 1.0, 1.0))
    # the header statistics are float64 reductions on the GPU (preprocessing.py): the four float32 fields equal numpy's on the host copy
    d64 = got.astype(np.float64)
    assert (np.float32(hd.dmin), np.float32(hd.dmax), np.float32(hd.dmean), np.float32(hd.rms)) == \
        (np.float32(d64.min()), np.float32(d64.max()), np.float32(d64.mean()), np.float32(d64.std()))
    # a NaN voxel poisons the whole map through the recursive prefilter: the reference reports failure (:163-165)
    bad = raw.copy(); bad[3, 4, 5] = np.nan
    from mica_amd.engine import MicaHipError
    with pytest.raises(MicaHipError, match="No positive values"):
        dp.normalize_array(bad, (1.0, 1.0, 1.0))


def test_full_size_512_map_tiling_properties(eng):
    """BASELINE configs[2] size: 512^3, reference tiling -> 1331 tiles; gather -> stitch is the identity and the
    zero padding of the last window (pad_end = 64 - 512 % 48 = 48 voxels) is exact."""
    g = torch.Generator(device="cuda").manual_seed(3)
    vol = torch.rand((512, 512, 512), generator=g, device="cuda")
    T = int(eng.lib.mica_tile_count(512, 512, 512, 48))
    assert T == 1331
    back = torch.zeros_like(vol)
    for first in range(0, T, 121):
        tiles = eng.gather_tiles(vol, 48, 8, first, 121)
        eng.stitch_tiles(tiles, back, 48, 8, first)
        if first + 121 == T:                                  # last tile: i=j=k=480, di=dj=dk=32
            last = tiles[-1, 0]
            assert torch.equal(last[8:40, 8:40, 8:40], vol[480:, 480:, 480:])
            assert float(last[40:].abs().max()) == 0.0 and float(last[:, 40:].abs().max()) == 0.0
    assert torch.equal(back, vol)
    from mica_amd._cabi import tile_table
    tab = tile_table(512, 512, 512, 48)
    assert tab[-1].tolist() == [480, 480, 480, 32, 32, 32] and tab[0].tolist() == [0, 0, 0, 48, 48, 48]


def _random_atoms(n, shape, seed):
    from oracle import af3_oracle as ao
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    coords = (rng.random((n, 3), dtype=np.float32) * np.array([nx + 6, ny + 6, nz + 6], np.float32) - 3.0).astype(np.float32)
    coords[: n // 8] = np.floor(coords[: n // 8]) + 0.5                  # exact halves: round half to even
    names = [["CA", "N", "C", "O", "CB", "CG", "HA", "OXT"][i] for i in rng.integers(0, 8, n)]
    res = [(ao.AMINO_ACIDS + ["MSE", "UNK"])[i] for i in rng.integers(0, 22, n)]
    return coords, names, res


@pytest.mark.parametrize("shape,origin", [((20, 20, 20), (0.0, 0.0, 0.0)), ((33, 17, 25), (-3.25, 4.5, 1.125)), ((12, 40, 12), (100.5, -7.0, 2.0))])
def test_af3_rasteriser_bit_exact_vs_oracle(eng, shape, origin):
    """mica_rasterise_atoms against the numpy restatement of preprocessing.py:172-178,283-298 (parity unpinned against the
    reference itself: Bio/mrcfile absent).  Shapes with nz >= max index reach keep the reference from raising."""
    from mica_amd import af3_encoding as ae
    from oracle import af3_oracle as ao
    coords, names, res = _random_atoms(4000, shape, 5)
    coords = coords + np.array(origin, np.float32)
    nz, ny, nx = shape
    if nz != nx:        # keep x within nx and z within nz so that the reference's cross-axis clip never indexes out
        coords[:, 0] = np.clip(coords[:, 0], origin[0], origin[0] + min(nx, nz) - 1)
        coords[:, 2] = np.clip(coords[:, 2], origin[2], origin[2] + min(nx, nz) - 1)
    ref = ao.rasterise_atoms(coords, names, res, origin, shape)
    got = ae.rasterise(eng, coords, names, res, origin, shape).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got, ref) and ref.sum() > 1000


def test_af3_rasteriser_edge_cases(eng, tmp_path):
    from mica_amd import af3_encoding as ae, mrc
    from mica_amd.engine import MicaHipError
    from mica_amd.preprocessing import DataPreprocessor
    from oracle import af3_oracle as ao
    # NaN / inf / huge coordinates: numpy's float->int cast gives INT64_MIN, clipped to 0
    coords = np.array([[np.nan, 1, 1], [np.inf, 2, 2], [-np.inf, 3, 3], [3e38, 4, 4], [1e10, 5, 5], [-1e10, 6, 6]], np.float32)
    names, res = ["CA"] * 6, ["GLY"] * 6
    ref = ao.rasterise_atoms(coords, names, res, (0, 0, 0), (8, 8, 8))
    got = ae.rasterise(eng, coords, names, res, (0, 0, 0), (8, 8, 8)).cpu().numpy()
    assert np.array_equal(got, ref) and got[0, 1, 1, 0] == 1 and got[0, 5, 5, 7] == 1
    # no atoms -> zeros; atoms without any channel -> zeros
    assert ae.rasterise(eng, np.zeros((0, 3), np.float32), [], [], (0, 0, 0), (4, 4, 4)).sum().item() == 0
    assert ae.rasterise(eng, np.ones((3, 3), np.float32), ["CB"] * 3, ["UNK"] * 3, (0, 0, 0), (4, 4, 4)).sum().item() == 0
    # where the reference raises IndexError the library reports MICA_ERR_RANGE
    with pytest.raises(IndexError):
        ao.rasterise_atoms(np.array([[0, 0, 30]], np.float32), ["CA"], ["GLY"], (0, 0, 0), (4, 5, 40))
    with pytest.raises(MicaHipError):
        ae.rasterise(eng, np.array([[0, 0, 30]], np.float32), ["CA"], ["GLY"], (0, 0, 0), (4, 5, 40))
    # the DataPreprocessor mirror: 24 MRC files with the map's header, readable by the tiler mirror
    shape = (16, 16, 16)
    mrc.write_mrc(str(tmp_path / "resampled_normalized_map.mrc"), np.zeros(shape, np.float32), origin=(1.0, 2.0, 3.0), nxstart=4)
    pdb = tmp_path / "t_af3_docked.pdb"
    pdb.write_text("ATOM      1  CA  GLY A   1       6.000   7.000   8.000  1.00 20.00           C\n")
    (tmp_path / "af3").mkdir()
    dp = DataPreprocessor(str(tmp_path / "map.mrc"), str(tmp_path / "af3"), quiet=True, engine=eng)
    dp.normalized_map_path = str(tmp_path / "resampled_normalized_map.mrc")
    assert dp.create_AF3_encodings(str(pdb)) is True
    ca, hd = mrc.read_mrc(str(tmp_path / "AF3_encodings" / "CA_encoding.mrc"))
    gly, _ = mrc.read_mrc(str(tmp_path / "AF3_encodings" / "GLY_encoding.mrc"))
    assert ca[5, 5, 5] == 1 and ca.sum() == 1 and gly[5, 5, 5] == 1 and hd.origin == (1.0, 2.0, 3.0) and hd.nxstart == 4
    assert len(list((tmp_path / "AF3_encodings").glob("*_encoding.mrc"))) == 24
    assert dp.create_AF3_encodings(str(tmp_path / "missing.pdb")) is False


def test_cluster_front_end_point_kernels_bit_exact(eng):
    """The volume-touching steps of Solver.clustering (modeler.py:767, 780-800, 834-858) against the reference's own numpy
    statements (oracle/cluster_oracle.py; parity unpinned against the module itself: open3d & co. are absent)."""
    from oracle import cluster_oracle as co
    rng = np.random.default_rng(77)
    shape = (37, 50, 41)
    ca = rng.random(shape, dtype=np.float32) ** 4            # mostly small, a few percent above the threshold
    ca[0, 0, 0] = ca[-1, -1, -1] = 0.9                       # corner points: kept by the threshold, skipped by the refinement
    ca[5, 6, 7] = np.nan                                      # NaN > thr is False
    bb = rng.random(shape, dtype=np.float32)
    aa = rng.random((20, *shape), dtype=np.float32)
    aa /= aa.sum(0, keepdims=True)
    d_ca, d_bb, d_aa = (torch.from_numpy(a).cuda() for a in (ca, bb, aa))

    pts = co.threshold_points(ca, 0.3)                       # int64 [n,3], lexicographic
    idx = eng.threshold_points(d_ca, 0.3)
    lin = (pts[:, 0] * shape[1] + pts[:, 1]) * shape[2] + pts[:, 2]
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), lin) and 1000 < len(lin) < ca.size // 2
    assert eng.threshold_points(d_ca, 0.3, capacity=10).numel() == len(lin)      # too small a buffer: retried with the count
    assert eng.threshold_points(d_ca, 2.0).numel() == 0
    assert np.array_equal(eng.gather_values(d_bb, idx).cpu().numpy(), co.gather(bb, pts))
    assert np.array_equal(eng.gather_values(d_aa, idx).cpu().numpy(), aa[:, pts[:, 0], pts[:, 1], pts[:, 2]])

    cands = pts[rng.permutation(len(pts))[:400]]
    cands = np.concatenate([cands, [[0, 0, 0], [36, 49, 40], [0, 10, 10], [10, 49, 10], [3, 3, 40]]])
    rc, ra, kept = co.refine_candidates(ca, aa, cands)
    gc, ga, ok = eng.refine_candidates(d_ca, d_aa, torch.from_numpy(cands.astype(np.int32)).cuda())
    ok = ok.cpu().numpy()
    assert np.array_equal(np.nonzero(ok)[0], kept) and (~ok).sum() >= 5
    assert rc.dtype == np.float64 and ra.dtype == np.float32
    assert np.array_equal(gc.cpu().numpy()[ok], rc), np.abs(gc.cpu().numpy()[ok] - rc).max()
    assert np.array_equal(ga.cpu().numpy()[ok], ra)
    # the host helpers on a volumes dict
    from mica_amd import clustering as cl
    vols = {"carbon_alpha_probability": d_ca, "backbone_probability": d_bb, "amino_acid_probability": d_aa}
    p2, cav, bbv = cl.candidate_points(eng, vols, 0.3)
    assert np.array_equal(p2, pts) and np.array_equal(cav, co.gather(ca, pts)) and np.array_equal(bbv, co.gather(bb, pts))
    nc, na, kp = cl.refine(eng, vols, cands)
    assert np.array_equal(nc, rc) and np.array_equal(na, ra) and np.array_equal(kp, kept)
    assert np.array_equal(cl.gather_at(eng, d_bb, pts[:50]), co.gather(bb, pts[:50]))
    # a neighbourhood summing to zero: the reference divides 0/0 and carries the NaNs
    z = np.zeros(shape, np.float32)
    gc2, _, ok2 = eng.refine_candidates(torch.from_numpy(z).cuda(), d_aa, torch.tensor([[5, 5, 5]], dtype=torch.int32).cuda())
    assert bool(ok2[0]) and torch.isnan(gc2).all()


def test_threshold_points_full_size_volume(eng):
    """512^3 (134 M voxels): ascending order, count equal to torch's, every selected voxel above the threshold."""
    g = torch.Generator(device="cuda").manual_seed(5)
    v = torch.rand((512, 512, 512), generator=g, device="cuda")
    idx = eng.threshold_points(v, 0.999)
    assert idx.numel() == int((v > 0.999).sum().item()) and idx.numel() > 100000
    assert bool((idx[1:] > idx[:-1]).all()) and bool((v.view(-1)[idx] > 0.999).all())


def _blobs(shape, centres, rng, sigma=1.6):
    """A CA-like probability volume: Gaussian blobs at `centres` (float positions) plus a little noise."""
    g = np.indices(shape, dtype=np.float32)
    v = np.zeros(shape, np.float32)
    for c in centres:
        v += np.exp(-((g[0] - c[0]) ** 2 + (g[1] - c[1]) ** 2 + (g[2] - c[2]) ** 2) / (2 * sigma ** 2)).astype(np.float32)
    return np.clip(v * 0.9 + rng.random(shape, dtype=np.float32) * 0.05, 0, 1).astype(np.float32)


def _chain(n, start, rng):
    """A CA-trace-like chain: consecutive points 3.8 apart in random directions."""
    pts = [np.array(start, float)]
    while len(pts) < n:
        d = rng.normal(size=3)
        pts.append(pts[-1] + 3.8 * d / np.linalg.norm(d))
    return np.array(pts)


def test_cluster_scores_nms_and_neighbour_matrix_bit_exact(eng):
    """The rest of Solver.clustering (modeler.py:775-797 cluster scores, :799-831 sorted greedy NMS, :860-888 distance /
    neighbour / backbone-density matrix) against the reference's own numpy statements (oracle/cluster_oracle.py).  DBSCAN
    (open3d) is replaced by connected components of a coarse grid here: any integer labelling exercises the same code."""
    from mica_amd import clustering as cl
    from oracle import cluster_oracle as co
    rng = np.random.default_rng(123)
    shape = (72, 64, 80)
    chains = [np.clip(_chain(22, (20, 20, 20), rng), 4, 58), np.clip(_chain(15, (50, 40, 60), rng), 4, 58),
              np.clip(_chain(8, (12, 50, 70), rng), 4, 58), np.clip(_chain(1, (60, 8, 8), rng), 4, 58)]
    centres = np.concatenate(chains)
    ca = _blobs(shape, centres, rng)
    strong = np.concatenate(chains[:2])             # backbone density along the first two chains only: the third cluster scores
    bb = _blobs(shape, np.concatenate([strong, (strong[1:] + strong[:-1]) / 2]), rng, sigma=2.0)      # low, the fourth is tiny
    bb += 0.35 * _blobs(shape, chains[2], rng, sigma=2.0)
    aa = rng.random((20, *shape), dtype=np.float32)
    aa /= aa.sum(0, keepdims=True)
    vols = {"carbon_alpha_probability": torch.from_numpy(ca).cuda(), "backbone_probability": torch.from_numpy(bb).cuda(),
            "amino_acid_probability": torch.from_numpy(aa).cuda()}
    thr, radius = 0.3, 9
    pts, cav, bbv = cl.candidate_points(eng, vols, thr)
    assert np.array_equal(pts, co.threshold_points(ca, thr)) and 2000 < len(pts) < 60000
    # labels: nearest chain by coarse position (three clusters of very different size and score), a few noise points
    dist = np.stack([np.min(np.linalg.norm(pts[:, None, :] - c[None], axis=2), axis=1) for c in chains])
    labels = np.argmin(dist, axis=0).astype(np.int64)
    labels[np.min(dist, axis=0) > 4.0] = -1
    labels[::97] = -1
    ref_sum, ref_avg, ref_val = co.cluster_scores(bb, pts, labels)
    got_sum, got_avg, got_val = cl.cluster_scores(eng, bbv, labels)
    assert [float(v) for v in got_sum] == [float(v) for v in ref_sum], (got_sum, ref_sum)
    assert [float(v) for v in got_avg] == [float(v) for v in ref_avg]
    assert np.array_equal(got_val, ref_val) and 0 < ref_val.sum() < len(pts)
    assert len(ref_sum) == 4 and ref_avg[3] == 0 and 0 < ref_avg[2] < max(ref_avg) / 2      # every branch of :783-796
    # long segments: numpy's piecewise pairwise order (8192-element pieces) on 1 .. 300k values
    lens = [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 20000, 100001, 300007]
    vals = (rng.random(sum(lens), dtype=np.float32) ** 2).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    sums = eng.segment_sums(torch.from_numpy(vals).cuda(), torch.from_numpy(off).cuda()).cpu().numpy()
    assert np.array_equal(sums, np.array([np.sum(vals[off[i]:off[i + 1]]) for i in range(len(lens))], dtype=np.float32))

    # NMS on the valid clusters
    pred = co.sorted_pred_list(ca, pts, ref_val)
    ref_cands = np.array(co.nms(pred.copy(), thr, radius))
    got_cands = cl.nms(eng, cav, pts, got_val, shape, thr, radius)
    assert got_cands.shape == ref_cands.shape and np.array_equal(got_cands, ref_cands) and 10 < len(ref_cands) < 400
    # a non-integer radius and a threshold that cuts the sorted list
    ref2 = np.array(co.nms(pred.copy(), 0.6, 5.5))
    got2 = cl.nms(eng, cav, pts, got_val, shape, 0.6, 5.5)
    assert np.array_equal(got2, ref2) and len(ref2) > len(ref_cands) // 4

    # refinement, then distances / neighbour lists / neighbour matrix on the refined positions
    new_c, new_a, kept = cl.refine(eng, vols, got_cands)
    rc, ra, rk = co.refine_candidates(ca, aa, ref_cands)
    assert np.array_equal(new_c, rc) and np.array_equal(new_a, ra) and np.array_equal(kept, rk)
    rdis, rlists, rmat = co.neighbour_matrix(rc, bb)
    gdis, glists, gmat = cl.neighbours(eng, vols, new_c)
    assert gdis.dtype == np.float64 and np.array_equal(gdis, rdis)
    for gl, rl in zip(glists, rlists):
        assert len(gl) == len(rl) and all(np.array_equal(a, b) for a, b in zip(gl, rl))
    assert np.array_equal(gmat, rmat) and (rmat > 0).sum() > len(rc)
    # both promotion branches of the last sum occur: distance term zero (3.3 <= dis <= 4.3) and non-zero
    inband = (rdis >= 2) & (rdis <= 6)
    assert ((np.abs(rdis - 3.8) <= 0.5) & inband).any() and ((np.abs(rdis - 3.8) > 0.5) & inband).any()
    # the reference's own environment (numpy 1.19.1): float64 density sums, stated explicitly in the oracle
    _, _, rmat1 = co.neighbour_matrix(rc, bb, numpy_legacy=True)
    _, _, gmat1 = cl.neighbours(eng, vols, new_c, numpy_legacy=True)
    assert np.array_equal(gmat1, rmat1) and not np.array_equal(rmat1, rmat) and np.abs(rmat1 - rmat).max() < 1e-6


@pytest.mark.parametrize("n,radius,seed", [(6000, 9, 1), (6000, 2.5, 2), (300, 30, 3), (1, 9, 4)])
def test_nms_points_random_dense_clouds_vs_oracle(eng, n, radius, seed):
    """mica_nms_points against the reference's greedy loop (modeler.py:822-831) on dense random candidate clouds: long chains of
    'suppressed by a candidate that is itself suppressed' are what the round-based GPU formulation has to get right."""
    from oracle import cluster_oracle as co
    rng = np.random.default_rng(seed)
    shape = (40, 36, 44)
    lin = rng.choice(shape[0] * shape[1] * shape[2], size=n, replace=False)
    pts = np.stack(np.unravel_index(lin, shape), axis=1).astype(np.int64)
    score = rng.random(n)
    order = np.argsort(-score)
    pred = np.concatenate([score[order, None], pts[order].astype(np.float64)], axis=1)
    ref = np.array(co.nms(pred.copy(), 0.0, radius)).reshape(-1, 3)
    keep = eng.nms_points(torch.from_numpy(pts[order].astype(np.int32)).cuda(), shape, radius).cpu().numpy()
    got = pts[order][keep]
    assert np.array_equal(got, ref) and 0 < len(ref) <= n
    from mica_amd.engine import MicaHipError
    if n > 1:
        dup = pts[order].astype(np.int32).copy()
        dup[1] = dup[0]
        with pytest.raises(MicaHipError, match="one voxel"):
            eng.nms_points(torch.from_numpy(dup).cuda(), shape, radius)


def test_uint8_encoding_volume_equals_float32(eng):
    """The 24 AF3 channels are binary, so a map's encodings may be held as uint8 (a quarter of the memory on every rank of a
    sharded run): the gather converts on the way, tiles and the whole pipeline are bit-identical to the float32 volume."""
    from mica_amd.pipeline import VolumePredictor
    shape = (70, 50, 60)
    af = synth_af(shape, 5, 0.01)
    d32 = torch.from_numpy(af).cuda()
    d8 = d32.to(torch.uint8)
    T = int(eng.lib.mica_tile_count(*shape, 48))
    for grid, pad in ((48, 8), (32, 16)):
        Tg = int(eng.lib.mica_tile_count(*shape, grid))
        assert torch.equal(eng.gather_tiles(d8, grid, pad, 1, Tg - 1), eng.gather_tiles(d32, grid, pad, 1, Tg - 1))
    vol = torch.from_numpy(synth_density(shape, 6)).cuda()
    vp = VolumePredictor(eng, 48, 8, batch=2)
    a, b = vp.predict_volume(vol, d32), vp.predict_volume(vol, d8)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert T == 8


def test_the_ctypes_stub_printed_in_integration_md_runs_as_printed(weights):
    """INTEGRATION.md section 3 shows a non-Python host how to bind the C ABI; the block is executed here verbatim (with the tensors it
    names defined around it) and its outputs compared with the Engine's, so the document cannot drift from the library."""
    import re
    from mica_amd.engine import AF_PER_TILE, Engine
    from mica_amd.synth import synth_af, synth_density
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"A ctypes stub.*?```python\n(.*?)```", text, re.S)
    assert m, "the ctypes stub of section 3 is gone"
    code = m.group(1)
    assert 'C.CDLL("mica_amd/lib/libmica_hip.so")' in code
    code = code.replace('"mica_amd/lib/libmica_hip.so"', repr(os.path.join(root, "mica_amd", "lib", "libmica_hip.so")))
    S, B = 64, 2
    x = torch.from_numpy(synth_density((B, 1, S, S, S), 5)).cuda()
    af = torch.stack([torch.from_numpy(synth_af((S, S, S), 5, 0.01)), torch.zeros(24, S, S, S)]).cuda()
    ns = {"state_dict": {k: torch.from_numpy(v) for k, v in weights.items()}, "x": x, "af": af, "B": B,
          "bb_prob": torch.empty(B, S, S, S, device="cuda"), "ca_prob": torch.empty(B, S, S, S, device="cuda"),
          "aa_prob": torch.empty(B, 20, S, S, S, device="cuda"), "aa_pred": torch.empty(B, S, S, S, device="cuda"),
          "rec": torch.empty(B, 23, S, S, S, device="cuda")}
    exec(compile(code, "INTEGRATION.md#3", "exec"), ns)           # noqa: S102
    torch.cuda.synchronize()
    assert ns["rc"] == 0
    e = Engine(0, max_batch=8, tile_size=S)
    e.load_state_dict(weights)
    bb, ca, aa, pred = e.forward_tiles(x.view(B, S, S, S), af, af_mode=AF_PER_TILE)
    assert torch.equal(ns["bb_prob"], bb) and torch.equal(ns["ca_prob"], ca) and torch.equal(ns["aa_prob"], aa) and torch.equal(ns["aa_pred"], pred)
    rec = ns["rec"]
    assert torch.equal(rec[:, 0], bb) and torch.equal(rec[:, 1], ca) and torch.equal(rec[:, 2], pred) and torch.equal(rec[:, 3:], aa)
    e.close()


def test_the_disk_free_snippet_printed_in_integration_md_runs_as_printed(tmp_path, weights):
    """INTEGRATION.md section 2's disk-free replacement of Solver.getData / nnPred, executed verbatim on a small map written here."""
    import re
    import types
    from mica_amd import mrc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"Disk-free variant.*?```python\n(.*?)```", text, re.S)
    assert m, "the disk-free snippet of section 2 is gone"
    data = synth_density((50, 60, 70), 21)                               # [nz, ny, nx]
    mrc.write_mrc(str(tmp_path / "n.mrc"), data, nxstart=3, nystart=-4, nzstart=9)
    torch.save({"model_state_dict": {k: torch.from_numpy(v) for k, v in weights.items()}}, str(tmp_path / "m.pth"))
    ns = {"cfg": types.SimpleNamespace(model_path=str(tmp_path / "m.pth"), normalized_map_path=str(tmp_path / "n.mrc")),
          "self": types.SimpleNamespace(), "af_volume_or_None": None, "torch": torch}
    exec(compile(m.group(1), "INTEGRATION.md#2", "exec"), ns)             # noqa: S102
    vols, me = ns["vols"], ns["self"]
    assert me.offset == [3.0, -4.0, 9.0] and me.AAPred.shape == (70, 60, 50) and me.AAPred.dtype == np.float32
    assert set(vols) == {"amino_acid_prediction", "amino_acid_probability", "backbone_probability", "carbon_alpha_probability"}
    assert tuple(vols["amino_acid_probability"].shape) == (20, 70, 60, 50)
    assert float(vols["backbone_probability"].min()) >= 0.0 and float(vols["backbone_probability"].max()) <= 1.0
    assert np.array_equal(np.unique(me.AAPred), np.unique(me.AAPred).round()) and me.AAPred.min() >= 0 and me.AAPred.max() <= 19
    ns["eng"].close()


def test_handoff_modes_no_files_sync_files_and_foreign_encodings(tmp_path, weights):
    """mica_amd/handoff.py behind the reference's call sites (utils/modeler.py:684-706, 724-738): `GridCreator(write_files=False)` leaves
    no tile file and the predictor of this process still returns the volumes; `write_files="sync"` has every file on disk when the
    wrapper returns; a predictor told to read the files agrees bit for bit; and when the encodings' tile files under grids_path were
    written by somebody else (no resident volume for them), the predictor reads files instead of mixing the two sources."""
    from mica_amd import handoff, mrc
    from mica_amd.af3_encoding import CHANNEL_NAMES
    from mica_amd.create_grids import GridCreator
    from mica_amd.predict import CryoEMPredictor
    handoff.clear()
    shape = (60, 40, 40)                                                   # (x, y, z) after the tiler's transpose: two tiles
    vol = synth_density(shape, 41)
    af = np.zeros((24, *shape), np.float32)
    af[:, 50:60, 10:30, 10:30] = synth_af((10, 20, 20), 3, 0.05)           # only the second tile's central region sees atoms
    inp = tmp_path / "in"
    os.makedirs(inp / "AF3_encodings")
    mp = str(inp / "resampled_normalized_map.mrc")
    mrc.write_mrc(mp, np.ascontiguousarray(vol.transpose(2, 1, 0)))
    for c, name in enumerate(CHANNEL_NAMES):
        mrc.write_mrc(str(inp / "AF3_encodings" / f"{name}_encoding.mrc"), np.ascontiguousarray(af[c].transpose(2, 1, 0)))
    ck = str(tmp_path / "ckpt.pth")
    torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in weights.items()}}, ck)

    def chain(tag, write_files, resident):
        grids = str(tmp_path / tag)
        gc = GridCreator(quiet=True, write_files=write_files)
        r1 = gc.create_normalized_map_grids(mp, os.path.join(grids, "normalized_map_grids"))
        r2 = gc.create_AF3_encodings_grids(str(inp / "AF3_encodings"), os.path.join(grids, "AF3_encoding_grids"))
        assert r1["success"] and r1["grid_count"] == 2 and r2["success"] and r2["total_grids"] == 48
        files_now = len(glob.glob(os.path.join(grids, "*", "*.npz"))) + len(glob.glob(os.path.join(grids, "*", "*", "*.npz")))
        pred = CryoEMPredictor(ck, grids + "/", str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
        pred.use_resident_volumes = resident
        ok, vols = pred.run_prediction()
        assert ok and (pred.resident is not None) == resident
        files_after = len(glob.glob(os.path.join(grids, "*", "*.npz"))) + len(glob.glob(os.path.join(grids, "*", "*", "*.npz")))
        return vols, files_now, files_after, grids

    v_none, n0, n1, _ = chain("nofiles", False, True)
    assert n0 == 0 and n1 == 0
    v_sync, n0, n1, g_sync = chain("sync", "sync", False)
    assert n0 == 50 and n1 == 50                                           # all there when the wrapper returned
    v_bg, n0, n1, _ = chain("background", True, True)
    assert n1 == 50                                                        # the predictor returns after the background writer is done
    for k in v_none:
        assert np.array_equal(v_none[k], v_sync[k]) and np.array_equal(v_none[k], v_bg[k]), k
    assert float(np.abs(v_none["backbone_probability"][:48] - v_none["backbone_probability"][48:58].mean()).max()) > 0     # not a constant volume
    # foreign encodings: the map is resident, the encodings' tile files come from elsewhere -> the files are read for both
    grids = str(tmp_path / "foreign")
    gc = GridCreator(quiet=True, write_files=False)
    assert gc.create_normalized_map_grids(mp, os.path.join(grids, "normalized_map_grids"))["success"]
    import shutil
    shutil.copytree(os.path.join(g_sync, "AF3_encoding_grids"), os.path.join(grids, "AF3_encoding_grids"))
    pred = CryoEMPredictor(ck, grids + "/", str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    ok, vols = pred.run_prediction()
    assert pred.resident is None and (ok, vols) == (False, {})              # no map tile FILES exist (write_files=False): loud failure, nothing mixed
    gc2 = GridCreator(quiet=True, write_files="sync")
    assert gc2.create_normalized_map_grids(mp, os.path.join(grids, "normalized_map_grids"))["success"]
    pred = CryoEMPredictor(ck, grids + "/", str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    ok, vols = pred.run_prediction()
    assert ok and pred.resident is None
    for k in v_none:
        assert np.array_equal(v_none[k], vols[k]), k
    # a predictor that fails (no checkpoint) still returns only after the background writers: nnPred's caller deletes grids_path next
    grids = str(tmp_path / "failing")
    gc = GridCreator(quiet=True, write_files=True)
    assert gc.create_normalized_map_grids(mp, os.path.join(grids, "normalized_map_grids"))["success"]
    assert gc.create_AF3_encodings_grids(str(inp / "AF3_encodings"), os.path.join(grids, "AF3_encoding_grids"))["success"]
    pred = CryoEMPredictor(str(tmp_path / "no_such_checkpoint.pth"), grids + "/", str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    assert pred.run_prediction() == (False, {}) and pred.resident is not None
    assert len(glob.glob(os.path.join(grids, "*", "*.npz"))) + len(glob.glob(os.path.join(grids, "*", "*", "*.npz"))) == 50
    assert all(w.done() for w in gc._writers)
    handoff.clear()


def test_tile_writer_started_at_once_on_a_256_cube_cuts_the_finished_volume(tmp_path, eng):
    """Advisor finding of round 5: TileFileWriter gathers on a private stream of its feeder thread, while the volume it cuts is produced
    by asynchronous kernels on the caller's stream (the axis transpose of GridCreator._device_volume: 0.3 ms on a 256^3 map).  With
    `wait()` right behind `start()` (create_grids_from_mrc; write_files="sync") the first gather used to be ordered behind NOTHING and
    could cut a half-transposed volume.  The writer now waits for an event recorded on the caller's stream when it is started.  A
    256^3 map, transposed ([z,y,x] -> [x,y,z]), joined at once: every one of the 216 tile files against the numpy tiler; then the
    same through the wrapper in "sync" mode with the map handed over on the GPU (no file read between the producer and the cut)."""
    from mica_amd import handoff, mrc
    from mica_amd.create_grids import GridCreator
    from mica_amd.synth import synth_map_fast
    N = 256
    raw = synth_map_fast(N, 77)
    mp = str(tmp_path / "map.mrc")
    mrc.write_mrc(mp, raw, stats=(0.0, 1.0, 0.5, 0.3))
    vol, _ = vo.transpose_axes(raw, 1, 2, 3, [0, 0, 0])
    ref, idx = vo.tile_volume(np.ascontiguousarray(vol), 48, 8)
    want = {f"i{i}_j{j}_k{k}": sha(ref[t]) for t, (i, j, k, _, _, _) in enumerate(idx)}
    del ref

    def check(d, prefix):
        files = sorted(glob.glob(os.path.join(d, "*.npz")))
        assert len(files) == len(want) == 216 and not glob.glob(os.path.join(d, ".*"))      # no temporary name left behind
        for f in files:
            key = os.path.basename(f)[len(prefix) + 1:-4]
            assert sha(np.load(f)["grid"]) == want[key], f

    gc = GridCreator(quiet=True, engine=eng)
    for rep in range(2):                                   # the second pass finds the pinned staging buffers cached: the tightest timing
        d = str(tmp_path / f"direct{rep}")
        n, off = gc.create_grids_from_mrc(mp, d, file_prefix="grid")
        assert n == 216 and off == [0.0, 0.0, 0.0]
        check(d, "grid")
    # the wrapper, volume handed over on the GPU by the stage that "wrote" the MRC: register it as DataPreprocessor does
    t = torch.from_numpy(raw).cuda()
    _, hd = mrc.read_mrc(mp)
    handoff.register_file(mp, t, hd)
    d = str(tmp_path / "grids" / "normalized_map_grids")
    res = gc.create_normalized_map_grids(mp, d)
    assert res["success"] and res["grid_count"] == 216
    check(d, "normalized_map_grid")
    handoff.clear()


def test_a_foreign_reader_never_sees_a_truncated_tile_file(tmp_path, eng):
    """Background tile files appear under their final name only when complete (hidden temporary name + rename; advisor, round 5: they used
    to be opened with O_TRUNC under the final name and filled later).  A reader that is NOT one of the mirrors polls the directory while
    the writer runs and loads every `*.npz` the moment it shows up: each must be a whole tile file (all keys, a 64^3 grid, the CRCs of
    the ZIP container intact); what it can see is FEWER files than the final count, never a truncated one."""
    import threading
    import zipfile
    from mica_amd import handoff, mrc
    from mica_amd.create_grids import GridCreator
    raw = synth_density((150, 150, 150), 88)
    mp = str(tmp_path / "map.mrc")
    mrc.write_mrc(mp, raw)
    d = str(tmp_path / "grids" / "normalized_map_grids")
    seen, bad, stop = set(), [], threading.Event()

    def reader():
        while not stop.is_set():
            for f in glob.glob(os.path.join(d, "*.npz")):
                if f in seen:
                    continue
                seen.add(f)
                try:
                    assert zipfile.ZipFile(f).testzip() is None
                    z = np.load(f)
                    assert z["grid"].shape == (64, 64, 64) and {"i", "j", "k", "di", "dj", "dk", "orig_shape", "mapc"} <= set(z.files)
                except Exception as ex:                   # noqa: BLE001
                    bad.append((f, repr(ex)))
    th = threading.Thread(target=reader)
    th.start()
    try:
        gc = GridCreator(quiet=True, engine=eng, write_files="background")
        res = gc.create_normalized_map_grids(mp, d)
        assert res["success"] and res["grid_count"] == 64
        early = len(glob.glob(os.path.join(d, "*.npz")))            # whatever is there now is complete; usually not all 64 yet
        assert gc.wait_for_files() == 64
    finally:
        stop.set()
        th.join()
    assert not bad, bad[:3]
    final = glob.glob(os.path.join(d, "*.npz"))
    assert len(final) == 64 and early <= 64 and not [f for f in os.listdir(d) if f.endswith(".part")]
    for f in final:                                                 # the reader may have stopped before the last ones: check the rest now
        assert zipfile.ZipFile(f).testzip() is None
    handoff.clear()


def test_a_failing_map_leaves_no_helper_thread_behind(weights):
    """`predict_volume(to_host=True)` runs helper threads (pinned staging allocation, slab unpack).  A forward that fails half way through
    the map - here a non-finite tile, which the library refuses with MICA_ERR_RANGE - must end them: a left-over thread waiting on its
    queue would keep the interpreter from exiting."""
    import threading
    from mica_amd.engine import Engine, MicaHipError
    from mica_amd.pipeline import VolumePredictor
    e = Engine(0, max_batch=2, tile_size=64)
    e.load_state_dict(weights)
    vol = torch.from_numpy(synth_density((100, 60, 50), 5)).cuda()
    vol[90, 30, 30] = float("nan")                           # inside the last x slab: the first slabs are already on their way to the host
    with pytest.raises(MicaHipError):
        VolumePredictor(e, 48, 8, 2).predict_volume(vol, None, to_host=True)
    names = [t.name for t in threading.enumerate() if t.is_alive()]
    assert not any(n.startswith(("mica-slab", "mica-pinned")) for n in names), names
    vol[90, 30, 30] = 0.5
    out = VolumePredictor(e, 48, 8, 2).predict_volume(vol, None, to_host=True)      # and the engine still works
    assert isinstance(out["backbone_probability"], np.ndarray) and np.isfinite(out["amino_acid_probability"]).all()
    e.close()
