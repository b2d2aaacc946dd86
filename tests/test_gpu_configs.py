"""BASELINE.json configs exercised at size on the GPU: configs[0] as a chained plumbing run on a synthetic MRC + PDB
(the EMD-15635 sample is a download), configs[2] in miniature (two real ranks on one GPU), configs[4] at full size
(4 maps of 384^3 streamed), the reference's batching mode, and the tiler wrappers that had no test (rows a5, a6)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mica_amd.synth import synth_af, synth_density
from oracle import volume_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng(weights):
    from mica_amd.engine import Engine
    e = Engine(0, max_batch=8, tile_size=64)
    e.load_state_dict(weights)
    yield e
    e.close()


def _pdb_line(serial, name, resname, resseq, x, y, z):
    nm = name if len(name) == 4 else " " + name.ljust(3)
    return "%-6s%5d %4s%1s%3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % (
        "ATOM", serial, nm, " ", resname, "A", resseq, x, y, z, 1.0, 20.0, name[0])


def _save_ckpt(path, weights):
    torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in weights.items()}}, path)


def test_config0_chained_plumbing_mrc_pdb_to_volumes(tmp_path, eng, weights):
    """Solver.getData + Solver.nnPred (utils/modeler.py:673-738) on a synthetic map and docked model: MRC ->
    DataPreprocessor.resample_and_normalize_map -> create_AF3_encodings -> both GridCreator wrappers -> file-based
    CryoEMPredictor (tiles with and without atoms) -> the four volumes; equal to the disk-free VolumePredictor bit for
    bit, and to the CPU oracle on one tile of each kind."""
    from mica_amd import mrc
    from mica_amd.af3_encoding import CHANNEL_NAMES
    from mica_amd.create_grids import GridCreator
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.predict import CryoEMPredictor
    from mica_amd.preprocessing import DataPreprocessor
    from oracle import af3_oracle as ao
    from oracle import model_oracle as mo

    raw = ((synth_density((40, 50, 60), 17) - 0.3) * 3.0).astype(np.float32)          # [nz, ny, nx]
    inp = tmp_path / "input" / "9999"
    os.makedirs(inp / "AF3_results")
    map_path = str(inp / "emd_9999.mrc")
    mrc.write_mrc(map_path, raw, origin=(2.0, -3.0, 1.5), nxstart=5, nystart=6, nzstart=7)
    rng = np.random.default_rng(5)
    names, res = ["N", "CA", "C", "O", "CB"], ao.AMINO_ACIDS
    lines, coords, anames, ares = [], [], [], []
    for r in range(60):                                   # atoms only at x < 30 (+origin): tiles with i = 48 see none
        c0 = rng.random(3) * np.array([24.0, 40.0, 30.0]) + np.array([4.0, 0.0, 3.0])
        for a in names:
            xyz = np.round(c0 + rng.random(3) * 2.0, 3)
            lines.append(_pdb_line(len(lines) + 1, a, res[r % 20], r + 1, *xyz))
            coords.append(xyz); anames.append(a); ares.append(res[r % 20])
    pdb = inp / "9999_af3_docked.pdb"                      # modeler.py:681
    pdb.write_text("".join(lines) + "END\n")

    # --- getData ------------------------------------------------------------------------------------------------------
    # write_files="background": the normalised map and the 24 encoding MRCs are written behind the call, the volumes go to GridCreator
    # through the registry of mica_amd/handoff.py (the default, "sync", is exercised by every other DataPreprocessor test)
    dp = DataPreprocessor(map_path=map_path, AF3_results=str(inp / "AF3_results"), quiet=True, engine=eng, write_files="background")
    dp.resample_and_normalize_map()
    assert dp.normalized_map_path == str(inp / "resampled_normalized_map.mrc")
    assert dp.create_AF3_encodings(str(pdb)) is True
    gc = GridCreator(quiet=True, engine=eng, write_files="background")         # the Solver-flow setting (mica_amd/solver_mirrors.py)
    grids = str(tmp_path / "grids")
    res_map = gc.create_normalized_map_grids(normalized_map_path=dp.normalized_map_path, output_dir=os.path.join(grids, "normalized_map_grids"))
    res_af = gc.create_AF3_encodings_grids(AF3_encodings_path=str(inp / "AF3_encodings"), output_dir=os.path.join(grids, "AF3_encoding_grids"),
                                           parallel=True)
    # the wrappers return when the volumes are resident and registered; the tile files come from a background writer
    assert gc.wait_for_files() == 4 + 96
    norm_ref, _, _ = vo.normalise_map(raw)
    vol_ref, off = vo.transpose_axes(norm_ref, 1, 2, 3, [7, 6, 5])
    enc_ref = ao.rasterise_atoms(np.array(coords, np.float32), anames, ares, (2.0, -3.0, 1.5), raw.shape)
    assert res_map["success"] and res_map["grid_count"] == 4 and res_map["offset"] == off == [5.0, 6.0, 7.0]
    # row a5: create_AF3_encodings_grids (utils/create_grids.py:269-397)
    assert res_af["success"] and res_af["successful_channels"] == 24 and res_af["failed_channels"] == 0
    assert res_af["total_channels"] == 24 and res_af["total_grids"] == 96 and res_af["processing_errors"] == []
    tiles_ref, idx = vo.tile_volume(vol_ref, 48, 8)
    af_tiles_ref = np.zeros((4, 24, 64, 64, 64), np.float32)
    for c, name in enumerate(CHANNEL_NAMES):
        ev, _ = vo.transpose_axes(enc_ref[c], 1, 2, 3, [7, 6, 5])
        tl, _ = vo.tile_volume(ev, 48, 8)
        af_tiles_ref[:, c] = tl
        for t, (i, j, k, di, dj, dk) in enumerate(idx):
            d = np.load(os.path.join(grids, "AF3_encoding_grids", f"{name}_grids", f"{name}_grid_i{i}_j{j}_k{k}.npz"))
            assert np.array_equal(d["grid"], tl[t]) and (int(d["di"]), int(d["dj"]), int(d["dk"])) == (di, dj, dk)
            assert tuple(d["orig_shape"]) == (60, 50, 40) and int(d["grid_size"]) == 48 and int(d["mapc"]) == 1
    has_atoms = af_tiles_ref.reshape(4, -1).any(axis=1)
    assert has_atoms.tolist() == [True, True, False, False] and enc_ref.sum() > 300

    # --- nnPred -------------------------------------------------------------------------------------------------------
    ck = str(tmp_path / "ckpt.pth")
    _save_ckpt(ck, weights)
    pred = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    ok, vols = pred.run_prediction()
    assert ok and vols["amino_acid_probability"].shape == (20, 60, 50, 40)
    assert pred.sample_count == 4 and pred.use_optimized_batching is False
    # the predictor took the volumes GridCreator left on the GPU (mica_amd/handoff.py) - the encodings as uint8 -, not the files
    assert pred.resident is not None and pred.resident[1] is not None and pred.resident[1].dtype == torch.uint8
    # ... and the same class reading the FILES (what a predictor in another process does) returns the same volumes bit for bit
    cold = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    cold.use_resident_volumes = False
    ok2, vols2 = cold.run_prediction()
    assert ok2 and cold.resident is None
    for k in vols:
        assert np.array_equal(vols[k], vols2[k]), k
    # ... and so does a predictor in a FRESH process, which has nothing but the files (the cold path)
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, sys.argv[1]);"
            "from mica_amd.predict import CryoEMPredictor;"
            "p = CryoEMPredictor(model_path=sys.argv[2], grids_path=sys.argv[3], output_path=sys.argv[4], save_output=False, device='cuda', quiet=True);"
            "ok, v = p.run_prediction(); assert ok and p.resident is None; np.savez(sys.argv[5], **v)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code, root, ck, grids + "/", str(tmp_path / "out3"), str(tmp_path / "cold.npz")], check=True, timeout=600)
    vols3 = np.load(str(tmp_path / "cold.npz"))
    for k in vols:
        assert np.array_equal(vols[k], vols3[k]), k
    # deleting grids_path (utils/modeler.py:755 does after every map) invalidates the hand-off: nothing resident is found for it again
    import shutil
    from mica_amd import handoff
    shutil.rmtree(grids)
    assert handoff.lookup_grids(os.path.join(grids, "normalized_map_grids")) is None
    gone = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
    assert gone.run_prediction() == (False, {})
    # the disk-free pipeline on the same normalised map + encodings
    vp = VolumePredictor(eng, 48, 8, batch=4)
    d_vol = torch.from_numpy(np.ascontiguousarray(vol_ref)).cuda()
    d_af = torch.from_numpy(np.ascontiguousarray(np.stack([vo.transpose_axes(e, 1, 2, 3, [7, 6, 5])[0] for e in enc_ref]))).cuda()
    mem = vp.predict_volume(d_vol, d_af)
    for k in vols:
        assert np.array_equal(vols[k], mem[k].cpu().numpy()), k
    # the oracle on one tile with atoms (tile 1) and one without (tile 3)
    for t in (1, 3):
        i, j, k, di, dj, dk = idx[t]
        lb, lc, la = mo.mica_forward(weights, torch.from_numpy(tiles_ref[t][None, None]), torch.from_numpy(af_tiles_ref[t][None]))
        pb, pc, pa, pp = mo.postprocess(lb, lc, la)
        crop = lambda a: a[..., 8:8 + di, 8:8 + dj, 8:8 + dk]
        assert np.abs(vols["backbone_probability"][i:i + di, j:j + dj, k:k + dk] - crop(pb.numpy()[0])).max() < 1e-4
        assert np.abs(vols["carbon_alpha_probability"][i:i + di, j:j + dj, k:k + dk] - crop(pc.numpy()[0])).max() < 1e-4
        assert np.abs(vols["amino_acid_probability"][:, i:i + di, j:j + dj, k:k + dk] - crop(pa.numpy()[0])).max() < 1e-4


def test_training_tilers_create_and_save_grids(tmp_path, eng):
    """Row a6: scripts_for_training_data/create_grids_for_normalized_map.py:18-101 (no transpose, prefix `grid_`, tiles
    with max < 0.01 skipped) and its four siblings (every tile kept)."""
    from mica_amd import mrc
    from mica_amd.create_grids import GridCreator
    data = synth_density((50, 100, 40), 23)               # [nz, ny, nx]; NOT transposed by these tilers
    data[:, 40:100, :] *= 0.009                            # the whole windows of the tiles j = 48 (rows 40..103) and j = 96 stay below 0.01
    p = str(tmp_path / "m.mrc")
    mrc.write_mrc(p, data, mapc=3, mapr=2, maps=1, nxstart=1, nystart=2, nzstart=3, origin=(1.0, 2.0, 3.0), voxel_size=(1.0, 1.0, 1.0))
    gc = GridCreator(quiet=True, engine=eng)
    n = gc.create_and_save_grids(p, str(tmp_path / "norm"), min_grid_max=0.01)
    tiles, idx = vo.tile_volume(data, 48, 8)
    keep = [t for t in range(len(idx)) if tiles[t].max() >= 0.01]
    assert len(idx) == 2 * 3 * 1 and 0 < len(keep) < len(idx) and n == len(keep)
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "norm" / "*.npz")))
    assert files == sorted(f"grid_i{idx[t][0]}_j{idx[t][1]}_k{idx[t][2]}.npz" for t in keep)
    for t in keep:
        i, j, k, di, dj, dk = idx[t]
        d = np.load(str(tmp_path / "norm" / f"grid_i{i}_j{j}_k{k}.npz"))
        assert np.array_equal(d["grid"], tiles[t]) and d["grid"].dtype == np.float32
        assert (int(d["i"]), int(d["j"]), int(d["k"]), int(d["di"]), int(d["dj"]), int(d["dk"])) == (i, j, k, di, dj, dk)
        assert tuple(d["orig_shape"]) == (50, 100, 40) and int(d["grid_size"]) == 48 and int(d["padding"]) == 8
        assert (int(d["mapc"]), int(d["mapr"]), int(d["maps"])) == (3, 2, 1)
        assert float(d["origin"]["x"]) == 1.0 and float(d["origin"]["z"]) == 3.0 and float(d["voxel_size"]["y"]) == 1.0
    # the mask / encoding variants keep every tile (create_grids_for_AF3_encodings.py:78-93)
    n_all = gc.create_and_save_grids(p, str(tmp_path / "mask"))
    assert n_all == len(idx) == len(glob.glob(str(tmp_path / "mask" / "*.npz")))
    # other tilings
    n32 = gc.create_and_save_grids(p, str(tmp_path / "g32"), grid_size=32, padding=16)
    t32, i32 = vo.tile_volume(data, 32, 16)
    assert n32 == len(i32) == 2 * 4 * 2
    d = np.load(str(tmp_path / "g32" / "grid_i32_j96_k32.npz"))
    assert np.array_equal(d["grid"], t32[[tuple(r[:3]) for r in i32.tolist()].index((32, 96, 32))])
    assert gc.create_and_save_grids(str(tmp_path / "missing.mrc"), str(tmp_path / "x")) == 0


def test_config2_through_the_product_entry_two_ranks_on_one_gpu(tmp_path, weights, monkeypatch):
    """BASELINE configs[2] behind the boundary the reference calls: ONE process builds `CryoEMPredictor(...)` and calls
    `run_prediction()` (utils/modeler.py:722-738).  `gpus=2`: this process is rank 0, rank 1 is a fresh child that
    mica_amd/multi.py starts; the volumes GridCreator left on the GPU are broadcast to it, both run their share of the tile batches,
    rank 0 stitches.  Both ranks on the one GPU of this box over gloo (host-staged broadcast and exchange - RCCL refuses two ranks
    on one device): the four volumes equal the single-rank ones bit for bit; a second map goes through the SAME worker process; a
    missing checkpoint fails loudly on every rank and leaves no process behind.  Then the RCCL branch in the only form one GPU
    allows: a pool of ONE rank with the collective forced, initialised AFTER this process has used the GPU (which rank 0 of a real
    node always has: the tiler ran in it)."""
    from mica_amd import handoff, mrc, multi
    from mica_amd.create_grids import GridCreator
    from mica_amd.predict import CryoEMPredictor

    ck = str(tmp_path / "ckpt.pth")
    _save_ckpt(ck, weights)

    def get_data(tag, shape, seed):
        inp = tmp_path / tag
        os.makedirs(inp / "AF3_encodings")
        raw = synth_density(shape, seed)
        mp = str(inp / "resampled_normalized_map.mrc")
        mrc.write_mrc(mp, raw, nxstart=1, nystart=2, nzstart=3)
        enc = synth_af(shape, seed, 2e-3)
        enc[:, :, :, : shape[2] // 2] = 0                       # tiles without atoms: per-tile gating on both ranks
        from mica_amd.af3_encoding import CHANNEL_NAMES
        for c, name in enumerate(CHANNEL_NAMES):
            mrc.write_mrc(str(inp / "AF3_encodings" / f"{name}_encoding.mrc"), enc[c], nxstart=1, nystart=2, nzstart=3)
        grids = str(inp / "grids")
        gc = GridCreator(quiet=True, write_files=False)
        assert gc.create_normalized_map_grids(mp, os.path.join(grids, "normalized_map_grids"))["success"]
        assert gc.create_AF3_encodings_grids(str(inp / "AF3_encodings"), os.path.join(grids, "AF3_encoding_grids"))["success"]
        return grids

    def predict(grids, gpus, model=ck):
        pred = CryoEMPredictor(model_path=model, grids_path=grids + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda",
                               quiet=True, batch_size=2, gpus=gpus)
        pred.rank_backend, pred.rank_devices = "gloo", [0, 0]
        pred.keep_resident_volumes = True                      # the same resident volumes serve the one-rank and the two-rank run
        ok, vols = pred.run_prediction()
        assert (pred.resident is not None) or not ok
        return ok, vols, pred

    g1 = get_data("m1", (70, 50, 100), 91)                      # 2 x 2 x 3 = 12 tiles = 6 batches of 2: three rounds of two ranks
    ok, ref1, _ = predict(g1, 1)
    assert ok
    ok, two1, p2 = predict(g1, 2)
    assert ok and p2.rank_pool is not None and p2.rank_pool.maps == 1
    worker = p2.rank_pool.procs[0]
    for k in ref1:
        assert np.array_equal(ref1[k], two1[k]), k
    st = p2.rank_pool.last_status
    assert [s["rank"] for s in st] == [0, 1] and st[0]["stats"]["collectives"] == 3 and st[0]["stats"]["world"] == 2
    print(p2.rank_pool.startup_report())
    # a second map of another shape: the same pool, the same worker process
    g2 = get_data("m2", (50, 40, 100), 92)                      # 2 x 1 x 3 = 6 tiles = 3 batches: the last round has an idle rank
    ok, ref2, _ = predict(g2, 1)
    ok2, two2, p3 = predict(g2, 2)
    assert ok and ok2 and p3.rank_pool is p2.rank_pool and p3.rank_pool.procs[0] is worker and worker.poll() is None and p3.rank_pool.maps == 2
    for k in ref2:
        assert np.array_equal(ref2[k], two2[k]), k
    # the tiler ran elsewhere (only its FILES exist): the volumes are rebuilt from the complete set of tile files and sharded the same
    # way; with one encoding tile file missing the set is not one the shortcut reproduces exactly (the reference feeds zeros for that
    # tile's 24 channels): one GPU, tile by tile, and the same volumes as gpus=1 on that directory
    gsync = str(tmp_path / "m2" / "grids_files")
    gcs = GridCreator(quiet=True, write_files="sync")
    assert gcs.create_normalized_map_grids(str(tmp_path / "m2" / "resampled_normalized_map.mrc"), os.path.join(gsync, "normalized_map_grids"))["success"]
    assert gcs.create_AF3_encodings_grids(str(tmp_path / "m2" / "AF3_encodings"), os.path.join(gsync, "AF3_encoding_grids"))["success"]

    def from_files(gpus):
        pred = CryoEMPredictor(model_path=ck, grids_path=gsync + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda",
                               quiet=True, batch_size=2, gpus=gpus)
        pred.rank_backend, pred.rank_devices = "gloo", [0, 0]
        pred.use_resident_volumes = False                      # what a predictor in another process sees: the files, nothing resident
        ok, vols = pred.run_prediction()
        assert ok and pred.resident is None
        return vols, pred
    files2, pf = from_files(2)
    assert pf.rank_pool is p2.rank_pool and pf.rank_pool.maps == 3 and pf.rank_pool.procs[0] is worker
    for k in ref2:
        assert np.array_equal(ref2[k], files2[k]), k
    os.remove(glob.glob(os.path.join(gsync, "AF3_encoding_grids", "CA_grids", "*.npz"))[0])
    one_gpu, _ = from_files(1)
    fallback, pb = from_files(2)
    assert pb.rank_pool is None or pb.rank_pool.maps == 3      # not sharded: read tile by tile, like the reference
    for k in one_gpu:
        assert np.array_equal(one_gpu[k], fallback[k]), k
    assert not np.array_equal(one_gpu["backbone_probability"], ref2["backbone_probability"])     # that tile lost its encodings
    # failure: the checkpoint vanishes between the strategy step and the workers' load -> (False, {}), pool closed, worker gone
    bad = str(tmp_path / "gone.pth")
    _save_ckpt(bad, weights)
    pred = CryoEMPredictor(model_path=bad, grids_path=g2 + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True,
                           batch_size=2, gpus=2)
    pred.rank_backend, pred.rank_devices, pred.keep_resident_volumes = "gloo", [0, 0], True
    real_load = pred.load_model

    def load_then_remove(**kw):
        r = real_load(**kw)
        os.remove(bad)
        return r
    pred.load_model = load_then_remove
    assert pred.run_prediction() == (False, {})
    assert worker.poll() is not None and p2.rank_pool.closed and not torch.distributed.is_initialized()
    # the RCCL branch on one GPU: a group of one rank, collective forced, in a process that has long used the GPU
    pool = multi.RankPool(1, tile=64, batch=2, backend="nccl", devices=[0], force_collective=True)
    try:
        pred = CryoEMPredictor(model_path=ck, grids_path=g2 + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True, batch_size=2)
        assert pred.select_processing_strategy() and pred.load_model() and pred.resident is not None
        m, a = pred.resident
        runner = multi.EngineRunner(None, 64, 2, engine=pred.engine, loaded_model=ck)
        vols = pool.predict(runner, ck, m.volume, a.volume, m.grid_size, m.padding, gather_to_root=False, to_host=True)
        assert pool.last_status[0]["stats"]["backend"] == "nccl" and pool.last_status[0]["stats"]["collectives"] == 3       # 6 tiles / 2
        for k in ref2:
            assert np.array_equal(ref2[k], vols[k]), k
        pred.engine.close()
    finally:
        pool.close()
    assert not torch.distributed.is_initialized()
    handoff.clear()
    # the whole Solver flow through the import shim with MICA_GPUS=2 in the environment: the worker is started
    # when DataPreprocessor is constructed - beside getData -, the predictor finds that very process, the volumes equal one GPU's
    from mica_amd import solver_mirrors as sm
    multi.shutdown()
    raw = ((synth_density((50, 40, 100), 93) - 0.3) * 3.0).astype(np.float32)
    inp = tmp_path / "solver" / "9999"
    os.makedirs(inp / "AF3_results")
    mrc.write_mrc(str(inp / "emd_9999.mrc"), raw)

    def solver_flow(tag):
        import time
        dp = sm.DataPreprocessor(map_path=str(inp / "emd_9999.mrc"), AF3_results=str(inp / "AF3_results"), quiet=True)
        t0 = time.time()                                   # the constructor has returned: the workers (if any) are already started
        dp.resample_and_normalize_map()
        grids = str(tmp_path / "solver" / tag)
        gc = sm.GridCreator(quiet=True)
        assert gc.create_normalized_map_grids(normalized_map_path=dp.normalized_map_path, output_dir=os.path.join(grids, "normalized_map_grids"))["success"]
        pred = sm.CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=str(tmp_path / "out"), save_output=False, device="cuda", quiet=True)
        ok, vols = pred.run_prediction()
        assert ok and len(glob.glob(os.path.join(grids, "normalized_map_grids", "*.npz"))) == 6       # the shim's background files, complete on return
        return vols, pred, t0
    one, _, _ = solver_flow("one")
    monkeypatch.setenv("MICA_GPUS", "2")
    monkeypatch.setenv("MICA_RANK_BACKEND", "gloo")
    monkeypatch.setenv("MICA_RANK_DEVICES", "0,0")
    try:
        two, pred, t0 = solver_flow("two")
        assert pred.gpus == 2 and pred.rank_pool is not None and pred.rank_pool.maps == 1 and pred.rank_pool.last_status[0]["stats"]["world"] == 2
        assert pred.rank_pool.t_spawn <= t0                            # spawned by DataPreprocessor's constructor, not by the predictor
        for k in one:
            assert np.array_equal(one[k], two[k]), k
    finally:
        multi.shutdown()
    handoff.clear()


def test_reference_batching_mode_vs_reference_golden(tmp_path, weights, golden_dir):
    """Row a18: the reference's >batch_threshold mode (utils/predict.py:176-215) tests the AF3 features of the whole batch
    (models/model.py:60).  Golden = the reference CryoEMPredictor with batch_threshold lowered to 1 on a 2-tile map whose
    tile 0 has empty AF3 channels; `reference_batching=True` reproduces it, the default (per-tile gate) reproduces the
    reference's single-sample mode."""
    from mica_amd.predict import CryoEMPredictor
    from oracle.gen_golden_r2 import refbatch_case, write_tile_files
    shape, vol, af = refbatch_case()
    write_tile_files(str(tmp_path), vol, af, shape)
    ck = str(tmp_path / "ckpt.pth")
    _save_ckpt(ck, weights)
    got = {}
    for tag, kw, thr in (("refbatch", dict(reference_batching=True), 1), ("single", dict(reference_batching=True), 200), ("default", {}, 1)):
        pred = CryoEMPredictor(model_path=ck, grids_path=str(tmp_path / "grids") + "/", output_path=str(tmp_path / "out"), save_output=False,
                               device="cuda", quiet=True, **kw)
        pred.batch_threshold = thr
        ok, vols = pred.run_prediction()
        assert ok
        assert pred.use_optimized_batching == (thr == 1) and pred.optimal_batch_size == (8 if thr == 1 else 1)
        got[tag] = vols
    for tag, gold in (("refbatch", "refbatch"), ("single", "single"), ("default", "single")):
        g = np.load(os.path.join(golden_dir, f"predictor_{gold}_af_60x40x40.npz"))
        v = got[tag]
        assert np.abs(v["backbone_probability"] - g["backbone_probability"]).max() < 1e-4, tag
        assert np.abs(v["carbon_alpha_probability"] - g["carbon_alpha_probability"]).max() < 1e-4, tag
        assert np.abs(v["amino_acid_probability"][:, ::2, ::2, ::2] - g["amino_acid_probability_sub"]).max() < 1e-4, tag
        top = np.sort(v["amino_acid_probability"], axis=0)
        mism = v["amino_acid_prediction"].astype(np.int64) != g["amino_acid_prediction"].astype(np.int64)
        assert not np.any(mism & (top[-1] - top[-2] > 2e-4)) and mism.mean() < 1e-3, tag
    # the two modes genuinely differ on tile 0
    assert np.abs(got["refbatch"]["backbone_probability"][:48] - got["single"]["backbone_probability"][:48]).max() > 1e-3


@pytest.mark.parametrize("mode", ["", "root"])
def test_config2_two_ranks_on_one_gpu_equal_single_rank(tmp_path, eng, mode):
    """BASELINE configs[2] in miniature: two fresh rank processes (gloo rendezvous, both on cuda:0) run
    predict_volume_sharded with the real engine - round-robin batches, double-buffered record exchange, rank 0 stitches -
    and must reproduce the single-process result bit for bit.  mode "root" (round 5): encodings resident as uint8 on both ranks,
    records gathered into rank 0 only, the volumes downloaded slab by slab behind the stitch."""
    from mica_amd.pipeline import VolumePredictor
    shape, batch = (100, 70, 50), 2                        # 3 x 2 x 2 = 12 tiles -> 6 batches -> 3 rounds per rank
    out = str(tmp_path / "sharded.npz")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   MICA_TEST_MODE=mode)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), out, "x".join(map(str, shape)), str(batch)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    assert str(got["collective"]) == ("gather" if mode == "root" else "all_gather")
    assert np.array_equal(got["coverage"], np.ones(shape, np.float32))
    vol = torch.from_numpy(synth_density(shape, 91)).cuda()
    af = torch.from_numpy(synth_af(shape, 91, 2e-3)).cuda()
    af[:, :, :, : shape[2] // 2] = 0
    ref = VolumePredictor(eng, 48, 8, batch).predict_volume(vol, af)
    for k in ref:
        assert np.array_equal(got[k], ref[k].cpu().numpy()), k
    assert float(ref["backbone_probability"].max()) > 0.5


@pytest.mark.parametrize("mode", ["", "root"])
def test_config2_rccl_branch_single_rank_equals_plain_pipeline(tmp_path, eng, mode):
    """(mode "root": the same with RCCL's gather into the stitching rank, uint8 encodings and the slab-by-slab download.)
    The production branch of the record exchange - RCCL all_gather_into_tensor(async_op=True) on device tensors, work.wait()
    ordering against the stitch kernels, double-buffered send / receive slots - executed on the hardware that exists: ONE rank in
    an `nccl` process group with force_collective (a fresh child process that initialises the group before any other GPU call).
    predict_volume_sharded under it equals predict_volume bit for bit, every batch went through a collective, and an all-ones
    record set stitched through the same exchange covers the volume exactly once."""
    from mica_amd.pipeline import VolumePredictor
    shape, batch = (100, 70, 50), 2                        # 12 tiles -> 6 rounds on the one rank: both slots reused three times
    out = str(tmp_path / "rccl1.npz")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", MICA_TEST_BACKEND="nccl", MICA_TEST_MODE=mode)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), out, "x".join(map(str, shape)), str(batch)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    got = np.load(out)
    assert str(got["backend"]) == "nccl" and int(got["collectives"]) == 6 and str(got["collective"]) == ("gather" if mode == "root" else "all_gather")
    assert np.array_equal(got["coverage"], np.ones(shape, np.float32))
    vol = torch.from_numpy(synth_density(shape, 91)).cuda()
    af = torch.from_numpy(synth_af(shape, 91, 2e-3)).cuda()
    af[:, :, :, : shape[2] // 2] = 0
    ref = VolumePredictor(eng, 48, 8, batch).predict_volume(vol, af)
    for k in ref:
        assert np.array_equal(got[k], ref[k].cpu().numpy()), k


def test_config4_four_384_maps_streamed(eng):
    """BASELINE configs[4] at full size: four independent 384^3 maps (two with AF3 encodings) streamed back to back with
    double-buffered H2D/D2H.  Checked against the one-map-at-a-time pipeline: map 0 completely (bit for bit), the others
    on sampled batches of tiles; plus size-independent properties of every volume."""
    from mica_amd.pipeline import VolumePredictor
    n, B = 384, 8
    vp = VolumePredictor(eng, 48, 8, batch=B)
    maps = [np.random.default_rng(1003 + i).random((n, n, n), dtype=np.float32) for i in range(4)]
    afs = [None, None, None, None]
    for i in (1, 3):
        g = torch.Generator(device="cuda").manual_seed(2001 + i)
        a = torch.empty((24, n, n, n), dtype=torch.float32, device="cuda")
        for c in range(24):
            a[c] = (torch.rand((n, n, n), generator=g, device="cuda") < 1e-3).float()
        a[:, : n // 2] = 0                                 # half of the tiles see no atoms: both AF branches in one map
        afs[i] = a.cpu().numpy()
        del a
    got = vp.predict_maps_streamed(maps, afs)
    assert len(got) == 4
    T = int(eng.lib.mica_tile_count(n, n, n, 48))
    assert T == 512
    from mica_amd._cabi import tile_table
    tab = tile_table(n, n, n, 48)
    for m in range(4):
        v = got[m]
        assert v["amino_acid_probability"].shape == (20, n, n, n) and v["backbone_probability"].dtype == np.float32
        s = v["amino_acid_probability"][:, ::7, ::5, ::3].sum(axis=0)
        assert np.abs(s - 1.0).max() < 1e-5                # softmax over the 20 residue classes, every stitched voxel written
        pr = v["amino_acid_prediction"][::3, ::5, ::7]
        assert pr.min() >= 0 and pr.max() <= 19 and np.array_equal(pr, np.round(pr))
        assert 0.0 <= v["backbone_probability"].min() and v["carbon_alpha_probability"].max() <= 1.0
        d_vol = torch.from_numpy(maps[m]).cuda()
        d_af = None if afs[m] is None else torch.from_numpy(afs[m]).cuda()
        if m == 0:
            ref = vp.predict_volume(d_vol, d_af)
            for k in ref:
                assert np.array_equal(v[k], ref[k].cpu().numpy()), (m, k)
            del ref
        else:
            for first in (0, 8 * 31, T - B):               # sampled batches: first, middle (AF boundary), last
                rec = vp.run_batch(d_vol, d_af, first, B).cpu().numpy()
                for q in range(B):
                    i, j, k, di, dj, dk = tab[first + q]
                    c = rec[q][:, 8:8 + di, 8:8 + dj, 8:8 + dk]
                    assert np.array_equal(v["backbone_probability"][i:i + di, j:j + dj, k:k + dk], c[0]), (m, first, q)
                    assert np.array_equal(v["amino_acid_prediction"][i:i + di, j:j + dj, k:k + dk], c[2])
                    assert np.array_equal(v["amino_acid_probability"][:, i:i + di, j:j + dj, k:k + dk], c[3:])
        del d_vol, d_af
    # maps differ, so do their volumes (no buffer was reused across maps by mistake)
    assert not np.array_equal(got[0]["backbone_probability"], got[2]["backbone_probability"])


def test_metric_config_512_map_whole_path_properties(eng):
    """BASELINE.json's metric config at full size through the whole path: the synthetic 512^3 map with 24-channel AF3 encodings,
    reference tiling (48, 8) -> 1331 windows, gather -> forward -> softmax/argmax -> stitch.  Size-independent properties: every
    voxel written exactly once with a proper distribution, sampled tiles equal to a direct run of their batch, and a second
    pass over the map reproduces the four volumes bit for bit (checksums)."""
    from mica_amd._cabi import tile_table
    from mica_amd.pipeline import VolumePredictor
    n, B = 512, 8
    vol = torch.from_numpy(np.random.default_rng(1002).random((n, n, n), dtype=np.float32)).cuda()
    g = torch.Generator(device="cuda").manual_seed(2001)
    af = torch.empty((24, n, n, n), dtype=torch.float32, device="cuda")
    for c in range(24):
        af[c] = (torch.rand((n, n, n), generator=g, device="cuda") < 1e-3).float()
    af[:, :, :, n // 2:] = 0                               # windows beyond z = 256 + halo see no atoms: both AF branches
    vp = VolumePredictor(eng, 48, 8, batch=B)
    out = vp.predict_volume(vol, af)
    T = int(eng.lib.mica_tile_count(n, n, n, 48))
    assert T == 1331
    aa = out["amino_acid_probability"]
    s = aa[:, ::5, ::7, ::3].sum(dim=0)
    assert float((s - 1.0).abs().max()) < 1e-5             # a softmax everywhere: no voxel left at its initial zero
    pred = out["amino_acid_prediction"]
    assert float(pred.min()) >= 0 and float(pred.max()) <= 19 and bool((pred[::4, ::4, ::4] == pred[::4, ::4, ::4].round()).all())
    for k in ("backbone_probability", "carbon_alpha_probability"):
        assert float(out[k].min()) >= 0.0 and float(out[k].max()) <= 1.0 and float(out[k].max()) > 0.5
    assert bool((pred == aa.argmax(dim=0).float()).float().mean() > 0.9999)      # prediction = first maximum of the stitched scores
    tab = tile_table(n, n, n, 48)
    for first in (0, 8 * 83, T - B - 3):                   # a corner batch, an interior batch across the AF boundary, the ragged end
        rec = vp.run_batch(vol, af, first, B)
        for q in range(B):
            i, j, k, di, dj, dk = (int(v) for v in tab[first + q])
            c = rec[q][:, 8:8 + di, 8:8 + dj, 8:8 + dk]
            assert torch.equal(out["backbone_probability"][i:i + di, j:j + dj, k:k + dk], c[0])
            assert torch.equal(aa[:, i:i + di, j:j + dj, k:k + dk], c[3:])
    sums = {k: (float(v.double().sum()), float(v.double().pow(2).sum())) for k, v in out.items()}
    del out, aa, pred
    again = vp.predict_volume(vol, af)
    for k, v in again.items():
        assert (float(v.double().sum()), float(v.double().pow(2).sum())) == sums[k], k


def test_command_line_entry_map_to_volumes(tmp_path, eng, weights):
    """`python -m mica_amd --map ... --model ... --docked-model ... --out ...`: the disk-free chain behind a thin CLI, against
    the same chain called step by step (non-trivial axis order in the header, anisotropic voxels)."""
    from mica_amd import mrc
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.preprocessing import DataPreprocessor
    raw = ((synth_density((30, 36, 40), 27) - 0.3) * 2.0).astype(np.float32)
    mp = str(tmp_path / "emd.mrc")
    mrc.write_mrc(mp, raw, voxel_size=(1.25, 1.0, 1.5), origin=(1.0, 2.0, -1.0), mapc=2, mapr=1, maps=3, nxstart=3, nystart=4, nzstart=5)
    pdb = tmp_path / "m_af3_docked.pdb"
    rng = np.random.default_rng(9)
    pdb.write_text("".join(_pdb_line(i + 1, ["N", "CA", "C", "O"][i % 4], "ALA", i // 4 + 1, *(rng.random(3) * 20 + 5)) for i in range(120)) + "END\n")
    ck = str(tmp_path / "ckpt.pth")
    _save_ckpt(ck, weights)
    out = str(tmp_path / "out")
    r = subprocess.run([sys.executable, "-m", "mica_amd", "--map", mp, "--model", ck, "--docked-model", str(pdb), "--out", out, "--batch", "4"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sub-grids" in r.stdout and "offset [4.0, 3.0, 5.0]" in r.stdout       # [nx, ny, nz]start permuted like the axes
    # the same chain, step by step
    data, hd = mrc.read_mrc(mp)
    dp = DataPreprocessor(mp, str(tmp_path / "af3"), quiet=True, engine=eng)
    norm, _, _ = dp.normalize_array(np.asarray(data), hd.voxel_size, 1.0)
    hdn = mrc.MrcHeader(nx=norm.shape[2], ny=norm.shape[1], nz=norm.shape[0], mapc=2, mapr=1, maps=3, nxstart=3, nystart=4, nzstart=5, origin=hd.origin)
    vol, off = mrc.transpose_to_xyz(norm, hdn)
    enc = dp.encode_AF3_volume(str(pdb), hd.origin, norm.shape).cpu().numpy()
    af = np.stack([mrc.transpose_to_xyz(e, hdn)[0] for e in enc])
    assert off == [4.0, 3.0, 5.0] and af.sum() > 50
    ref = VolumePredictor(eng, 48, 8, 4).predict_volume(torch.from_numpy(np.ascontiguousarray(vol)).cuda(), torch.from_numpy(np.ascontiguousarray(af)).cuda())
    for k, v in ref.items():
        got = np.load(os.path.join(out, f"{k}.npy"))
        assert got.dtype == np.float32 and np.array_equal(got, v.cpu().numpy()), k
    # the same command on two ranks (`--gpus 2`: this box has one GPU, so both ranks on cuda:0 over gloo): the CLI process is rank 0 and
    # starts the worker itself (mica_amd/multi.py); same files, bit for bit
    out2 = str(tmp_path / "out2")
    r = subprocess.run([sys.executable, "-m", "mica_amd", "--map", mp, "--model", ck, "--docked-model", str(pdb), "--out", out2, "--batch", "4",
                        "--gpus", "2", "--rank-backend", "gloo", "--rank-devices", "0,0"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ranks 2 backend gloo" in r.stdout and "rank 1: python + torch import" in r.stdout
    for k in ref:
        assert np.array_equal(np.load(os.path.join(out2, f"{k}.npy")), np.load(os.path.join(out, f"{k}.npy"))), k
