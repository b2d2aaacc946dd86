"""The reference's getData + nnPred sequence (utils/modeler.py:673-738) through the three mirrors, end to end, on one map:

    GridCreator.create_normalized_map_grids + create_AF3_encodings_grids  ->  CryoEMPredictor(grids_path).run_prediction()

timed (a) with the in-process hand-off (mica_amd/handoff.py: the volumes stay on the GPU, the tile files are written in the
background - `write_files=True`, the default - or not at all), (b) with the predictor reading the 25 .npz files per tile (what a
predictor in another process does), next to (c) the disk-free VolumePredictor on the same map.  All three must agree bit for bit.
usage: python tools/file_predictor_bench.py [n=256]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd import handoff, mrc
from mica_amd.af3_encoding import CHANNEL_NAMES
from mica_amd.create_grids import GridCreator
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.predict import CryoEMPredictor
from mica_amd.weights import synth_state_dict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 256
tmp = tempfile.mkdtemp(prefix="mica_filebench_")
try:
    vol = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
    af = (np.random.default_rng(2).random((24, n, n, n), dtype=np.float32) < 1e-3).astype(np.float32)
    w = synth_state_dict(2022)
    mp = os.path.join(tmp, "resampled_normalized_map.mrc")
    mrc.write_mrc(mp, vol)
    os.makedirs(os.path.join(tmp, "AF3_encodings"))
    for c, name in enumerate(CHANNEL_NAMES):
        mrc.write_mrc(os.path.join(tmp, "AF3_encodings", f"{name}_encoding.mrc"), af[c])
    ck = os.path.join(tmp, "ckpt.pth")
    torch.save({"model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in w.items()}}, ck)
    grids = os.path.join(tmp, "grids")
    eng = Engine(0, max_batch=1, tile_size=64)           # the tiler's own context (gather kernel for the file writer)
    results = {}

    def chain(tag, write_files, resident=True, threads=4):
        """getData's two tiler calls + nnPred, as the reference's call sites make them; then what modeler.py:755 does"""
        shutil.rmtree(grids, ignore_errors=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gc = GridCreator(quiet=True, engine=eng, write_files=write_files)
        r1 = gc.create_normalized_map_grids(normalized_map_path=mp, output_dir=os.path.join(grids, "normalized_map_grids"))
        r2 = gc.create_AF3_encodings_grids(AF3_encodings_path=os.path.join(tmp, "AF3_encodings"), output_dir=os.path.join(grids, "AF3_encoding_grids"))
        assert r1["success"] and r2["success"]
        t1 = time.perf_counter()
        pred = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=os.path.join(tmp, "out"), save_output=False, device="cuda", quiet=True)
        pred.use_resident_volumes = resident
        pred.loader_threads = threads
        ok, vols = pred.run_prediction()
        t2 = time.perf_counter()
        assert ok and (pred.resident is not None) == resident
        T = r1["grid_count"]
        nfiles = sum(len([f for f in fs if f.endswith('.npz')]) for _, _, fs in os.walk(grids))
        shutil.rmtree(grids)
        t3 = time.perf_counter()
        print(f"{tag}: tiling {t1 - t0:.2f} s + prediction {t2 - t1:.2f} s (model load {pred.timing_stats['model_loading']:.2f}, inference "
              f"{pred.timing_stats['inference']:.2f}) = {t2 - t0:.2f} s -> {T / (t2 - t0):.1f} sub-grids/s end to end ({T} tiles of the {n}^3 map; "
              f"{nfiles} tile files on disk when nnPred returned; rmtree {t3 - t2:.2f} s)", flush=True)
        results[tag] = vols
        return T

    chain("warm-up (hand-off, no files)", False)
    T = chain("hand-off, tile files written in the background (default)", True)
    chain("hand-off, no tile files (write_files=False)", False)
    chain("predictor reads the tile files (cold path), 4 reader threads", "sync", resident=False)
    vp = VolumePredictor(Engine(0, max_batch=8, tile_size=64), 48, 8, 8)
    vp.e.load_state_dict(w)
    dv, da = torch.from_numpy(np.ascontiguousarray(vol.transpose(2, 1, 0))).cuda(), torch.from_numpy(np.ascontiguousarray(af.transpose(0, 3, 2, 1))).cuda()
    vp.predict_volume(dv, da); torch.cuda.synchronize()
    t0 = time.perf_counter(); mem = vp.predict_volume(dv, da); torch.cuda.synchronize(); dm = time.perf_counter() - t0
    print(f"disk-free VolumePredictor on the same map, volumes left on the GPU: {dm:.2f} s -> {T / dm:.1f} sub-grids/s")
    for tag, vols in results.items():
        for k in vols:
            assert np.array_equal(vols[k], mem[k].cpu().numpy()), (tag, k)
    print("every route == disk-free volumes: bit-identical")
finally:
    handoff.clear()
    shutil.rmtree(tmp, ignore_errors=True)
