"""Throughput of the FILE-BASED CryoEMPredictor mirror (25 .npz files per tile, the reference's on-disk wire format) next to the
disk-free VolumePredictor on the same map (development aid).  usage: python tools/file_predictor_bench.py [n=192]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd import mrc
from mica_amd.af3_encoding import CHANNEL_NAMES
from mica_amd.create_grids import GridCreator
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.predict import CryoEMPredictor
from mica_amd.weights import synth_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
tmp = tempfile.mkdtemp(prefix="mica_filebench_")
try:
    vol = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
    af = (np.random.default_rng(2).random((24, n, n, n), dtype=np.float32) < 1e-3).astype(np.float32)
    w = synth_state_dict(2022)
    eng = Engine(0, max_batch=8, tile_size=64)
    eng.load_state_dict(w)
    gc = GridCreator(quiet=True, engine=eng)
    mp = os.path.join(tmp, "resampled_normalized_map.mrc")
    mrc.write_mrc(mp, vol)
    os.makedirs(os.path.join(tmp, "AF3_encodings"))
    for c, name in enumerate(CHANNEL_NAMES):
        mrc.write_mrc(os.path.join(tmp, "AF3_encodings", f"{name}_encoding.mrc"), af[c])
    t0 = time.perf_counter()
    r1 = gc.create_normalized_map_grids(mp, os.path.join(tmp, "grids", "normalized_map_grids"))
    r2 = gc.create_AF3_encodings_grids(os.path.join(tmp, "AF3_encodings"), os.path.join(tmp, "grids", "AF3_encoding_grids"))
    t_tile = time.perf_counter() - t0
    T = r1["grid_count"]
    ck = os.path.join(tmp, "ckpt.pth")
    torch.save({"model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in w.items()}}, ck)
    for threads in (1, 2, 4):
        pred = CryoEMPredictor(ck, os.path.join(tmp, "grids") + "/", os.path.join(tmp, "out"), save_output=False, device="cuda", quiet=True)
        if threads:
            pred.loader_threads = threads
        t0 = time.perf_counter()
        ok, vols = pred.run_prediction()
        dt = time.perf_counter() - t0
        assert ok
        print(f"file-based CryoEMPredictor, {n}^3 map, {T} tiles x 25 npz files, {pred.loader_threads} reader threads: {dt:.2f} s total "
              f"(model load {pred.timing_stats['model_loading']:.2f} s, inference {pred.timing_stats['inference']:.2f} s) -> "
              f"{T / pred.timing_stats['inference']:.1f} sub-grids/s", flush=True)
    vp = VolumePredictor(eng, 48, 8, 8)
    dv, da = torch.from_numpy(np.ascontiguousarray(vol.transpose(2, 1, 0))).cuda(), torch.from_numpy(np.ascontiguousarray(af.transpose(0, 3, 2, 1))).cuda()
    vp.predict_volume(dv, da); torch.cuda.synchronize()
    t0 = time.perf_counter(); mem = vp.predict_volume(dv, da); torch.cuda.synchronize(); dm = time.perf_counter() - t0
    print(f"disk-free VolumePredictor on the same map: {dm:.2f} s -> {T / dm:.1f} sub-grids/s; tiling to files took {t_tile:.1f} s "
          f"({T * 25} npz files)")
    for k in vols:
        assert np.array_equal(vols[k], mem[k].cpu().numpy()), k
    print("file-based == disk-free volumes: bit-identical")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
