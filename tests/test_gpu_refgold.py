"""The product's GridCreator / DataPreprocessor / AF3 rasteriser (HIP kernels behind the C ABI) against goldens that the
REFERENCE's own code produced (oracle/gen_golden_r3.py ran utils/create_grids.py, utils/preprocessing.py and the training
tiler unmodified under I/O-only `mrcfile` / `Bio` adapters): tiles, index tables, offsets, result dicts, directory layouts,
normalised maps of every MRC mode the reference accepts, the 24-channel encodings.  Bit-exact throughout.
Plus the whole path at BASELINE's stride-32 tiling (32, 16) on one rank and on two."""
import hashlib
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from mica_amd.synth import synth_af, synth_density
from oracle import volume_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def eng(weights):
    from mica_amd.engine import Engine
    e = Engine(0, max_batch=4, tile_size=64)
    e.load_state_dict(weights)
    yield e
    e.close()


def _tiles_of(gdir, prefix):
    recs = [np.load(os.path.join(gdir, f)) for f in os.listdir(gdir)]
    recs.sort(key=lambda z: (int(z["i"]), int(z["j"]), int(z["k"])))
    return recs


def test_gridcreator_vs_reference_run_goldens(tmp_path, eng, golden_dir):
    """Rows a3/a4: 20 cases (4 shapes x axis orders x tilings (48,8) and (32,16)) cut by the reference's GridCreator."""
    from mica_amd import mrc
    from mica_amd.create_grids import GridCreator
    ref = json.load(open(os.path.join(golden_dir, "tiler_ref.json")))
    gc = GridCreator(quiet=True, engine=eng)
    for n, c in enumerate(ref["cases"]):
        vol = synth_density(tuple(c["shape"]), c["seed"])
        sx, sy, sz = c["starts_xyz"]
        p = str(tmp_path / "in.mrc")
        mrc.write_mrc(p, vol, voxel_size=(1.0, 1.0, 1.0), origin=tuple(c["meta"]["origin"]), mapc=c["axes"][0], mapr=c["axes"][1],
                      maps=c["axes"][2], nxstart=sx, nystart=sy, nzstart=sz)
        gdir = str(tmp_path / f"g{n}")
        count, offset = gc.create_grids_from_mrc(p, gdir, grid_size=c["grid"], padding=c["pad"], file_prefix="pfx")
        assert count == c["count"] and [float(o) for o in offset] == c["offset"], c["shape"]
        recs = _tiles_of(gdir, "pfx")
        assert sorted(os.listdir(gdir)) == sorted(f"pfx_i{i}_j{j}_k{k}.npz" for i, j, k, *_ in c["idx"])
        assert [[int(z[k]) for k in ("i", "j", "k", "di", "dj", "dk")] for z in recs] == c["idx"]
        assert [sha(z["grid"]) for z in recs] == c["tile_sha256"]
        z, m = recs[-1], c["meta"]
        assert sorted(z.files) == m["keys"] and str(z["grid"].dtype) == m["grid_dtype"]
        assert [int(v) for v in z["orig_shape"]] == m["orig_shape"] and int(z["grid_size"]) == m["grid_size"] and int(z["padding"]) == m["padding"]
        assert [float(z["voxel_size"][()][k]) for k in "xyz"] == m["voxel_size"] and [float(z["origin"][()][k]) for k in "xyz"] == m["origin"]
        assert (int(z["mapc"]), int(z["mapr"]), int(z["maps"])) == (m["mapc"], m["mapr"], m["maps"])


def test_gridcreator_wrappers_and_training_tiler_vs_reference_run_goldens(tmp_path, eng, golden_dir):
    """Rows a5/a6: create_normalized_map_grids, create_AF3_encodings_grids (result dicts, directory layout, tiles) and
    scripts_for_training_data/create_grids_for_normalized_map.py::create_and_save_grids."""
    from mica_amd import mrc
    from mica_amd.create_grids import GridCreator
    ref = json.load(open(os.path.join(golden_dir, "tiler_ref.json")))
    gc = GridCreator(quiet=True, engine=eng)
    w = ref["normalized_map_grids"]
    p = str(tmp_path / "resampled_normalized_map.mrc")
    sx, sy, sz = w["starts_xyz"]
    mrc.write_mrc(p, synth_density(tuple(w["shape"]), w["seed"]), nxstart=sx, nystart=sy, nzstart=sz)
    gdir = str(tmp_path / "grids" / "normalized_map_grids")
    res = gc.create_normalized_map_grids(p, gdir)
    assert sorted(res) == w["result_keys"] and {k: res[k] for k in w["result"]} == w["result"]
    # default write_files="sync" (the reference's contract): every file is there, complete, when the wrapper has returned
    assert len(glob.glob(os.path.join(gdir, "*.npz"))) == res["grid_count"] and gc.wait_for_files() == 0
    assert sorted(os.listdir(gdir)) == w["files"]
    assert sha(np.stack([z["grid"] for z in _tiles_of(gdir, "normalized_map_grid")])) == w["tiles_sha256"]
    miss = gc.create_normalized_map_grids(str(tmp_path / "nope.mrc"), gdir)
    assert sorted(miss) == w["missing_result_keys"] and miss["success"] is w["missing_success"] is False

    a = ref["AF3_encoding_grids"]
    enc = (synth_density((len(a["channels"]), *a["shape"]), a["seed"]) < a["threshold"]).astype(np.float32)
    edir = tmp_path / "AF3_encodings"
    os.makedirs(edir)
    for c, name in enumerate(a["channels"]):
        mrc.write_mrc(str(edir / f"{name}_encoding.mrc"), enc[c])
    adir = str(tmp_path / "grids" / "AF3_encoding_grids")
    res = gc.create_AF3_encodings_grids(str(edir), adir, parallel=False)
    assert sorted(res) == a["result_keys"] and {k: res[k] for k in a["result"]} == a["result"]
    assert len(glob.glob(os.path.join(adir, "*", "*.npz"))) == res["total_grids"] and gc.wait_for_files() == 0
    assert sorted(os.listdir(adir)) == a["dirs"]
    for name, lay in a["layout"].items():
        d = os.path.join(adir, f"{name}_grids")
        assert sorted(os.listdir(d)) == lay["files"]
        assert sha(np.stack([z["grid"] for z in _tiles_of(d, f"{name}_grid")])) == lay["tiles_sha256"]

    t = ref["training_tiler"]
    vol = synth_density(tuple(t["shape"]), t["seed"])
    vol[:t["slab"][0]] *= t["slab"][1]
    p = str(tmp_path / "train.mrc")
    mrc.write_mrc(p, vol, mapc=t["axes"][0], mapr=t["axes"][1], maps=t["axes"][2])
    for key, rec in t["tilings"].items():
        grid, pad = (int(v) for v in key.split("_"))
        d = str(tmp_path / f"train_{key}")
        n = gc.create_and_save_grids(p, d, grid_size=grid, padding=pad, min_grid_max=0.01)
        assert n == rec["count"] and sorted(os.listdir(d)) == rec["files"]
        assert sha(np.stack([z["grid"] for z in _tiles_of(d, "grid")])) == rec["tiles_sha256"]
        assert gc.create_and_save_grids(p, str(tmp_path / f"all_{key}"), grid_size=grid, padding=pad) == rec["all"]


def test_preprocessor_vs_reference_run_goldens_all_mrc_modes(tmp_path, eng, golden_dir):
    """Row a1 on the reference's own outputs: float32 maps (incl. anisotropic voxels, odd sizes, NaN, all-negative, constant),
    int8 / int16 / uint16 maps (MRC modes 0/1/6: scipy keeps the integer dtype through zoom, numpy normalises in float64) and
    float16 (mode 12: scipy refuses it, the reference fails and writes nothing)."""
    from mica_amd import mrc
    from mica_amd.preprocessing import DataPreprocessor
    from oracle.gen_golden_r3 import _norm_inputs           # seeded input arrays only
    ref = json.load(open(os.path.join(golden_dir, "normaliser_ref.json")))
    inputs = _norm_inputs()
    assert set(inputs) == set(ref["cases"]) and {r["dtype"] for r in ref["cases"].values()} == {"float32", "int8", "int16", "uint16", "float16"}
    for name, rec in ref["cases"].items():
        vol, voxel = inputs[name]
        d = tmp_path / name
        os.makedirs(d / "AF3_results")
        src = str(d / "map.mrc")
        mrc.write_mrc(src, vol, voxel_size=voxel, origin=(4.0, 5.0, 6.0), nxstart=1, nystart=2, nzstart=3)
        dp = DataPreprocessor(src, str(d / "AF3_results"), quiet=True, engine=eng)
        dp.resample_and_normalize_map()
        dst = str(d / "resampled_normalized_map.mrc")
        assert os.path.exists(dst) == rec["written"], name
        if not rec["written"]:
            continue
        got, hd = mrc.read_mrc(dst)
        assert got.dtype == np.float32 and list(got.shape) == rec["out_shape"], name
        assert np.array_equal(got[::3, ::3, ::3], np.load(os.path.join(golden_dir, f"normaliser_ref_{name}.npy"))), name
        assert sha(got) == rec["sha256"], name
        h = rec["header"]
        assert list(hd.voxel_size) == h["voxel"] and list(hd.origin) == h["origin"] and [hd.nxstart, hd.nystart, hd.nzstart] == h["starts_xyz"]
        assert [hd.mapc, hd.mapr, hd.maps] == h["axes"] and dp.normalized_map_path == dst
        _, med, pct = dp.normalize_array(vol, voxel)
        assert med == rec["median"] and pct == rec["percentile"], name


def _write_pdb(path, residues):
    """The atom list of a golden case as the PDB records Bio.PDB.PDBIO writes (ATOM / HETATM by hetero flag)."""
    lines, serial = [], 0
    for rs, (chain, resname, het, atoms) in enumerate(residues):
        for a in atoms:
            serial += 1
            nm = a[0] if len(a[0]) == 4 else " " + a[0].ljust(3)
            lines.append("%-6s%5d %4s%1s%3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % (
                "ATOM" if het == " " else "HETATM", serial, nm, " ", resname, chain, rs + 1, a[1], a[2], a[3], 1.0, 20.0, a[0][0]))
    open(path, "w").write("".join(lines) + "END\n")


def test_af3_rasteriser_vs_reference_run_goldens(tmp_path, eng, golden_dir):
    """Row f3 / a2: create_AF3_encodings' atom loop run by the reference (5 cases: cubic, non-cubic inside the clip box, and
    non-cubic where its scatter raises IndexError) - through the kernel directly and through DataPreprocessor with a PDB file."""
    from mica_amd import af3_encoding, mrc
    from mica_amd.engine import MicaHipError
    from mica_amd.preprocessing import DataPreprocessor
    ref = json.load(open(os.path.join(golden_dir, "af3_ref.json")))
    for c in ref["cases"]:
        shape = tuple(c["shape"])
        flat = [(a[0], r[1], a[1:4]) for r in c["atoms"] if r[2] == " " for a in r[3]]
        args = (eng, np.array([f[2] for f in flat], np.float32), [f[0] for f in flat], [f[1] for f in flat], c["origin"], shape)
        if c["success"]:
            got = af3_encoding.rasterise(*args).cpu().numpy()
            assert got.dtype == np.float32 and sha(got) == c["sha256"] and int(got.sum()) == c["ones"]
            assert [int(v) for v in got.sum(axis=(1, 2, 3))] == c["per_channel"]
        else:
            with pytest.raises(MicaHipError):
                af3_encoding.rasterise(*args)
        # the file route: normalised map header + PDB -> 24 <CH>_encoding.mrc files
        d = tmp_path / f"af_{c['seed']}"
        os.makedirs(d / "AF3_results")
        mp = str(d / "resampled_normalized_map.mrc")
        mrc.write_mrc(mp, synth_density(shape, c["seed"]), origin=tuple(c["origin"]), nxstart=1, nystart=2, nzstart=3)
        pdb = str(d / "docked.pdb")
        _write_pdb(pdb, c["atoms"])
        dp = DataPreprocessor(mp, str(d / "AF3_results"), quiet=True, engine=eng)
        dp.normalized_map_path = mp
        assert dp.create_AF3_encodings(pdb) is c["success"]
        if c["success"]:
            files = sorted(os.listdir(d / "AF3_encodings"))
            assert files == sorted(f"{ch}_encoding.mrc" for ch in af3_encoding.CHANNEL_NAMES)
            enc = np.stack([mrc.read_mrc(str(d / "AF3_encodings" / f"{ch}_encoding.mrc"))[0] for ch in af3_encoding.CHANNEL_NAMES])
            assert sha(enc) == c["sha256"]


def test_clustering_front_end_vs_reference_run_goldens(eng, golden_dir):
    """Row f4: the reference's own Solver.clustering (utils/modeler.py:762-899; DBSCAN labels taken from the golden, that step is
    the caller's) against the device helpers of mica_amd/clustering.py: threshold -> cluster scores -> sorted greedy NMS ->
    refinement -> distances, neighbour lists, neighbour matrix, best neighbours.  Bit-exact."""
    from mica_amd import clustering as cl
    from oracle.gen_golden_r3 import cluster_volumes      # seeded input volumes only
    ref = json.load(open(os.path.join(golden_dir, "cluster_ref.json")))
    for c in ref["cases"]:
        shape = tuple(c["shape"])
        ca, bb, aa, aapred = cluster_volumes(shape, c["seed"])
        vols = {"carbon_alpha_probability": torch.from_numpy(ca).cuda(), "backbone_probability": torch.from_numpy(bb).cuda(),
                "amino_acid_probability": torch.from_numpy(aa).cuda(), "amino_acid_prediction": torch.from_numpy(aapred).cuda()}
        pts, cav, bbv = cl.candidate_points(eng, vols, c["thr"])
        labels = np.array(c["labels"])
        assert len(pts) == c["n_points"]
        sums, avgs, val = cl.cluster_scores(eng, bbv, labels)
        assert [float(v) for v in sums] == c["scores_sum"] and [float(v) for v in avgs] == c["scores_avg"] and int(val.sum()) == c["n_valid"]
        cands = cl.nms(eng, cav, pts, val, shape, c["thr"], c["nms_radius"])
        assert cands.tolist() == c["nms_cands"]
        newc, newa, kept = cl.refine(eng, vols, cands)
        assert kept.tolist() == c["kept"] and newc.tolist() == c["CA_cands"] and sha(np.ascontiguousarray(newa.T)) == c["CA_cands_AAProb_sha256"]
        assert [float(v) for v in cl.gather_at(eng, vols["amino_acid_prediction"], np.round(newc).astype(int))] == c["CA_cands_AA"]
        dis, lists, mat = cl.neighbours(eng, vols, newc)            # numpy-2 promotion rules: the golden was evaluated with numpy 2.2.6
        assert sha(dis) == c["cand_self_dis_sha256"] and sha(mat) == c["neigh_mat_sha256"] and int((mat != 0).sum()) == c["neigh_mat_nonzero"]
        best = []
        for i in range(mat.shape[0]):
            second, first = mat[i].argsort()[-2:]
            best.append([int(v) for v in ([first] if mat[i, first] != 0 else []) + ([second] if mat[i, second] != 0 else [])])
        assert best == c["best_neigh"]


# ---- BASELINE configs[1]/[2]: "stride-32 tiles" = grid 32 + 2 x 16 halo, the whole path, one rank and two -------------------
SHAPE32, SEED32 = (70, 50, 40), 93


def _inputs32():
    vol = synth_density(SHAPE32, SEED32)
    af = synth_af(SHAPE32, SEED32, 2e-3)
    af[:, :50] = 0                                           # the windows of the tiles i = 0 end at x = 48: empty AF3 channels
    return vol, af


def test_stride32_tiling_end_to_end_vs_oracle_and_two_ranks(tmp_path, eng, weights):
    """gather (32,16) -> forward -> softmax/argmax -> stitch with the 16-voxel crop, against the CPU oracle (oracle tiles ->
    model_oracle -> oracle stitch) on tiles of every kind (corner, ragged edge, with and without atoms), against a per-batch
    run for all 12 tiles, and against two real rank processes (gloo, both on this GPU) bit for bit."""
    from mica_amd._cabi import tile_table
    from mica_amd.pipeline import VolumePredictor
    from oracle import model_oracle as mo
    vol, af = _inputs32()
    vp = VolumePredictor(eng, 32, 16, batch=4)
    d_vol, d_af = torch.from_numpy(vol).cuda(), torch.from_numpy(af).cuda()
    out = {k: v.cpu().numpy() for k, v in vp.predict_volume(d_vol, d_af).items()}
    tiles, idx = vo.tile_volume(vol, 32, 16)
    af_tiles = np.stack([vo.tile_volume(a, 32, 16)[0] for a in af], axis=1)           # [T, 24, 64, 64, 64]
    assert len(idx) == 3 * 2 * 2 and np.array_equal(tile_table(*SHAPE32, 32), idx)
    has_atoms = af_tiles.reshape(len(idx), -1).any(axis=1)
    assert has_atoms.any() and not has_atoms.all()
    # gather at (32,16) bit-exact for every tile, both volumes
    got_tiles = eng.gather_tiles(d_vol, 32, 16, 0, len(idx)).cpu().numpy()[:, 0]
    assert np.array_equal(got_tiles, tiles)
    # oracle forward on three tiles: first corner, a tile without atoms, the ragged last tile
    picks = [0, int(np.flatnonzero(~has_atoms)[0]), len(idx) - 1]
    rec_ref = np.zeros((len(idx), 23, 64, 64, 64), np.float32)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
    for t in picks:
        lb, lc, la = mo.mica_forward(weights, torch.from_numpy(tiles[t][None, None]), torch.from_numpy(af_tiles[t][None]))
        pb, pc, pa, pp = mo.postprocess(lb, lc, la)
        rec_ref[t, 0], rec_ref[t, 1], rec_ref[t, 2], rec_ref[t, 3:] = pb.numpy()[0], pc.numpy()[0], pp.numpy()[0].astype(np.float32), pa.numpy()[0]
    ref_vol = vo.stitch_volume(rec_ref, idx, SHAPE32, 16)                             # the reference's crop [16:16+di] at this tiling
    for t in picks:
        i, j, k, di, dj, dk = idx[t]
        sl = (slice(i, i + di), slice(j, j + dj), slice(k, k + dk))
        assert np.abs(out["backbone_probability"][sl] - ref_vol[0][sl]).max() < 1e-4, t
        assert np.abs(out["carbon_alpha_probability"][sl] - ref_vol[1][sl]).max() < 1e-4, t
        assert np.abs(out["amino_acid_probability"][(slice(None), *sl)] - ref_vol[3:][(slice(None), *sl)]).max() < 1e-4, t
        top = np.sort(ref_vol[3:][(slice(None), *sl)], axis=0)
        mism = out["amino_acid_prediction"][sl] != ref_vol[2][sl]
        assert not np.any(mism & (top[-1] - top[-2] > 2e-4)) and mism.mean() < 1e-3, t
    # every tile: the stitched volume holds exactly the cropped record of a direct run of its batch
    for first in range(0, len(idx), 4):
        rec = vp.run_batch(d_vol, d_af, first, 4).cpu().numpy()
        for q in range(4):
            i, j, k, di, dj, dk = idx[first + q]
            c = rec[q][:, 16:16 + di, 16:16 + dj, 16:16 + dk]
            assert np.array_equal(out["backbone_probability"][i:i + di, j:j + dj, k:k + dk], c[0])
            assert np.array_equal(out["amino_acid_prediction"][i:i + di, j:j + dj, k:k + dk], c[2])
            assert np.array_equal(out["amino_acid_probability"][:, i:i + di, j:j + dj, k:k + dk], c[3:])
    assert np.abs(out["amino_acid_probability"].sum(axis=0) - 1.0).max() < 1e-5       # every voxel written once
    # two ranks at the same tiling
    res = str(tmp_path / "sharded32.npz")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), res, "x".join(map(str, SHAPE32)), "2",
                                       "32", "16", str(SEED32), "50"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(res)
    for k in out:
        assert np.array_equal(got[k], out[k]), k
    # exactly-once coverage of the sharded stitch: the worker also stitched a counter channel
    assert np.array_equal(got["coverage"], np.ones(SHAPE32, np.float32))
