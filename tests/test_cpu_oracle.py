"""CPU suite: the oracle against the committed golden vectors (which the reference's own code produced,
oracle/gen_golden.py), host-side logic, and the C ABI's symbol table.  No GPU needed."""
import hashlib
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from mica_amd.synth import synth_af, synth_density
from oracle import model_oracle as mo
from oracle import volume_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("S,tag", [(8, "af"), (8, "zeroaf"), (16, "af")])
def test_model_oracle_matches_reference_golden(weights, golden_dir, S, tag):
    g = np.load(os.path.join(golden_dir, f"model_S{S}_{tag}.npz"))
    seed, afp = int(g["seed"]), float(g["afp"])
    x = torch.from_numpy(synth_density((1, 1, S, S, S), seed))
    af = torch.from_numpy(synth_af((S, S, S), seed, afp))[None]
    if tag == "zeroaf":
        af = torch.zeros_like(af)
    torch.set_num_threads(8)
    bb, ca, aa, inter = mo.mica_forward(weights, x, af, return_intermediates=True)
    # same machine class + same ATen kernels => tight; the thread count changes summation order (~3e-5 scaled)
    for got, key in ((bb, "bb"), (ca, "ca"), (aa, "aa")):
        ref = g[key]
        scale = np.maximum(np.abs(ref), np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        assert np.max(np.abs(got.numpy() - ref) / scale) < 1e-4
    if S == 8 and tag == "af":
        for k in ("stem", "enc0", "enc1", "enc2", "fpn"):
            ref = g["inter_" + k]
            assert np.max(np.abs(inter[k].numpy() - ref)) < 1e-4 * max(1.0, float(np.abs(ref).max()))


def test_batchwide_gating_quirk(weights, golden_dir):
    """model.py:60: a zero-AF tile batched with a non-zero one takes the AF branch (SURVEY hard part 4)."""
    g = np.load(os.path.join(golden_dir, "model_S8_batchwide.npz"))
    S = 8
    x = torch.from_numpy(synth_density((2, 1, S, S, S), 21))
    af = torch.stack([torch.zeros(24, S, S, S), torch.from_numpy(synth_af((S, S, S), 21, 0.02))])
    bb, ca, aa = mo.mica_forward(weights, x, af)
    assert np.abs(bb.numpy() - g["bb"]).max() < 1e-3
    pb, _, _ = mo.mica_forward_per_tile(weights, x, af)
    assert float((pb[0] - bb[0]).abs().max()) > 1e-3          # per-tile gating differs on the zero-AF tile
    assert float((pb[1] - bb[1]).abs().max()) < 1e-3


@pytest.mark.parametrize("tag,wseed,gain", [("w2022g6", 2022, 6.0), ("w7g3", 7, 3.0), ("w99g10", 99, 10.0)])
def test_oracle_float64_mode_is_the_references_double_run(golden_dir, tag, wseed, gain):
    """`mica_forward(dtype=torch.float64)` - the oracle's "exact answer" mode, what tools/parity_full_tile.py and the whole-tile GPU test
    compare with on the GPU box - against the reference module run as `MICA().double()` (oracle/noise_floor.py --truth, committed as
    tests/golden/truth64_S16_<weights>.npz): bit for bit.  (At 64^3 oracle/gen_golden_r5.py asserts the same on every logit of the
    sixteen tiles: manifest.json["oracle64_vs_reference64_maxabs"].)"""
    import json
    from mica_amd.synth import synth_af, synth_density
    from mica_amd.weights import synth_state_dict
    g = np.load(os.path.join(golden_dir, f"truth64_S16_{tag}.npz"))
    x = synth_density((1, 1, 16, 16, 16), int(g["seed"]))
    af = synth_af((16, 16, 16), int(g["seed"]), float(g["afp"]))[None]
    out = mo.mica_forward(synth_state_dict(wseed, gain), x, af, dtype=torch.float64)
    for o, k in zip(out, ("bb", "ca", "aa")):
        assert o.dtype == torch.float64 and np.array_equal(o.numpy(), g[k]), k
    m = json.load(open(os.path.join(golden_dir, "manifest.json")))
    assert len(m["oracle64_vs_reference64_maxabs"]) == 16 and max(m["oracle64_vs_reference64_maxabs"].values()) == 0.0
    assert max(m["oracle32_vs_reference32_maxabs_S64"].values()) == 0.0 and m["S64_lattice"] == {"stride": 5, "offset": [1, 2, 3]}


def test_postprocess_properties():
    g = torch.Generator().manual_seed(3)
    bb, ca, aa = (torch.randn((2, c, 4, 4, 4), generator=g) * 3 for c in (4, 4, 21))
    pb, pc, pa, pp = mo.postprocess(bb, ca, aa)
    assert pb.shape == (2, 4, 4, 4) and pa.shape == (2, 20, 4, 4, 4) and pp.dtype == torch.int64
    assert torch.allclose(pa.sum(1), torch.ones(2, 4, 4, 4), atol=1e-6)
    assert torch.equal(pp, aa[:, 1:].argmax(1))
    # class 1 is dropped: changing it must not change the probability (predict.py:342)
    bb2 = bb.clone(); bb2[:, 1] += 100
    assert torch.equal(mo.postprocess(bb2, ca, aa)[0], pb)


def test_tiler_oracle_vs_golden_and_survey_facts(golden_dir):
    tiler = json.load(open(os.path.join(golden_dir, "tiler.json")))
    for key, rec in tiler.items():
        shape = tuple(int(v) for v in key.split("x"))
        tiles, idx = vo.tile_volume(synth_density(shape, rec["seed"]), 48, 8)
        assert idx.tolist() == rec["idx"]
        assert sha(tiles) == rec["tiles_sha256"]
        assert len(idx) == np.prod([-(-s // 48) for s in shape])
    # facts the survey recorded from the reference's own GridCreator
    tv, off = vo.transpose_axes(np.zeros((100, 70, 50), np.float32), 1, 2, 3, [7, 6, 5])
    assert tv.shape == (50, 70, 100) and off == [5.0, 6.0, 7.0]
    _, idx = vo.tile_volume(tv, 48, 8)
    assert len(idx) == 12 and idx[-1].tolist() == [48, 48, 96, 2, 22, 4]
    tv2, off2 = vo.transpose_axes(np.zeros((100, 70, 50), np.float32), 3, 2, 1, [7, 6, 5])
    assert tv2.shape == (100, 70, 50) and off2 == [7.0, 6.0, 5.0]


@pytest.mark.parametrize("shape,grid,pad", [((50, 70, 100), 48, 8), ((96, 96, 96), 48, 8), ((33, 64, 7), 32, 16), ((5, 5, 5), 48, 8)])
def test_tile_stitch_round_trip_and_c_table(shape, grid, pad):
    """tile -> stitch is the identity; the C ABI's host-side tile table equals the oracle's bit for bit."""
    from mica_amd._cabi import tile_table
    vol = synth_density(shape, 9)
    tiles, idx = vo.tile_volume(vol, grid, pad)
    assert np.array_equal(vo.stitch_volume(tiles, idx, shape, pad), vol)
    multi = np.stack([tiles, tiles * 2], axis=1)
    assert np.array_equal(vo.stitch_volume(multi, idx, shape, pad)[1], vol * 2)
    assert np.array_equal(tile_table(*shape, grid), idx)
    # N % grid == 0 edge case: the last window still exists and pad_end = window (create_grids.py:130)
    if shape == (96, 96, 96):
        assert len(idx) == 8 and tiles.shape[1:] == (64, 64, 64)


def test_normaliser_oracle_vs_golden(golden_dir):
    norm = json.load(open(os.path.join(golden_dir, "normaliser.json")))
    for n in ("40", "64"):
        rec = norm[n]
        vol = (synth_density((int(n),) * 3, rec["seed"]) - 0.3) * 3.0
        out, med, pct = vo.normalise_map(vol)
        assert med == rec["median"] and pct == rec["percentile"] and sha(out) == rec["sha256"]
        assert out.dtype == np.float32 and out.min() == 0.0 and out.max() == 1.0
    rec = norm["zoom_20x24x28"]
    out, med, pct = vo.normalise_map(synth_density((20, 24, 28), rec["seed"]) - 0.3, voxel_size=tuple(rec["voxel"]))
    assert list(out.shape) == rec["shape"] and sha(out) == rec["sha256"]
    vol = synth_density((16, 16, 16), norm["nan_16"]["seed"]) - 0.3
    vol[1, 2, 3] = np.nan
    with pytest.raises(ValueError, match="No positive values"):
        vo.normalise_map(vol)


# ---- round 3: goldens produced by the reference's own GridCreator / DataPreprocessor / training tiler, run unmodified under
# I/O-only adapters for `mrcfile` and `Bio` (oracle/gen_golden_r3.py): these pin the restatements to the reference. ----------
def test_tiler_oracle_vs_reference_run_goldens(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "tiler_ref.json")))
    assert len(ref["cases"]) == 20
    for c in ref["cases"]:
        vol = synth_density(tuple(c["shape"]), c["seed"])
        sx, sy, sz = c["starts_xyz"]
        tv, off = vo.transpose_axes(vol, *c["axes"], [sz, sy, sx])
        tiles, idx = vo.tile_volume(tv, c["grid"], c["pad"])
        assert off == c["offset"] and list(tv.shape) == c["meta"]["orig_shape"]
        assert idx.tolist() == c["idx"] and len(idx) == c["count"]
        assert [sha(t) for t in tiles] == c["tile_sha256"] and sha(tiles) == c["tiles_sha256"]
        assert c["meta"]["grid_dtype"] == "float32" and c["meta"]["grid_size"] == c["grid"] and c["meta"]["padding"] == c["pad"]
    w = ref["normalized_map_grids"]
    tv, off = vo.transpose_axes(synth_density(tuple(w["shape"]), w["seed"]), 1, 2, 3, w["starts_xyz"][::-1])
    tiles, idx = vo.tile_volume(tv, 48, 8)
    assert sha(tiles) == w["tiles_sha256"] and w["result"] == {"success": True, "grid_count": len(idx), "offset": off}
    assert w["files"] == sorted(f"normalized_map_grid_i{i}_j{j}_k{k}.npz" for i, j, k, *_ in idx.tolist())
    t = ref["training_tiler"]
    vol = synth_density(tuple(t["shape"]), t["seed"])
    vol[:t["slab"][0]] *= t["slab"][1]
    for key, rec in t["tilings"].items():
        grid, pad = (int(v) for v in key.split("_"))
        tiles, idx = vo.tile_volume(vol, grid, pad)                     # no transpose in the training tilers
        keep = [i for i in range(len(idx)) if tiles[i].max() >= 0.01]
        assert len(idx) == rec["all"] and len(keep) == rec["count"] < rec["all"] and sha(tiles[keep]) == rec["tiles_sha256"]


def _normaliser_ref_inputs():
    from oracle.gen_golden_r3 import _norm_inputs      # input builders only (seeded arrays); nothing of the reference
    return _norm_inputs()


def test_normaliser_oracle_vs_reference_run_goldens(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "normaliser_ref.json")))
    inputs = _normaliser_ref_inputs()
    assert set(inputs) == set(ref["cases"])
    for name, rec in ref["cases"].items():
        vol, voxel = inputs[name]
        assert str(vol.dtype) == rec["dtype"] and list(voxel) == rec["voxel"]
        if not rec["written"]:                                           # the reference logged a failure and wrote nothing
            with pytest.raises((ValueError, RuntimeError)):
                vo.normalise_map(vol, voxel_size=voxel)
            continue
        out, med, pct = vo.normalise_map(vol, voxel_size=voxel)
        assert out.dtype == np.float32 and list(out.shape) == rec["out_shape"]
        assert med == rec["median"] and pct == rec["percentile"] and sha(out) == rec["sha256"], name
        assert np.array_equal(out[::3, ::3, ::3], np.load(os.path.join(golden_dir, f"normaliser_ref_{name}.npy")))
    assert not ref["cases"]["f32_nan_16"]["written"] and not ref["cases"]["f16_16"]["written"]


def test_af3_oracle_vs_reference_run_goldens(golden_dir):
    from oracle import af3_oracle as ao
    ref = json.load(open(os.path.join(golden_dir, "af3_ref.json")))
    assert sum(c["success"] for c in ref["cases"]) == 3 and len(ref["cases"]) == 5
    for c in ref["cases"]:
        flat = [(a[0], r[1], a[1:4]) for r in c["atoms"] if r[2] == " " for a in r[3]]
        args = (np.array([f[2] for f in flat], np.float32), [f[0] for f in flat], [f[1] for f in flat], c["origin"], tuple(c["shape"]))
        if not c["success"]:                                             # the reference's scatter raised IndexError (non-cubic map)
            with pytest.raises(IndexError):
                ao.rasterise_atoms(*args)
            continue
        got = ao.rasterise_atoms(*args)
        assert sha(got) == c["sha256"] and int(got.sum()) == c["ones"] and len(flat) == c["n_atoms"]


def _best_neigh(neigh_mat):
    """modeler.py:890-899 (host numpy in the caller)."""
    out = []
    for cand in range(neigh_mat.shape[0]):
        second, first = neigh_mat[cand].argsort()[-2:]
        out.append([int(v) for v in ([first] if neigh_mat[cand, first] != 0 else []) + ([second] if neigh_mat[cand, second] != 0 else [])])
    return out


def test_cluster_oracle_vs_reference_run_goldens(golden_dir):
    """Solver.clustering (utils/modeler.py:762-899) run by the reference itself on duck-typed state, DBSCAN labels from the golden."""
    from oracle import cluster_oracle as co
    from oracle.gen_golden_r3 import cluster_volumes
    ref = json.load(open(os.path.join(golden_dir, "cluster_ref.json")))
    assert len(ref["cases"]) == 2
    for c in ref["cases"]:
        ca, bb, aa, aapred = cluster_volumes(tuple(c["shape"]), c["seed"])
        pts = co.threshold_points(ca, c["thr"])
        labels = np.array(c["labels"])
        assert len(pts) == c["n_points"] == len(labels)
        sums, avgs, val = co.cluster_scores(bb, pts, labels)
        assert [float(v) for v in sums] == c["scores_sum"] and [float(v) for v in avgs] == c["scores_avg"] and int(val.sum()) == c["n_valid"]
        cands = np.array(co.nms(co.sorted_pred_list(ca, pts, val).copy(), c["thr"], c["nms_radius"]))
        assert cands.tolist() == c["nms_cands"]
        newc, newa, kept = co.refine_candidates(ca, aa, cands)
        assert kept.tolist() == c["kept"] and newc.tolist() == c["CA_cands"] and sha(newa.T) == c["CA_cands_AAProb_sha256"]
        assert [float(v) for v in co.gather(aapred, np.round(newc).astype(int))] == c["CA_cands_AA"]
        dis, lists, mat = co.neighbour_matrix(newc, bb)
        assert sha(dis) == c["cand_self_dis_sha256"] and sha(mat) == c["neigh_mat_sha256"] and _best_neigh(mat) == c["best_neigh"]


def test_weights_table_and_generator_are_stable(weights):
    from mica_amd.weights import param_shapes, synth_state_dict
    shapes = param_shapes()
    assert len(shapes) == 125 and sum(int(np.prod(s)) for s in shapes.values()) == 14127813
    again = synth_state_dict(2022)
    h = hashlib.sha256()
    for k in shapes:
        assert weights[k].dtype == np.float32 and weights[k].shape == shapes[k]
        assert np.array_equal(weights[k], again[k])
        h.update(weights[k].tobytes())
    # pins the generator across hosts / numpy versions: golden vectors depend on it
    assert h.hexdigest()[:16] == WEIGHTS_SHA16


WEIGHTS_SHA16 = "14f1324fa67c233e"


def test_checkpoint_loader_round_trip(tmp_path, weights):
    from mica_amd.weights import load_checkpoint_state_dict
    p = tmp_path / "ck.pth"
    torch.save({"epoch": 3, "model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in weights.items()},
                "val_loss": 0.1}, p)                         # reference train.py:298-304 format
    sd = load_checkpoint_state_dict(str(p))
    assert set(sd) == set(weights) and all(np.array_equal(sd[k], weights[k]) for k in weights)
    bad = dict(weights); bad.pop("fpn.weights")
    torch.save({"model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in bad.items()}}, p)
    with pytest.raises(KeyError):
        load_checkpoint_state_dict(str(p))


def test_cabi_exports_every_declared_symbol():
    """include/mica_hip.h <-> ctypes table <-> the built .so (no compute calls: no GPU here)."""
    from mica_amd import _cabi
    hdr = open(os.path.join(ROOT, "include", "mica_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mica_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_cabi.SIGNATURES), declared ^ set(_cabi.SIGNATURES)
    lib = _cabi.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mica_abi_version() == _cabi.ABI_VERSION
    assert lib.mica_tile_count(512, 512, 512, 48) == 1331 and lib.mica_tile_count(256, 256, 256, 32) == 512
    assert lib.mica_tile_count(0, 1, 1, 48) < 0


def test_product_fails_loudly_without_gpu():
    from mica_amd.engine import Engine, MicaHipError
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(MicaHipError, match="no CPU fallback"):
        Engine(0)
    from mica_amd.model import MICA
    with pytest.raises(MicaHipError):
        MICA().to("cpu")


def test_missing_library_is_an_error(tmp_path):
    from mica_amd import _cabi
    with pytest.raises(_cabi.MicaHipError, match="not found"):
        _cabi.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    for fn in os.listdir(os.path.join(ROOT, "mica_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "mica_amd", fn)).read()
            assert "oracle" not in src, fn


def test_inline_asm_weight_prefetch_is_hazard_free(tmp_path):
    """conv_wino16_kernel fetches its weight fragments with inline-asm loads and hand-counted s_waitcnt.  hipcc does not
    model those loads, so the emitted code is audited: between an asm load and the wait that retires it no other
    instruction may touch its destination registers (tools/audit_asm_loads.py); cross-compiles without a GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "mica_amd", "csrc", "kernels_conv.hip")
    asm = str(tmp_path / "kernels_conv.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, src],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read()
    # the persistent 16x16x32 variants: the audit models their LDS-DMA instructions as occupying queue slots.  Per chunk and wave the
    # 128-channel variant walks 14 steps (448 MFMAs; 9 waits that leave the previous step's one DMA in flight, 5 + 1 plain
    # ones), the 64-channel variant 7 steps (224 MFMAs; previous step issued 2, 1 or 0 DMAs), the 32-channel variant 7 steps of
    # two column tiles (112 MFMAs, two weight loads per step); one copy of the chunk body each
    for bn, n_mfma, waits in ((128, 448, {5: 9, 4: 6}), (64, 224, {6: 3, 5: 3, 4: 2}), (32, 112, {4: 3, 3: 3, 2: 2})):
        m = re.search(r"^_ZN4mica18conv_wino16_kernelILi%dE.*?s_endpgm" % bn, text, flags=re.S | re.M)
        assert m, bn
        part = str(tmp_path / ("wino16_%d.s" % bn))
        open(part, "w").write(m.group(0))
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py"), part], text=True)
        assert "violations: 0" in out, out[-2000:]
        body = m.group(0)
        assert body.count("v_mfma_f32_16x16x32_f16") == n_mfma
        for cnt, times in waits.items():
            assert body.count("s_waitcnt vmcnt(%d)" % cnt) == times, (bn, cnt)
        assert body.count("global_load_lds_dwordx4") == 18 and "scratch_" not in body
    # no kernel of the file spills (a spilled asm destination would be reloaded/stored around in-flight loads)
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    assert spills and max(spills) == 0, spills


def test_stem_mfma_kernel_keeps_its_register_and_lds_budget(tmp_path):
    """stem_mfma_kernel (kernels_stem.hip) shares a CU between two 512-thread workgroups: that needs <= 128 VGPRs per wave, no scratch
    (26 spilled registers once cost 9 % of the kernel) and <= 80 KB of LDS; one copy of the K-step body (48 MFMAs, 16 fragment
    reads).  Cross-compiles without a GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "mica_amd", "csrc", "kernels_stem.hip")
    asm = str(tmp_path / "kernels_stem.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, src],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read()
    m = re.search(r"^_ZN4mica16stem_mfma_kernelE.*?s_endpgm", text, flags=re.S | re.M)
    assert m
    body = m.group(0)
    assert body.count("v_mfma_f32_16x16x32_f16") == 48 and "scratch_" not in body
    lds = int(re.search(r"\.amdhsa_kernel _ZN4mica16stem_mfma_kernelE.*?\.amdhsa_group_segment_fixed_size (\d+)", text, flags=re.S).group(1))
    meta = re.search(r"\.name:\s+_ZN4mica16stem_mfma_kernelE.*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?"
                     r"\.vgpr_spill_count:\s+(\d+)", text, flags=re.S)
    assert meta, "kernel metadata not found"
    scratch, vgprs, spilled = (int(v) for v in meta.groups())
    assert lds > 50 * 1024, lds
    assert lds <= 80 * 1024 and scratch == 0 and vgprs <= 128 and spilled == 0, (lds, scratch, vgprs, spilled)


def test_inline_asm_weight_prefetch_of_the_f43_kernel_is_hazard_free(tmp_path):
    """conv_wino43_kernel (kernels_conv43.hip): the same audit.  Per chunk and wave 14 steps of 16 MFMAs, ten weight-fragment sets in
    three register sets (40 asm loads), six slab DMAs, 36 A-fragment reads; the waits a step places leave exactly the fragments
    requested since and the DMAs issued since in flight; one copy of the chunk body, no spill anywhere in the file."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "mica_amd", "csrc", "kernels_conv43.hip")
    asm = str(tmp_path / "kernels_conv43.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, src],
                          stderr=subprocess.DEVNULL)
    text = open(asm).read()
    # <128>: 14 steps per chunk and wave; <64> (round 5, the tap-split variant): 7 steps, five fragment sets in the same three register
    # sets (20 asm loads), the same six slab DMAs, 20 A-fragment reads
    for bn, mfmas, loads_want, reads_want in ((128, 224, 40, (36, 37, 38, 39)), (64, 112, 20, (20, 21, 22, 23))):
        m = re.search(r"^_ZN4mica18conv_wino43_kernelILi%dEE.*?s_endpgm" % bn, text, flags=re.S | re.M)
        assert m, bn
        body = m.group(0)
        part = str(tmp_path / f"wino43_{bn}.s")
        open(part, "w").write(body)
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py"), part], text=True)
        assert "violations: 0" in out, out[-2000:]
        assert body.count("v_mfma_f32_16x16x32_f16") == mfmas and "scratch_" not in body
        # the chunk loop: from the first to the last MFMA
        lines = body.split("\n")
        idx = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32_f16" in l]
        loop = "\n".join(lines[idx[0] - 120:idx[-1] + 1])
        assert loop.count("global_load_lds_dwordx4") == 6
        assert loop.count("ds_read_b128") + loop.count("ds_load_b128") in reads_want      # the first few may sit above the window
        loads = len(re.findall(r"global_load_dwordx4", loop))
        assert loads == loads_want, (bn, loads)
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    assert spills and max(spills) == 0, spills


@pytest.mark.parametrize("shape,factors", [((6, 7, 5), (1.5, 1.25, 0.8)), ((9, 4, 11), (0.83, 0.83, 0.83)), ((5, 5, 5), (1.0, 1.0, 1.0)),
                                           ((4, 1, 8), (47.0, 1.0, 0.5)), ((12, 10, 3), (1.07, 2.0, 1.3))])
def test_zoom_restatement_is_bit_exact_vs_scipy(shape, factors):
    """oracle.zoom_cubic restates scipy.ndimage.zoom(order=3) - the reference's resampler (preprocessing.py:117) - and
    must agree with the installed scipy to the last bit, including the (4 -> 188) edge quirk where the last
    coordinate rounds past the edge and scipy returns cval."""
    from scipy.ndimage import zoom
    x = (synth_density(shape, 17) - 0.3).astype(np.float32)
    ref = zoom(x, factors, order=3)
    got = vo.zoom_cubic(x, factors)
    assert got.dtype == np.float32 and got.shape == ref.shape and np.array_equal(got, ref)
    if shape == (4, 1, 8):
        assert got.shape[0] == 188 and np.all(got[-1] == 0.0)      # scipy quirk reproduced


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_golden_generator_r3_check():
    """The committed reference-run fixtures can be regenerated from the tree: `gen_golden_r3.py --check` runs the
    reference's own tiler / normaliser / AF3 rasteriser / Solver.clustering again and compares bit for bit (≈ 15 s).
    Guards the generator itself (round 3 shipped one whose normaliser and af3 legs crashed)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_r3.py"), "--check"],
                       capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "0 differences (bit-exact)" in r.stdout, tail


def test_check_mode_detects_a_changed_fixture(tmp_path, golden_dir):
    """oracle/_check.py: equal files pass, a flipped bit / a changed JSON leaf / a missing twin are reported."""
    import shutil
    from oracle._check import compare_dirs
    prod = tmp_path / "p"
    prod.mkdir()
    for f in ("tiler_ref.json", "normaliser_ref_f32_40.npy"):
        shutil.copy(os.path.join(golden_dir, f), prod / f)
    names, log = compare_dirs(str(prod), golden_dir, exact=True)
    assert len(names) == 2 and log == []
    a = np.load(prod / "normaliser_ref_f32_40.npy")
    a.reshape(-1)[5] = np.nextafter(a.reshape(-1)[5], 2.0)
    np.save(prod / "normaliser_ref_f32_40.npy", a)
    j = json.load(open(prod / "tiler_ref.json"))
    k = next(iter(j))
    j[k] = {"changed": 1}
    json.dump(j, open(prod / "tiler_ref.json", "w"))
    (prod / "extra.json").write_text("{}")
    _, log = compare_dirs(str(prod), golden_dir, exact=True)
    assert len(log) >= 3, log
    _, log2 = compare_dirs(str(prod), golden_dir, exact=False, tol=1e-4)      # one ulp is inside the float tolerance
    assert not any("normaliser_ref_f32_40" in line for line in log2)


@pytest.mark.parametrize("kind", ["heavy", "blob"])
def test_model_oracle_matches_reference_on_stress_goldens(golden_dir, kind):
    """Round-4 goldens (reference MICA on heavy-tailed weights / a normaliser-shaped map, oracle/gen_golden_r4.py): the oracle
    reproduces the reference's float32 logits, and the generators of those inputs are what the fixture metadata says."""
    from mica_amd.synth import stress_case
    g = np.load(os.path.join(golden_dir, f"r4_{kind}_S16.npz"))
    w, x, af = stress_case(kind, 16)
    if kind == "blob":
        assert (x == 0).mean() > 0.5 and x.max() == 1.0 and af.sum() > 0
    else:
        k = np.abs(w["encoder.1.dense_block.conv3.0.weight"]).reshape(128, -1).max(1)
        assert np.sort(k)[-1] > 20 * np.median(k)                      # outlier output channels
    torch.set_num_threads(8)
    out = mo.mica_forward(w, torch.from_numpy(x), torch.from_numpy(af))
    for got, key in zip(out, ("bb", "ca", "aa")):
        ref = g[key]
        scale = np.maximum(np.abs(ref), np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        assert np.max(np.abs(got.numpy() - ref) / scale) < 1e-4


def test_whole_network_winograd_emulation_supports_the_shipped_choice(golden_dir):
    """oracle/wino_network.py (the emulation that decided which layers run on Winograd F(4,3), DESIGN.md section 4 "Round 4") on one
    weight set: the shipped choice - F(4,3) with the balanced points on encoder.2, F(2,3) elsewhere - stays inside the 1e-4 bar
    against the reference's float32 logits and against the float64 truth, and F(4,3) everywhere with the textbook points is
    measurably worse (rms distance from the truth)."""
    from oracle import wino_network as wn
    torch.set_num_threads(8)
    e32, e64, ref, rms, ref_rms = wn.run_case("f43s@e2", "w2022g6", "model_S16_af.npz", 2022, 6.0, golden_dir)
    assert max(e32) < 1e-4 and max(e64) < 1e-4
    _, _, _, rms_all, _ = wn.run_case("f43", "w2022g6", "model_S16_af.npz", 2022, 6.0, golden_dir)
    assert rms_all > 1.3 * rms and rms < 1.5 * ref_rms


def test_stale_library_is_refused_with_a_rebuild_message(monkeypatch):
    """the library is git-ignored, so a checkout can meet an older build; the ABI version check says 'rebuild' instead of
    failing later on a missing symbol or a changed struct."""
    from mica_amd import _cabi
    monkeypatch.setattr(_cabi, "ABI_VERSION", _cabi.ABI_VERSION + 1)
    with pytest.raises(_cabi.MicaHipError, match="rebuild"):
        _cabi.load_library(_cabi.LIB_PATH)


def test_bench_refuses_counter_figures_of_other_sources(tmp_path, monkeypatch):
    """bench.py quotes PMC traffic / MFMA-busy figures from committed profiles only when they were collected on the
    library sources that are running (mica_amd/_cabi.py::source_hash, stamped by tools/profile.sh)."""
    import bench
    from mica_amd._cabi import source_hash
    h = source_hash()
    assert len(h) == 16 and h == source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "r09_pmc_traffic.json").write_text(json.dumps({"library_source_hash": h, "kernels": {"conv_wino43_kernel": {"hbm_bytes": 5.0}}}))
    (prof / "r09_pmc_sq_summary.txt").write_text(
        "kernel                       calls     avg us clock GHz   mfma%  wait_any wait_inst    active  lds_act%  bank_cf% valu/wave-cyc\n"
        "conv_wino43_kernel              20    12100.0      1.69    70.7     0.316     0.497     0.187      17.0      1.42      0.085\n"
        "depthwise_kernel<4, 8>          15      593.7      1.98     0.0     0.355     0.219     0.426      37.9      1.17      0.283\n"
        "mfma% = note line\nlibrary_source_hash: " + h + "\n")
    t, note = bench.load_traffic(h, True)
    assert t["conv_wino43_kernel"]["hbm_bytes"] == 5.0 and note == "r09_pmc_traffic.json"
    sq, note = bench.load_sq_summary(h, True)
    assert sq["conv_wino43_kernel"] == {"calls": 20.0, "avg_us": 12100.0, "clock_ghz": 1.69, "mfma_busy": 70.7 / 100.0}
    assert sq["depthwise_kernel<4, 8>"]["avg_us"] == 593.7
    t, note = bench.load_traffic("0" * 16, True)
    assert t == {} and "REFUSED" in note
    sq, note = bench.load_sq_summary("0" * 16, True)
    assert sq == {} and "REFUSED" in note


def test_committed_counter_profiles_belong_to_these_library_sources():
    """The newest profiles/rNN_pmc_traffic.json / rNN_pmc_sq_summary.txt - what bench.py quotes `roofline.traffic`, `mfma_busy` and
    `held_clock_ghz` from - were collected on exactly the library sources in this tree (a kernel change without a re-profile makes
    bench.py report null for them: this test is the reminder to run tools/profile.sh again), and the committed bench line's hardware
    fraction agrees with the counters' MFMA-busy x held clock / 2.4 GHz within 5 %."""
    import bench
    from mica_amd._cabi import source_hash
    h = source_hash()
    t, tnote = bench.load_traffic(h, True)
    sq, sqnote = bench.load_sq_summary(h, True)
    assert t and "REFUSED" not in tnote, tnote
    assert sq and "REFUSED" not in sqnote, sqnote
    assert t["conv_wino43_kernel<128>"]["hbm_bytes"] > 1e9 and sq["conv_wino43_kernel<128>"]["mfma_busy"] > 0.3
    line = json.load(open(os.path.join(ROOT, "profiles", bench._newest("r[0-9][0-9]_pmc_traffic.json")[-20:-17] + "_bench_default.json")))
    r = line["roofline"]
    assert r["profile_source"]["library_source_hash"] == h
    assert 0.0 < r["frac"] <= 1.0 and r["peak"] == 2500.0 and r["traffic"] is not None
    assert abs(r["frac"] - r["mfma_busy_x_clock_over_2p4"]) / r["frac"] < 0.05
    assert line["sustained"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
