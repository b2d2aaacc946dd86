"""Generate tests/golden/* by running the REFERENCE's own Python (build container only).

Run:  python oracle/gen_golden.py [--check]  (needs /root/reference; ~5 min on 8 cores)
      --check regenerates into a scratch directory and compares with tests/golden/ (oracle/_check.py)

What is imported from the reference, unmodified: models.model.MICA, dataset.dataset,
utils.predict.CryoEMPredictor.  (utils/create_grids.py, utils/preprocessing.py and utils/modeler.py
are run by oracle/gen_golden_r3.py, round 3, under I/O-only adapters for the packages this image
lacks; tiler.json / normaliser.json written HERE are the older restatement-only fixtures and are
kept as additional regression data.)

Only arrays (inputs regenerated from seeds, outputs stored) are committed; no reference
source travels.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

from mica_amd.weights import synth_state_dict            # noqa: E402
from mica_amd.synth import synth_density, synth_af       # noqa: E402
from oracle import model_oracle as mo                    # noqa: E402
from oracle import volume_oracle as vo                   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED_W = 2022


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_model():
    from models.model import MICA
    m = MICA()
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth_state_dict(SEED_W).items()}
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def main():
    from oracle._check import CheckRun
    with CheckRun(globals(), sys.argv[1:], exact=False):   # float32 network outputs: scaled tolerance, see oracle/_check.py
        _generate()


def _generate():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    manifest = {"numpy": np.__version__, "torch": torch.__version__, "weights_seed": SEED_W,
                "oracle_vs_reference_maxabs": {}}
    import scipy
    manifest["scipy"] = scipy.__version__
    m = ref_model()
    w = synth_state_dict(SEED_W)

    # ---- model goldens at small S (whole tensors) ------------------------------------
    for S, seed, afp in ((8, 11, 0.02), (16, 12, 0.01)):
        x = torch.from_numpy(synth_density((1, 1, S, S, S), seed))
        af = torch.from_numpy(synth_af((S, S, S), seed, afp))[None]
        for tag, a in (("af", af), ("zeroaf", torch.zeros_like(af)), ("noneaf", None)):
            with torch.no_grad():
                rb, rc, ra = m(x, a)
            ob, oc, oa, inter = mo.mica_forward(w, x, a, return_intermediates=True)
            d = max(float((rb - ob).abs().max()), float((rc - oc).abs().max()), float((ra - oa).abs().max()))
            manifest["oracle_vs_reference_maxabs"][f"model_S{S}_{tag}"] = d
            assert d < 1e-5, (S, tag, d)
            if tag == "noneaf":
                continue
            rec = {"bb": rb.numpy(), "ca": rc.numpy(), "aa": ra.numpy(),
                   "seed": seed, "afp": afp, "S": S}
            if S == 8 and tag == "af":
                # reference intermediates via forward hooks on the reference modules
                got = {}
                hooks = [m.input_processing.register_forward_hook(lambda _m, _i, o: got.__setitem__("stem", o)),
                         m.fpn.register_forward_hook(lambda _m, _i, o: got.__setitem__("fpn", o))]
                for e in range(3):
                    hooks.append(m.encoder[e].register_forward_hook(
                        lambda _m, _i, o, e=e: got.__setitem__(f"enc{e}", o)))
                with torch.no_grad():
                    m(x, a)
                for h in hooks:
                    h.remove()
                for k, v in got.items():
                    dd = float((v - inter[k]).abs().max())
                    manifest["oracle_vs_reference_maxabs"][f"S8_{k}"] = dd
                    assert dd < 1e-5, (k, dd)
                    rec["inter_" + k] = v.numpy()
            np.savez_compressed(os.path.join(OUT, f"model_S{S}_{tag}.npz"), **rec)
            print("model", S, tag, "oracle-vs-ref", d, flush=True)

    # ---- batch-wide AF gating quirk (model.py:60): batch of [zero-AF tile, AF tile] ---
    S = 8
    x2 = torch.from_numpy(synth_density((2, 1, S, S, S), 21))
    af2 = torch.stack([torch.zeros(24, S, S, S), torch.from_numpy(synth_af((S, S, S), 21, 0.02))])
    with torch.no_grad():
        rb, rc, ra = m(x2, af2)
    ob, oc, oa = mo.mica_forward(w, x2, af2)
    d = max(float((rb - ob).abs().max()), float((ra - oa).abs().max()))
    manifest["oracle_vs_reference_maxabs"]["batchwide_S8"] = d
    assert d < 1e-5
    np.savez_compressed(os.path.join(OUT, "model_S8_batchwide.npz"), bb=rb.numpy(), ca=rc.numpy(),
                        aa=ra.numpy(), seed=21, afp=0.02, S=S)

    # ---- one full 64^3 tile: strided subsample + stats -------------------------------
    S = 64
    x = torch.from_numpy(synth_density((1, 1, S, S, S), 31))
    af = torch.from_numpy(synth_af((S, S, S), 31, 1e-3))[None]
    t0 = time.time()
    with torch.no_grad():
        rb, rc, ra = m(x, af)
    t_ref = time.time() - t0
    t0 = time.time()
    ob, oc, oa = mo.mica_forward(w, x, af)
    t_or = time.time() - t0
    d = max(float((rb - ob).abs().max()), float((rc - oc).abs().max()), float((ra - oa).abs().max()))
    manifest["oracle_vs_reference_maxabs"]["model_S64_af"] = d
    manifest["ref_forward_S64_seconds_8threads"] = t_ref
    manifest["oracle_forward_S64_seconds_8threads"] = t_or
    assert d < 1e-4, d
    from utils.predict import CryoEMPredictor  # reference softmax/argmax lines restated in mo.postprocess
    pb, pc, pa, pp = mo.postprocess(rb, rc, ra)
    # reference post-processing lines (utils/predict.py:342-349) executed literally:
    sm = torch.nn.Softmax(dim=1)
    rbb = sm(torch.cat((rb[:, :1], rb[:, 2:]), dim=1)); rcc = sm(torch.cat((rc[:, :1], rc[:, 2:]), dim=1))
    raa = sm(ra[:, 1:, :, :, :]); rpp = torch.max(raa, 1)[1]
    assert torch.equal(rbb[:, 2], pb) and torch.equal(rcc[:, 2], pc) and torch.equal(raa, pa) and torch.equal(rpp, pp)
    from oracle._check import LATTICE_OFFSET, LATTICE_STRIDE, lattice
    np.savez_compressed(os.path.join(OUT, "model_S64_af_sub.npz"),
                        bb=lattice(rb.numpy()), ca=lattice(rc.numpy()), aa=lattice(ra.numpy()),
                        bb_prob=lattice(pb.numpy()), ca_prob=lattice(pc.numpy()),
                        aa_prob=lattice(pa.numpy()), aa_pred=lattice(pp.numpy()).astype(np.uint8),
                        mean=np.array([rb.mean(), rc.mean(), ra.mean()], dtype=np.float64),
                        std=np.array([rb.std(), rc.std(), ra.std()], dtype=np.float64),
                        seed=31, afp=1e-3, S=64, stride=LATTICE_STRIDE, offset=np.array(LATTICE_OFFSET))
    print("model 64 oracle-vs-ref", d, "ref s", t_ref, "oracle s", t_or, flush=True)

    # ---- predictor end to end: 60x40x40 map = 2 tiles, reference CryoEMPredictor -----
    tmp = tempfile.mkdtemp(prefix="mica_golden_")
    try:
        shape = (60, 40, 40)
        vol = synth_density(shape, 41)
        tiles, idx = vo.tile_volume(vol, 48, 8)
        gdir = os.path.join(tmp, "grids", "normalized_map_grids")
        os.makedirs(gdir)
        for t, (i, j, k, di, dj, dk) in enumerate(idx):
            # the reference tile file format (utils/create_grids.py:159-174)
            np.savez(os.path.join(gdir, f"normalized_map_grid_i{i}_j{j}_k{k}.npz"), grid=tiles[t],
                     i=i, j=j, k=k, di=di, dj=dj, dk=dk, orig_shape=shape, grid_size=48, padding=8,
                     voxel_size=np.array([1.0, 1.0, 1.0]), origin=np.zeros(3), mapc=1, mapr=2, maps=3)
        ck = os.path.join(tmp, "ckpt.pth")
        torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v.copy())
                                                     for k, v in w.items()}}, ck)
        pred = CryoEMPredictor(model_path=ck, grids_path=os.path.join(tmp, "grids") + "/",
                               output_path=os.path.join(tmp, "out"), save_output=False, device="cpu", quiet=True)
        t0 = time.time()
        ok, vols = pred.run_prediction()
        manifest["ref_predictor_2tiles_seconds"] = time.time() - t0
        assert ok and set(vols) == {"backbone_probability", "carbon_alpha_probability",
                                    "amino_acid_prediction", "amino_acid_probability"}
        manifest["predictor_dtypes"] = {k: str(v.dtype) for k, v in vols.items()}
        manifest["predictor_shapes"] = {k: list(v.shape) for k, v in vols.items()}
        np.savez_compressed(os.path.join(OUT, "predictor_60x40x40.npz"),
                            backbone_probability=vols["backbone_probability"],
                            carbon_alpha_probability=vols["carbon_alpha_probability"],
                            amino_acid_prediction=vols["amino_acid_prediction"].astype(np.uint8),
                            amino_acid_probability_sub=vols["amino_acid_probability"][:, ::2, ::2, ::2],
                            aa_prob_sum_sha=sha(vols["amino_acid_probability"]),
                            seed=41, shape=np.array(shape))
        print("predictor ok", manifest["ref_predictor_2tiles_seconds"], flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    # ---- tiler: index tables + tile hashes from the restatement; survey-recorded facts --
    tiler = {}
    for shape in ((50, 70, 100), (96, 96, 96), (100, 100, 100), (60, 40, 40), (7, 130, 48)):
        vol = synth_density(shape, 51)
        tiles, idx = vo.tile_volume(vol, 48, 8)
        tiler["x".join(map(str, shape))] = {"idx": idx.tolist(), "tiles_sha256": sha(tiles), "seed": 51}
    # facts recorded by the survey from the reference's own GridCreator:

    tv, off = vo.transpose_axes(np.zeros((100, 70, 50), np.float32), 1, 2, 3, [7, 6, 5])
    assert tv.shape == (50, 70, 100) and off == [5.0, 6.0, 7.0]
    _, idx = vo.tile_volume(tv, 48, 8)
    assert len(idx) == 12 and idx[-1].tolist() == [48, 48, 96, 2, 22, 4]
    tv2, off2 = vo.transpose_axes(np.zeros((100, 70, 50), np.float32), 3, 2, 1, [7, 6, 5])
    assert tv2.shape == (100, 70, 50) and off2 == [7.0, 6.0, 5.0]
    json.dump(tiler, open(os.path.join(OUT, "tiler.json"), "w"))

    # ---- normaliser: the reference's numpy/scipy calls on synthetic volumes ------------
    norm = {}
    for n, seed in ((40, 61), (64, 62)):
        vol = (synth_density((n, n, n), seed) - 0.3) * 3.0
        out, med, pct = vo.normalise_map(vol)
        norm[str(n)] = {"seed": seed, "median": med, "percentile": pct, "sha256": sha(out),
                        "sub": out[::8, ::8, ::8].tolist()}
    # a NaN voxel: scipy's cubic-spline prefilter (zoom order=3) is an IIR filter, so the NaN
    # spreads over the volume, nan_to_num zeroes it and the reference reports failure (:163-165)
    vol = (synth_density((16, 16, 16), 64) - 0.3)
    vol[1, 2, 3] = np.nan
    try:
        vo.normalise_map(vol)
        norm["nan_16"] = {"seed": 64, "raises": False}
    except ValueError as e:
        norm["nan_16"] = {"seed": 64, "raises": True, "message": str(e)}
    vol = (synth_density((20, 24, 28), 63) - 0.3)
    out, med, pct = vo.normalise_map(vol, voxel_size=(1.5, 1.25, 0.8))
    norm["zoom_20x24x28"] = {"seed": 63, "voxel": [1.5, 1.25, 0.8], "shape": list(out.shape), "median": med,
                             "percentile": pct, "sha256": sha(out)}
    json.dump(norm, open(os.path.join(OUT, "normaliser.json"), "w"))

    mp = os.path.join(OUT, "manifest.json")
    if os.path.exists(mp):                               # the later generators (r2, r4, r5, noise_floor) keep their records in the same file
        old = json.load(open(mp))
        old.update(manifest)
        manifest = old
    json.dump(manifest, open(mp, "w"), indent=1)
    print(json.dumps({k: v for k, v in manifest.items() if k not in ("S64", "noise_floor")}, indent=1))


if __name__ == "__main__":
    main()
