"""How far does the REFERENCE's own CPU path move when only the summation order changes?  (build container only)

Run:  python oracle/noise_floor.py [--s64]          (needs /root/reference; ~1 min, ~4 min with --s64)

Runs the reference `models.model.MICA` (unmodified, synthetic weights loaded through load_state_dict) on the same inputs
with 1 and with 8 intra-op threads - ATen then splits its reductions differently - and, for the 2-sample case, alone and
inside a batch.  The network is a 25-layer stack of InstanceNorms that amplifies fp32 rounding noise; the figures below
are the floor under any "1e-4 relative per voxel" comparison of this model and are recorded in
tests/golden/manifest.json["noise_floor"] with the same two metrics the GPU tests use:

  scaled   max |a-b| / max(|b|, rms(b))                       (tests/test_gpu_model.py::scaled_err)
  frac_rel fraction of voxels with |a-b| / |b| > 1e-4          (true per-voxel relative error)

With --truth it also runs the reference module in float64 (`MICA().double()`) on the S = 16 inputs: the exact result up to
~1e-15, against which both the reference's own float32 path and the GPU path can be measured.  The float64 logits are
committed as tests/golden/truth64_S16_<weights>.npz and the float32 reference's distance from them is recorded in the manifest;
tests/test_gpu_model.py asserts that the GPU path is not further from the truth than the reference's float32 path is.

TEST INFRASTRUCTURE - documentation of the tolerance; only the --truth fixtures are read by a test.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from mica_amd.synth import synth_af, synth_density       # noqa: E402
from mica_amd.weights import synth_state_dict            # noqa: E402


def metrics(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    scaled = float(np.max(d / np.maximum(np.abs(b), np.sqrt(np.mean(b ** 2)))))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = d / np.abs(b)
    return {"scaled": scaled, "frac_rel_gt_1e-4": float(np.mean(rel > 1e-4)), "maxabs": float(d.max())}


def run(m, x, af, threads):
    torch.set_num_threads(threads)
    with torch.no_grad():
        return [t.numpy().copy() for t in m(x, af)]


def main():
    from models.model import MICA
    out = {"torch": torch.__version__, "cases": {}}
    sets = {"w2022g6": (2022, 6.0), "w7g3": (7, 3.0), "w99g10": (99, 10.0)}
    sizes = ([] if "--truth-only" in sys.argv else [16]) + ([64] if "--s64" in sys.argv else [])
    for S in sizes:
        for tag, (seed, gain) in sets.items():
            if S == 64 and tag != "w2022g6":
                continue
            m = MICA()
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth_state_dict(seed, gain).items()}, strict=True)
            m.eval()
            x = torch.from_numpy(synth_density((1, 1, S, S, S), 12))
            af = torch.from_numpy(synth_af((S, S, S), 12, 0.01 if S == 16 else 1e-3))[None]
            r1 = run(m, x, af, 1)
            r8 = run(m, x, af, 8)
            rec = {n: metrics(a, b) for n, a, b in zip(("bb", "ca", "aa"), r1, r8)}
            if S == 16:
                # the same tile alone and as sample 0 of a batch of two (both with atoms: same AF branch)
                x2 = torch.cat([x, torch.from_numpy(synth_density((1, 1, S, S, S), 13))])
                af2 = torch.cat([af, torch.from_numpy(synth_af((S, S, S), 13, 0.01))[None]])
                rb = run(m, x2, af2, 8)
                rec["batch2_vs_alone"] = {n: metrics(a[:1], b) for n, a, b in zip(("bb", "ca", "aa"), rb, r8)}
            out["cases"][f"S{S}_{tag}_1_vs_8_threads"] = rec
            print(S, tag, json.dumps(rec), flush=True)
    mp = os.path.join(ROOT, "tests", "golden", "manifest.json")
    manifest = json.load(open(mp))
    if "--truth" in sys.argv:
        S = 16
        truth = {}
        for tag, (seed, gain) in sets.items():
            sd = {k: torch.from_numpy(v.copy()) for k, v in synth_state_dict(seed, gain).items()}
            m = MICA()
            m.load_state_dict(sd, strict=True)
            m.eval()
            x = torch.from_numpy(synth_density((1, 1, S, S, S), 12))
            af = torch.from_numpy(synth_af((S, S, S), 12, 0.01))[None]
            r32 = run(m, x, af, 8)
            m64 = MICA().double()
            m64.load_state_dict({k: v.double() for k, v in sd.items()}, strict=True)
            m64.eval()
            with torch.no_grad():
                t64 = [t.numpy().copy() for t in m64(x.double(), af.double())]
            truth[tag] = {n: metrics(a, b) for n, a, b in zip(("bb", "ca", "aa"), r32, t64)}
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"truth64_S16_{tag}.npz"), bb=t64[0], ca=t64[1], aa=t64[2], seed=12, afp=0.01,
                                S=S, wseed=seed, wgain=gain,
                                ref32_scaled=np.array([truth[tag][n]["scaled"] for n in ("bb", "ca", "aa")]))
            print("truth64", tag, json.dumps(truth[tag]), flush=True)
        manifest["reference_float32_vs_float64_truth_S16"] = truth
    if "--truth-only" in sys.argv:
        json.dump(manifest, open(mp, "w"), indent=1)
        return
    manifest["noise_floor"] = out
    json.dump(manifest, open(mp, "w"), indent=1)


if __name__ == "__main__":
    main()
