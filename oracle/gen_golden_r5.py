"""Round-5 goldens: the production tile size against the exact answer (build container only).

Run:  python oracle/gen_golden_r5.py [--check] [case ...]      (needs /root/reference; ~4 min per case on 8 cores)

Every golden family that has a 64^3 fixture (three uniform weight sets, the zero-AF branch, the two round-4 stress
families) plus FOUR more input seeds of the family that sits closest to the 1e-4 bar (weights seed 99 / gain 10, AF path), two more
input seeds for each of the other two uniform weight sets and two further weight seeds (mica_amd/synth.py::CASES64: sixteen tiles) goes
through the reference's own `models.model.MICA` (reference models/model.py:331-348, imported unmodified) three times on ONE 64^3
tile:

  float32, 8 intra-op threads     the reference CPU path as the other goldens record it
  float32, 1 intra-op thread      the same arithmetic with ATen's reductions split differently: the reference's own noise floor
  float64 (`MICA().double()`)     the exact answer up to ~1e-15

Written per case: tests/golden/truth64_S64_sub_<case>.npz = the float64 logits on the lattice the other 64^3 fixtures use
(oracle/_check.py::lattice: z = 1, y = 2, x = 3 (mod 5), coprime to every kernel's output tile; rounds 2-5 kept [::4, ::4, ::4], one
position of every such tile) (bb64 / ca64 / aa64), the float32 logits on the same lattice (bb / ca / aa), and the reference float32
path's own distances from the truth on that lattice (`ref32_scaled`, `ref32_rms`, `ref32_frac_rel` = its fraction of voxels beyond
1e-4 literal relative error; `floor_scaled` / `floor_frac_rel` = scaled max difference and that fraction between its 1-thread and
8-thread runs) and on the WHOLE tile (`*_full`, round 6: what a whole-tile comparison on the GPU box - tools/parity_full_tile.py,
tests/test_gpu_model.py::test_whole_tile_64_every_voxel_vs_oracle_f32_f64_and_reference_whole_head - is bounded by).  The
manifest (tests/golden/manifest.json["S64"]) keeps, per case and head, on the WHOLE tile and on the lattice: scaled max, rms of
the scaled error and the fraction of voxels beyond 1e-4 true relative error, for reference-f32 vs truth and for 1 vs 8 threads.

Round 6 also (a) pins the oracle's float64 mode (oracle/model_oracle.py::mica_forward(dtype=torch.float64)) on EVERY logit of every
case against the reference module's float64 run (max |difference| in manifest.json["oracle64_vs_reference64_maxabs"], asserted
0.0), and its float32 mode likewise (["oracle32_vs_reference32_maxabs_S64"]); (b) writes for WHOLE_HEAD_CASES one WHOLE head of the
reference (tests/golden/whole_bb_S64_<case>.npz: `bb` = the float32 backbone logits [4, 64, 64, 64] of the 8-thread run, `bb64_f32` =
the float64 logits rounded to float32 (6e-8 relative: three orders below the distances measured)).

TEST INFRASTRUCTURE.  Only arrays are committed; inputs are regenerated from seeds on both sides (mica_amd/synth.py::case64).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

from mica_amd.synth import CASES64, case64                                                    # noqa: E402
from oracle._check import LATTICE_OFFSET, LATTICE_STRIDE, lattice                             # noqa: E402
from oracle import model_oracle as mo                                                         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
WHOLE_HEAD_CASES = ("w2022g6", "w99g10_s104")          # the default weights; the tile furthest from the reference's float32 logits


def metrics(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    sc = np.maximum(np.abs(b), np.sqrt(np.mean(b ** 2)))
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = d / np.abs(b)
    return {"scaled": float(np.max(d / sc)), "rms": float(np.sqrt(np.mean((d / sc) ** 2))),
            "frac_rel_gt_1e-4": float(np.mean(rel > 1e-4)), "maxabs": float(d.max())}


def both(a, b):
    """metrics on the whole tile and on the lattice the fixtures keep"""
    return {"full": metrics(a, b), "sub": metrics(lattice(a), lattice(b))}


def ref_model(w, double=False):
    from models.model import MICA
    m = MICA()
    if double:
        m = m.double()
        m.load_state_dict({k: torch.from_numpy(v.copy()).double() for k, v in w.items()}, strict=True)
    else:
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    return m.eval()


def run(m, x, af, threads):
    torch.set_num_threads(threads)
    with torch.no_grad():
        return [t.numpy().copy() for t in m(x, af)]


def gen(case, manifest):
    t0 = time.time()
    w, x, af = case64(case)
    x, af = torch.from_numpy(x), torch.from_numpy(af)
    m = ref_model(w)
    r8 = run(m, x, af, 8)
    r1 = run(m, x, af, 1)
    del m
    t64 = run(ref_model(w, double=True), x.double(), af.double(), 8)
    # the oracle's two modes against the reference module on every logit of the tile
    torch.set_num_threads(8)
    o32 = [t.numpy() for t in mo.mica_forward(w, x, af)]
    o64 = [t.numpy() for t in mo.mica_forward(w, x, af, dtype=torch.float64)]
    d32 = max(float(np.abs(a - b).max()) for a, b in zip(o32, r8))
    d64 = max(float(np.abs(a - b).max()) for a, b in zip(o64, t64))
    manifest.setdefault("oracle32_vs_reference32_maxabs_S64", {})[case] = d32
    manifest.setdefault("oracle64_vs_reference64_maxabs", {})[case] = d64
    assert d64 == 0.0 and d32 == 0.0, (case, d32, d64)
    del o32, o64
    rec = {"reference_f32_vs_truth": {}, "threads_1_vs_8": {}, "reference_f32_1thread_vs_truth": {}}
    for n, a8, a1, t in zip(("bb", "ca", "aa"), r8, r1, t64):
        rec["reference_f32_vs_truth"][n] = both(a8, t)
        rec["reference_f32_1thread_vs_truth"][n] = both(a1, t)
        rec["threads_1_vs_8"][n] = both(a1, a8)
    manifest.setdefault("S64", {})[case] = rec
    manifest["S64_lattice"] = {"stride": LATTICE_STRIDE, "offset": list(LATTICE_OFFSET)}
    H = ("bb", "ca", "aa")
    col = lambda which, part, key: np.array([rec[which][n][part][key] for n in H])
    arrays = {"bb64": lattice(t64[0]), "ca64": lattice(t64[1]), "aa64": lattice(t64[2]),
              "bb": lattice(r8[0]), "ca": lattice(r8[1]), "aa": lattice(r8[2]),
              "S": 64, "stride": LATTICE_STRIDE, "offset": np.array(LATTICE_OFFSET)}
    for part, suffix in (("sub", ""), ("full", "_full")):
        arrays["ref32_scaled" + suffix] = col("reference_f32_vs_truth", part, "scaled")
        arrays["ref32_rms" + suffix] = col("reference_f32_vs_truth", part, "rms")
        arrays["ref32_frac_rel" + suffix] = col("reference_f32_vs_truth", part, "frac_rel_gt_1e-4")
        arrays["floor_scaled" + suffix] = col("threads_1_vs_8", part, "scaled")
        arrays["floor_rms" + suffix] = col("threads_1_vs_8", part, "rms")
        arrays["floor_frac_rel" + suffix] = col("threads_1_vs_8", part, "frac_rel_gt_1e-4")
    np.savez_compressed(os.path.join(OUT, f"truth64_S64_sub_{case}.npz"), **arrays)
    if case in WHOLE_HEAD_CASES:
        np.savez_compressed(os.path.join(OUT, f"whole_bb_S64_{case}.npz"), bb=np.ascontiguousarray(r8[0][0]),
                            bb64_f32=np.ascontiguousarray(t64[0][0]).astype(np.float32), S=64)
    print(case, "%.0f s" % (time.time() - t0), json.dumps({k: {n: v[n]["full"] for n in v} for k, v in rec.items()}), flush=True)


def main():
    from oracle._check import CheckRun
    with CheckRun(globals(), sys.argv[1:], exact=False, seed=("manifest.json",)) as chk:
        mp = os.path.join(OUT, "manifest.json")
        manifest = json.load(open(mp))
        for case in (chk.argv or list(CASES64)):
            gen(case, manifest)
            json.dump(manifest, open(mp, "w"), indent=1)       # per case: a long run that is cut short keeps what it has


if __name__ == "__main__":
    main()
