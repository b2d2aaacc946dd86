"""Round-4 goldens: the reference MICA module where a TRAINED network would stress the numerics (build container only).

Run:  python oracle/gen_golden_r4.py [--check] [heavy] [blob]      (needs /root/reference; about 3 minutes on 8 cores)

No trained checkpoint or real map is reachable offline (reference README.md:27-39), and the goldens of rounds 1-2 all use
U(-b, b) weights and dense uniform-random density.  Two more input families, both through the reference's own
`models.model.MICA` (imported unmodified, weights through load_state_dict):

  heavy   heavy-tailed weights (mica_amd/weights.py::synth_state_dict_heavy: magnitudes spread log-uniformly over 128x, ~3 % of
          the output channels of every layer 30x larger), uniform density + Bernoulli AF3 encodings
  blob    the default weights on a map shaped like the normaliser's output (utils/preprocessing.py:122-133): > 80 % exact
          zeros, compact blobs reaching 1.0, and AF3 encodings clustered as residues around the blob centres

each at S = 16 (whole logits, float32 and - `MICA().double()` - float64 truth with the reference's own float32 distance from
it) and on one 64^3 tile (strided subsample + probabilities + argmax, as model_S64_*_sub_*.npz).  Every case asserts that
oracle/model_oracle.py reproduces the reference module exactly (max |diff| recorded in the manifest).

TEST INFRASTRUCTURE.  Only arrays are committed; inputs are regenerated from seeds on both sides.
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
REF = "/root/reference"
sys.path.insert(0, REF)

from mica_amd.synth import stress_case                                                        # noqa: E402
from oracle import model_oracle as mo                                                         # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def case_inputs(kind, S):
    """(weights dict, x [1,1,S,S,S], af [1,24,S,S,S]) - mica_amd/synth.py::stress_case, which the tests call too."""
    w, x, af = stress_case(kind, S)
    return w, torch.from_numpy(x), torch.from_numpy(af)


def scaled(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), np.sqrt(np.mean(b ** 2)))))


def ref_model(w, double=False):
    from models.model import MICA
    m = MICA()
    if double:
        m = m.double()
        m.load_state_dict({k: torch.from_numpy(v.copy()).double() for k, v in w.items()}, strict=True)
    else:
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    return m.eval()


def gen(kind, manifest):
    # ---- S = 16: whole logits, float32 and float64 ----
    S = 16
    w, x, af = case_inputs(kind, S)
    m = ref_model(w)
    with torch.no_grad():
        r32 = [t.numpy().copy() for t in m(x, af)]
    o = mo.mica_forward(w, x, af)
    d = max(float(np.abs(a - b.numpy()).max()) for a, b in zip(r32, o))
    manifest["oracle_vs_reference_maxabs"][f"r4_{kind}_S16"] = d
    assert d < 1e-5, d
    with torch.no_grad():
        t64 = [t.numpy().copy() for t in ref_model(w, double=True)(x.double(), af.double())]
    r32s = np.array([scaled(a, b) for a, b in zip(r32, t64)])
    np.savez_compressed(os.path.join(OUT, f"r4_{kind}_S16.npz"), bb=r32[0], ca=r32[1], aa=r32[2],
                        bb64=t64[0], ca64=t64[1], aa64=t64[2], ref32_scaled=r32s, S=S)
    print(kind, "S16 oracle-vs-ref", d, "reference float32 vs float64", r32s.tolist(), flush=True)
    # ---- one 64^3 tile ----
    S = 64
    w, x, af = case_inputs(kind, S)
    with torch.no_grad():
        rb, rc, ra = m(x, af)
    pb, pc, pa, pp = mo.postprocess(rb, rc, ra)
    from oracle._check import LATTICE_OFFSET, LATTICE_STRIDE, lattice
    np.savez_compressed(os.path.join(OUT, f"r4_{kind}_S64_sub.npz"),
                        bb=lattice(rb.numpy()), ca=lattice(rc.numpy()), aa=lattice(ra.numpy()),
                        bb_prob=lattice(pb.numpy()), ca_prob=lattice(pc.numpy()),
                        aa_prob=lattice(pa.numpy()), aa_pred=lattice(pp.numpy()).astype(np.uint8),
                        S=64, stride=LATTICE_STRIDE, offset=np.array(LATTICE_OFFSET))
    print(kind, "S64 done; logits rms", [float(t.pow(2).mean().sqrt()) for t in (rb, rc, ra)], flush=True)


def main():
    from oracle._check import CheckRun
    with CheckRun(globals(), sys.argv[1:], exact=False, seed=("manifest.json",)) as chk:
        torch.set_num_threads(8)
        mp = os.path.join(OUT, "manifest.json")
        manifest = json.load(open(mp))
        for kind in (chk.argv or ["heavy", "blob"]):
            gen(kind, manifest)
        json.dump(manifest, open(mp, "w"), indent=1)


if __name__ == "__main__":
    main()
