"""Every whole-network fixture against the GPU path, both conv variants: the margins the parity tests leave (profiles/rNN_parity_margins.txt).

For each fixture: scaled max error, rms of the scaled error and the fraction of voxels beyond 1e-4 literal relative error, against the
reference module's float32 logits and - where the fixture has them - against its float64 logits, for conv variant 0 (every 3x3x3 conv on
the F(2,3) kernel), variant 1 (encoder.2 on the F(4,3) kernel) and variant 3 (that and the late narrow layers on its 64-channel variant).
usage: python tools/parity_margins.py [variants, e.g. 0,1,3] > out.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd.engine import AF_BATCH, AF_PER_TILE, Engine
from mica_amd.synth import CASES64, case64, stress_case, synth_af, synth_density
from mica_amd.weights import synth_state_dict

VARIANTS = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (0, 1, 3)
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def metrics(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    sc = np.maximum(np.abs(ref), np.sqrt(np.mean(ref ** 2)))
    with np.errstate(divide="ignore", invalid="ignore"):
        fr = float(np.mean(d / np.abs(ref) > 1e-4))
    return float(np.max(d / sc)), float(np.sqrt(np.mean((d / sc) ** 2))), fr


def run(w, x, af, S, mode, variant):
    e = Engine(0, max_batch=1, tile_size=S, conv_variant=variant)
    e.load_state_dict(w)
    out = [o.cpu().numpy() for o in e.forward_logits(torch.from_numpy(x).cuda(), None if af is None else torch.from_numpy(af).cuda(), mode)]
    e.close()
    return out


def line(name, variant, what, got, refs, note=""):
    m = [metrics(g, r) for g, r in zip(got, refs)]
    print(f"{name:34s} v{variant} {what:14s} max " + " / ".join(f"{a[0]:.2e}" for a in m) + "   rms " + " / ".join(f"{a[1]:.2e}" for a in m) +
          "   rel>1e-4 " + " / ".join(f"{a[2]:.3f}" for a in m) + ("   " + note if note else ""), flush=True)


print("fixture                            variant  against        scaled max bb / ca / aa              rms                                  literal-relative fraction")
for tag, gname, ws, wg in (("w2022g6", "model_S16_af.npz", 2022, 6.0), ("w7g3", "model_S16_af_w7g3.npz", 7, 3.0), ("w99g10", "model_S16_af_w99g10.npz", 99, 10.0)):
    g, t = np.load(os.path.join(G, gname)), np.load(os.path.join(G, f"truth64_S16_{tag}.npz"))
    x, af = synth_density((1, 1, 16, 16, 16), int(g["seed"])), synth_af((16, 16, 16), int(g["seed"]), float(g["afp"]))[None]
    for v in VARIANTS:
        out = run(synth_state_dict(ws, wg), x, af, 16, AF_BATCH, v)
        line(f"S16 {tag}", v, "reference f32", out, [g[k] for k in ("bb", "ca", "aa")])
        line(f"S16 {tag}", v, "float64 truth", out, [t[k] for k in ("bb", "ca", "aa")], "reference f32 vs truth max " + " / ".join(f"{a:.2e}" for a in t["ref32_scaled"]))
for kind in ("heavy", "blob"):
    g = np.load(os.path.join(G, f"r4_{kind}_S16.npz"))
    w, x, af = stress_case(kind, 16)
    for v in VARIANTS:
        out = run(w, x, af, 16, AF_BATCH, v)
        line(f"S16 r4_{kind}", v, "reference f32", out, [g[k] for k in ("bb", "ca", "aa")])
        line(f"S16 r4_{kind}", v, "float64 truth", out, [g[k] for k in ("bb64", "ca64", "aa64")], "reference f32 vs truth max " + " / ".join(f"{a:.2e}" for a in g["ref32_scaled"]))
for case in CASES64:
    g = np.load(os.path.join(G, f"truth64_S64_sub_{case}.npz"))
    st = int(g["stride"])
    off = [int(q) for q in g["offset"]] if "offset" in g.files else [0, 0, 0]
    w, x, af = case64(case)
    for v in VARIANTS:
        out = [o[..., off[0]::st, off[1]::st, off[2]::st] for o in run(w, x, af, 64, AF_PER_TILE, v)]
        line(f"S64 {case} (lattice {st}+{off})", v, "reference f32", out, [g[k] for k in ("bb", "ca", "aa")],
             "reference 1 vs 8 threads rel>1e-4 " + " / ".join(f"{a:.3f}" for a in g["floor_frac_rel"]))
        line(f"S64 {case} (lattice {st}+{off})", v, "float64 truth", out, [g[k] for k in ("bb64", "ca64", "aa64")],
             "reference f32 vs truth max " + " / ".join(f"{a:.2e}" for a in g["ref32_scaled"]) + " rms " + " / ".join(f"{a:.2e}" for a in g["ref32_rms"]))
