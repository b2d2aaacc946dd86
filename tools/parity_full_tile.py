"""Whole-tile, lattice-free parity at the production tile size: EVERY logit of a 64^3 tile, GPU against the CPU oracle.

For each of the sixteen 64^3 cases (mica_amd/synth.py::CASES64) the oracle (oracle/model_oracle.py: torch CPU, bit-equal to the reference
module in float32 AND in float64 on every logit of these very tiles - tests/golden/manifest.json["oracle64_vs_reference64_maxabs"],
oracle/gen_golden_r5.py) is run on THIS host in float32 and in float64, and the GPU path with conv variant 0 (every 3^3 conv on the
F(2,3) kernel) and 3 (the shipped graph) is compared with both on all 29 x 262 144 logits:

  scaled max / rms of |gpu - ref| / max(|ref|, rms(ref))  (the tests' metric), the fraction of voxels beyond 1e-4 literal relative error,
  where the maximum sits (z, y, x) and its position inside the kernels' output tiles (z mod 4, y mod 2, x mod 4), and the rms of the
  scaled error PER POSITION of those tiles (x mod 4: the four outputs of an F(4,3) quad / two F(2,3) pairs; y mod 2: the rows of the
  2-row tile; z mod 4: the planes of the 4-plane tile) - a lattice position that were systematically worse would show there.

Beside every line: the reference float32 path's own whole-tile distance from the float64 truth and between its 1-thread and 8-thread
runs, as recorded from the reference itself in the build container (tests/golden/truth64_S64_sub_<case>.npz, `*_full`).

usage: python tools/parity_full_tile.py [--variants 0,3] [--cases a,b,...] [--threads N]  > profiles/rNN_parity_full_tile.txt
(GPU box; ~40 s of CPU per case: progress lines go to stderr)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mica_amd.engine import AF_PER_TILE, Engine
from mica_amd.synth import CASES64, case64
from oracle import model_oracle as mo        # the checker

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
HEADS = ("bb", "ca", "aa")


def scaled_error(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref) / np.maximum(np.abs(ref), np.sqrt(np.mean(ref ** 2)))


def metrics(got, ref):
    e = scaled_error(got, ref)
    with np.errstate(divide="ignore", invalid="ignore"):
        fr = float(np.mean(np.abs(np.asarray(got, np.float64) - ref) / np.abs(np.asarray(ref, np.float64)) > 1e-4))
    am = np.unravel_index(int(np.argmax(e)), e.shape)
    z, y, x = (int(v) for v in am[-3:])
    pos = {"x%4": [float(np.sqrt(np.mean(e[..., :, :, r::4] ** 2))) for r in range(4)],
           "y%2": [float(np.sqrt(np.mean(e[..., :, r::2, :] ** 2))) for r in range(2)],
           "z%4": [float(np.sqrt(np.mean(e[..., r::4, :, :] ** 2))) for r in range(4)]}
    return {"max": float(e.max()), "rms": float(np.sqrt(np.mean(e ** 2))), "frac": fr, "argmax": (int(am[1]), z, y, x), "res": (z % 4, y % 2, x % 4), "pos": pos}


def gpu_logits(w, x, af, variant):
    e = Engine(0, max_batch=1, tile_size=64, conv_variant=variant)
    e.load_state_dict(w)
    out = [o.cpu().numpy() for o in e.forward_logits(torch.from_numpy(x).cuda(), torch.from_numpy(af).cuda(), AF_PER_TILE)]
    e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,3")
    ap.add_argument("--cases", default=",".join(CASES64))
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",")]
    if a.threads:
        torch.set_num_threads(a.threads)
    print(f"# whole 64^3 tiles, every logit (29 x 262144 per tile): GPU conv variants {variants} vs oracle float32 / float64 on this host "
          f"({torch.get_num_threads()} threads); scaled = |gpu - ref| / max(|ref|, rms(ref))")
    print("# case | variant | against | head: scaled max, rms, fraction beyond 1e-4 literal relative, argmax (channel, z, y, x) -> (z%4, y%2, x%4) | reference's own whole-tile figure")
    worst = {}
    pos_acc = {}
    for case in a.cases.split(","):
        t0 = time.time()
        w, x, af = case64(case)
        o32 = [t.numpy() for t in mo.mica_forward(w, x, af)]
        t1 = time.time()
        o64 = [t.numpy() for t in mo.mica_forward(w, x, af, dtype=torch.float64)]
        t2 = time.time()
        fx = np.load(os.path.join(G, f"truth64_S64_sub_{case}.npz"))
        have_full = "ref32_scaled_full" in fx.files
        # the oracle's float32 run on this host against its own float64 run: the reference arithmetic's distance from the truth here
        own = [metrics(a32, a64) for a32, a64 in zip(o32, o64)]
        print(f"{case:18s} oracle-f32 vs oracle-f64 (this host): " + "  ".join(f"{h} max {m['max']:.2e} rms {m['rms']:.2e} rel>1e-4 {m['frac']:.3f}" for h, m in zip(HEADS, own))
              + ("   | build container, reference module: max " + " / ".join(f"{v:.2e}" for v in fx["ref32_scaled_full"]) + " rms " + " / ".join(f"{v:.2e}" for v in fx["ref32_rms_full"])
                 + "; 1 vs 8 threads max " + " / ".join(f"{v:.2e}" for v in fx["floor_scaled_full"]) if have_full else ""), flush=True)
        for v in variants:
            g = gpu_logits(w, x, af, v)
            for what, ref in (("oracle f32", o32), ("oracle f64", o64)):
                ms = [metrics(gg, rr) for gg, rr in zip(g, ref)]
                spread = max(abs(p / m["rms"] - 1.0) for m in ms for ax in m["pos"].values() for p in ax)
                print(f"{case:18s} v{v} vs {what}: " + "  ".join(
                    f"{h} max {m['max']:.2e} rms {m['rms']:.2e} rel>1e-4 {m['frac']:.3f} at {m['argmax']}->{m['res']}" for h, m in zip(HEADS, ms))
                      + f"  | rms per tile position (x%4, y%2, z%4) within {100 * spread:.1f} % of the tile's", flush=True)
                for h, m in zip(HEADS, ms):
                    k = (v, what)
                    worst[k] = max(worst.get(k, (0, None, None)), (m["max"], case, h))
                    acc = pos_acc.setdefault((v, what, h), {"x%4": [], "y%2": [], "z%4": []})
                    for ax in acc:
                        acc[ax].append(m["pos"][ax])
        print(f"# {case}: oracle f32 {t1 - t0:.0f} s, f64 {t2 - t1:.0f} s, total {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    print("\n# worst case per (variant, against): scaled max, case, head")
    for k in sorted(worst):
        print(f"v{k[0]} vs {k[1]}: {worst[k][0]:.3e}  {worst[k][1]}  {worst[k][2]}")
    print("\n# rms of the scaled error per position inside the kernels' output tiles, averaged (rms) over the cases: a systematically worse lattice position would stand out")
    for k in sorted(pos_acc):
        line = []
        for ax, rows in pos_acc[k].items():
            r = np.sqrt(np.mean(np.square(np.array(rows)), axis=0))
            line.append(f"{ax} " + " ".join(f"{v:.3e}" for v in r) + f" (spread {100 * (r.max() / r.min() - 1):.1f} %)")
        print(f"v{k[0]} vs {k[1]} {k[2]}: " + "   ".join(line))


if __name__ == "__main__":
    main()
