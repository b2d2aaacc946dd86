"""Whole-network CPU emulation of the GPU path's conv arithmetic under different Winograd variants.

TEST INFRASTRUCTURE (numerics experiment, never imported by the product path).  The op list of oracle/model_oracle.py is run
with every dense 3x3x3 conv replaced by an emulation of what `conv_wino16_kernel` does to it:

    f32 input transform along x  ->  x * ascale split into f16 hi + f16 lo  ->  transformed weights (f64-exact constants,
    rounded to f32) scaled by a power of two so that max |u| lands in (2048, 4096], split the same way  ->  the three products
    hi*hi + hi*lo + lo*hi accumulated in f32 over K = 9 (dz, dy) taps x Cin  ->  f32 output transform, unscale, bias.

and every 1x1x1 conv with Cout >= 64 by the split-f16 direct product of `conv1x1_kernel`.  InstanceNorm statistics are taken in
float64 (the GPU merges its partials in f64).  Everything else (stem, depthwise, gates, heads' final 1x1) is torch fp32, as on
the GPU (f32 VALU).

Variants of the 3^3 convs:
    f32        torch's own conv (the emulator with nothing emulated: shows what the f64 statistics alone change)
    direct     split-f16 products, no Winograd
    f23        F(2,3) along x everywhere                     = the shipped kernel; must land where the GPU tests land
    f43        F(4,3) along x everywhere, points {0, +-1, +-2, inf}
    f43h       F(4,3) along x, points {0, +-1, +-1/2, inf}
    f43@128    F(4,3) on the layers with Cout >= 128 (the bn = 128 kernel variant), F(2,3) elsewhere
    f43h@128   the same with the +-1/2 points
    f43@big    F(4,3) on encoder.2's conv3 (512->256) and transition (256->512) only - 51 % of the network's FLOPs
    f43h@big   the same with the +-1/2 points
    f43@e2     F(4,3) on all four 3^3 convs of encoder.2 (68 % of the FLOPs)
    f43s, f43s@e2   the same with the points {0, +-3/2, +-2/3, inf} of the shipped kernel (kernels_conv43.hip)
    f43s@e2t1, f43s@e2t   encoder.2 plus the transition conv of encoder.1 / of encoder.0 and encoder.1 (the other Cout % 128 == 0 layers
                    whose operand comes straight out of a 1x1 conv's epilogue)
    f43s@late, @latefpn, @lateheads, @latec1, @latec2   (round 5) encoder.2 plus the LATE narrow layers - the FPN's three smooth convs
                    (64 -> 64) and / or the heads' conv1 (192 / 196 / 200 -> 64) and conv2 (64 -> 32): the layers the bn = 64 / 32
                    variants of the F(2,3) kernel spend 0.47 / 0.29 of the MFMA peak on

Criterion (tests/test_gpu_model.py): scaled error max |got - ref| / max(|ref|, rms(ref)) < 1e-4 against the reference module's
float32 logits (tests/golden/model_S16_*.npz) AND against its float64 logits (truth64_S16_*.npz), the latter also <= 1.5x the
reference's own float32 distance from that truth.

Run:  python oracle/wino_network.py [variants...]        (CPU, about a minute per variant and weight set; no reference needed)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import model_oracle as mo                    # noqa: E402

ASCALE = 16.0


# ----------------------------------------------------------------------------------------------
# transforms  (B^T: alpha x alpha input, G: alpha x 3 weights, A^T: m x alpha output)
# ----------------------------------------------------------------------------------------------
def cook_toom(points, m, r=3):
    """Winograd F(m, r) matrices for the given finite interpolation points plus the point at infinity (Toom-Cook /
    Lagrange construction, as in Lavin & Gray 2015); returned in float64 with A^T G-scaling chosen so that the matrices
    for {0, 1, -1} are the textbook F(2,3) ones up to sign."""
    from fractions import Fraction as Fr
    pts = [Fr(p) for p in points]
    a = m + r - 1
    assert len(pts) == a - 1
    # A^T [m x a]: rows i = powers p^i; last column = infinity (only the highest power)
    AT = [[pts[j] ** i for j in range(a - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    # G [a x r]: rows j = p_j^k / N_j with N_j = prod_{l != j}(p_j - p_l); last row = infinity
    G = []
    for j in range(a - 1):
        n = Fr(1)
        for l in range(a - 1):
            if l != j:
                n *= (pts[j] - pts[l])
        G.append([pts[j] ** k / n for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    # B^T [a x a]: rows j = coefficients of M(x)/(x - p_j) (Lagrange numerators), last row = M(x) itself
    def polymul(p, q):
        out = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for k, y in enumerate(q):
                out[i + k] += x * y
        return out
    BT = []
    for j in range(a - 1):
        poly = [Fr(1)]
        for l in range(a - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fr(1)])
        BT.append(poly + [Fr(0)])
    poly = [Fr(1)]
    for l in range(a - 1):
        poly = polymul(poly, [-pts[l], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return f(BT), f(G), f(AT)


def _check_transform(T, m):
    BT, G, AT = T
    rng = np.random.default_rng(1)
    d = rng.standard_normal(m + 2)
    g = rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([d[i] * g[0] + d[i + 1] * g[1] + d[i + 2] * g[2] for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)


F23 = cook_toom([0, 1, -1], 2)
F43 = cook_toom([0, 1, -1, 2, -2], 4)
F43H = cook_toom([0, 1, -1, 0.5, -0.5], 4)
from fractions import Fraction as _Fr                      # noqa: E402
F43S = cook_toom([0, _Fr(3, 2), _Fr(-3, 2), _Fr(2, 3), _Fr(-2, 3)], 4)      # the shipped kernel's points (a b = 1: balanced products)
for _T, _m in ((F23, 2), (F43, 4), (F43H, 4), (F43S, 4)):
    _check_transform(_T, _m)


# ----------------------------------------------------------------------------------------------
# split-f16 arithmetic
# ----------------------------------------------------------------------------------------------
def split(x, scale):
    """x*scale = hi + lo with hi, lo in f16 (values returned as f32 tensors holding exactly those halves)."""
    xs = (x * scale).float()
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi.float(), lo.float()


def wscale(u):
    m = float(u.abs().max())
    return 2.0 ** np.floor(np.log2(4096.0 / m))


def split_mm(a, b, ws):
    """a [M, K] f32, b [K, N] f32 -> sum_k a b with the three split products, f32 accumulate, still scaled by ASCALE*ws."""
    ah, al = split(a, ASCALE)
    bh, bl = split(b, ws)
    return ah @ bh + (ah @ bl + al @ bh)


def conv3_emulated(x, weight, bias, T):
    """x f32 [1, Cin, D, H, W]; weight [Cout, Cin, 3, 3, 3] (kd, kh, kw); T = (BT, G, AT) or None for the direct form."""
    assert x.shape[0] == 1
    _, cin, D, H, W = x.shape
    cout = weight.shape[0]
    xp = F.pad(x[0], (1, 1, 1, 1, 1, 1))                                   # [Cin, D+2, H+2, W+2]
    w64 = weight.double()
    if T is None:
        cols = torch.stack([xp[:, dz:dz + D, dy:dy + H, dx:dx + W] for dz in range(3) for dy in range(3) for dx in range(3)], 0)
        a = cols.permute(2, 3, 4, 0, 1).reshape(D * H * W, 27 * cin)
        b = w64.permute(2, 3, 4, 1, 0).reshape(27 * cin, cout).float()
        ws = wscale(b)
        y = split_mm(a, b, ws) / (ASCALE * ws)
        y = y.reshape(D, H, W, cout).permute(3, 0, 1, 2)
    else:
        BT, G, AT = T
        m, al = AT.shape
        assert W % m == 0, (W, m)
        nt = W // m
        # input transform in f32 (the producer pass): tiles of alpha inputs at stride m along x
        idx = (torch.arange(nt)[:, None] * m + torch.arange(al)[None, :]).reshape(-1)
        d = xp[..., idx].reshape(cin, D + 2, H + 2, nt, al)
        td = torch.einsum("pa,czyna->pczyn", torch.from_numpy(BT).float(), d)           # [al, Cin, D+2, H+2, nt] f32
        u = torch.einsum("pk,oczyk->pzyco", torch.from_numpy(G), w64).float()   # [al, 3, 3, Cin, Cout]
        ws = wscale(u)
        mm = []
        for p in range(al):
            a = torch.stack([td[p, :, dz:dz + D, dy:dy + H] for dz in range(3) for dy in range(3)], 0)   # [9, Cin, D, H, nt]
            a = a.permute(2, 3, 4, 0, 1).reshape(D * H * nt, 9 * cin)
            b = u[p].reshape(9 * cin, cout)
            mm.append(split_mm(a, b, ws))
        mm = torch.stack(mm, 0)                                                         # [al, D*H*nt, Cout]
        y = torch.einsum("mp,pvo->vmo", torch.from_numpy(AT).float(), mm) / (ASCALE * ws)
        y = y.reshape(D, H, nt * m, cout).permute(3, 0, 1, 2)
    return (y + bias.view(-1, 1, 1, 1))[None].contiguous()


def conv1_emulated(x, weight, bias):
    _, cin, D, H, W = x.shape
    cout = weight.shape[0]
    a = x[0].reshape(cin, -1).t()
    b = weight.reshape(cout, cin).t().contiguous()
    ws = wscale(b)
    y = split_mm(a, b, ws) / (ASCALE * ws)
    return (y.t().reshape(cout, D, H, W) + bias.view(-1, 1, 1, 1))[None].contiguous()


# ----------------------------------------------------------------------------------------------
# the network with hooks
# ----------------------------------------------------------------------------------------------
class Emulated:
    """Context manager: patches oracle.model_oracle's `_conv` / `_in_relu` for the duration of one forward."""

    def __init__(self, variant):
        self.variant = variant
        self.log = []

    def pick(self, cin, cout, name=""):
        v = self.variant
        if "@late" in v:                                  # encoder.2 plus subsets of the late narrow layers behind it (round 5)
            sets = {"@late": ("fpn.smooth.", "backbone_head.conv", "ca_head.conv", "aa_head.conv"), "@latefpn": ("fpn.smooth.",),
                    "@lateheads": ("backbone_head.conv", "ca_head.conv", "aa_head.conv"),
                    "@latec1": ("backbone_head.conv1", "ca_head.conv1", "aa_head.conv1"),
                    "@latec2": ("backbone_head.conv2", "ca_head.conv2", "aa_head.conv2")}
            return F43S if name.startswith(("encoder.2.",) + sets["@" + v.split("@")[1]]) else F23
        if "@e2t" in v:                                   # encoder.2 and the transition of encoder.1 (@e2t1) or of encoder.0 and encoder.1 (@e2t)
            extra = ("encoder.1.transition",) if "@e2t1" in v else ("encoder.0.transition", "encoder.1.transition")
            return {"f43": F43, "f43h": F43H, "f43s": F43S}[v.split("@")[0]] if name.startswith(("encoder.2.",) + extra) else F23
        if "@e2" in v:                                    # all four 3^3 convs of encoder.2 (conv1, conv2, conv3, transition)
            return {"f43": F43, "f43h": F43H, "f43s": F43S}[v.split("@")[0]] if name.startswith("encoder.2.") else F23
        if "@e2c" in v:
            pass
        if v == "direct":
            return None
        if "@128" in v:
            return {"f43": F43, "f43h": F43H}[v.split("@")[0]] if cout >= 128 else F23
        if "@big" in v:                                   # encoder.2's conv3 (512->256) and transition (256->512): 51 % of the FLOPs
            return {"f43": F43, "f43h": F43H}[v.split("@")[0]] if cin * cout >= 512 * 256 else F23
        return {"f23": F23, "f43": F43, "f43h": F43H, "f43s": F43S}[v]

    def __enter__(self):
        self._conv, self._in = mo._conv, mo._in_relu
        orig = self._conv

        def conv(w, name, x, pad=0, groups=1):
            wt, bs = mo._t(w, name + ".weight"), mo._t(w, name + ".bias")
            if self.variant != "f32" and groups == 1 and x.shape[0] == 1 and x.shape[2] > 1:
                if tuple(wt.shape[2:]) == (3, 3, 3) and wt.shape[1] >= 16:
                    return conv3_emulated(x, wt, bs, self.pick(wt.shape[1], wt.shape[0], name))
                if tuple(wt.shape[2:]) == (1, 1, 1) and wt.shape[0] >= 64:
                    return conv1_emulated(x, wt, bs)
            return orig(w, name, x, pad, groups)

        def in_relu(x):
            xd = x.double()
            mean = xd.mean(dim=(2, 3, 4), keepdim=True)
            var = xd.var(dim=(2, 3, 4), unbiased=False, keepdim=True)
            rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
            return F.relu((x - mean.float()) * rstd)

        mo._conv, mo._in_relu = conv, in_relu
        return self

    def __exit__(self, *exc):
        mo._conv, mo._in_relu = self._conv, self._in
        return False


def scaled_err(got, ref, rms=False):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    s = np.maximum(np.abs(ref), np.sqrt(np.mean(ref ** 2)))
    e = np.abs(got - ref) / s
    return float(np.sqrt(np.mean(e * e))) if rms else float(np.max(e))


CASES = [("w2022g6", "model_S16_af.npz", 2022, 6.0), ("w7g3", "model_S16_af_w7g3.npz", 7, 3.0),
         ("w99g10", "model_S16_af_w99g10.npz", 99, 10.0)]


def run_case(variant, tag, golden_name, wseed, wgain, golden_dir=os.path.join(ROOT, "tests", "golden")):
    from mica_amd.synth import synth_af, synth_density
    from mica_amd.weights import synth_state_dict
    g = np.load(os.path.join(golden_dir, golden_name))
    t = np.load(os.path.join(golden_dir, f"truth64_S16_{tag}.npz"))
    S, seed, afp = 16, int(g["seed"]), float(g["afp"])
    w = synth_state_dict(wseed, wgain)
    x = torch.from_numpy(synth_density((1, 1, S, S, S), seed))
    af = torch.from_numpy(synth_af((S, S, S), seed, afp))[None]
    with Emulated(variant):
        out = mo.mica_forward(w, x, af)
    e32 = [scaled_err(o.numpy(), g[k]) for o, k in zip(out, ("bb", "ca", "aa"))]
    e64 = [scaled_err(o.numpy(), t[k]) for o, k in zip(out, ("bb", "ca", "aa"))]
    rms = max(scaled_err(o.numpy(), t[k], rms=True) for o, k in zip(out, ("bb", "ca", "aa")))
    ref_rms = max(scaled_err(g[k], t[k], rms=True) for k in ("bb", "ca", "aa"))
    return e32, e64, [float(v) for v in t["ref32_scaled"]], rms, ref_rms


def main():
    torch.set_num_threads(8)
    variants = sys.argv[1:] or ["f32", "direct", "f23", "f43", "f43h", "f43@128", "f43h@128", "f43@big", "f43h@big"]
    print("scaled error bb / ca / aa  (bar: < 1e-4 vs the reference's float32 logits; vs float64 truth also <= 1.5x the reference's own)")
    for tag, gname, ws, wg in CASES:
        for v in variants:
            e32, e64, r, rms, ref_rms = run_case(v, tag, gname, ws, wg)
            ok = max(e32) < 1e-4 and max(e64) < 1e-4 and all(a <= 1.5 * b for a, b in zip(e64, r))
            print(f"{tag:8s} {v:9s} vs ref32 " + " / ".join(f"{e:.1e}" for e in e32) + "   vs truth64 " +
                  " / ".join(f"{e:.1e}" for e in e64) + "   (ref32 vs truth64 " + " / ".join(f"{e:.1e}" for e in r) + f")  rms vs truth64 {rms:.1e} (ref32 {ref_rms:.1e})  " +
                  ("PASS" if ok else "FAIL"), flush=True)


if __name__ == "__main__":
    main()
