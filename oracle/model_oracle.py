"""CPU restatement of the reference network forward (reference models/model.py).

TEST INFRASTRUCTURE - never imported by the product path.  Uses torch's CPU fp32 ops
(`F.conv3d`, `F.instance_norm`) because ATen CPU *is* the arithmetic of the reference's
CPU path (layer L1); the module structure is flattened into one
explicit op list over a plain ``{name: array}`` weight dict, so that no reference
Python is needed at run time.  Pinned by oracle/gen_golden.py against
``models.model.MICA`` with identical weights (max |diff| recorded in
tests/golden/manifest.json).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, name):
    v = w[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def _conv(w, name, x, pad=0, groups=1):
    return F.conv3d(x, _t(w, name + ".weight"), _t(w, name + ".bias"), padding=pad, groups=groups)


def _in_relu(x):
    # nn.InstanceNorm3d(C): affine=False, eps=1e-5, biased variance (model.py:81,108,...) + ReLU
    return F.relu(F.instance_norm(x, eps=1e-5))


def _gap(x):
    return x.mean(dim=(2, 3, 4), keepdim=True)              # nn.AdaptiveAvgPool3d(1)


def multi_scale_input(w, exp_map, af, p="input_processing"):
    """model.py:43-74 (eval mode: dropout is identity)."""
    feats = [_conv(w, f"{p}.exp_convs.{i}", exp_map, pad=k // 2) for i, k in enumerate((3, 5, 7, 9))]
    x_exp = torch.cat(feats, dim=1)                                           # :51
    g = torch.sigmoid(_conv(w, f"{p}.exp_attention.3",
                            F.relu(_conv(w, f"{p}.exp_attention.1", _gap(x_exp)))))
    x_exp = x_exp * g                                                         # :54
    if af is None or bool(af.abs().sum() < 1e-6):                             # :56-63 (batch-wide test)
        return _conv(w, f"{p}.exp_downsizing", x_exp)
    x_feat = _conv(w, f"{p}.feat_conv", af, pad=1)                            # :69
    imp = torch.sigmoid(_conv(w, f"{p}.feat_gate.2", F.relu(_conv(w, f"{p}.feat_gate.0", x_feat))))
    return _conv(w, f"{p}.fusion", torch.cat([x_exp, x_feat * imp], dim=1))   # :70-74


def se_block(w, p, x):
    """model.py:254-258."""
    b, c = x.shape[:2]
    y = _gap(x).view(b, c)
    y = F.relu(F.linear(y, _t(w, p + ".fc.0.weight"), _t(w, p + ".fc.0.bias")))
    y = torch.sigmoid(F.linear(y, _t(w, p + ".fc.3.weight"), _t(w, p + ".fc.3.bias")))
    return x * y.view(b, c, 1, 1, 1)


def residual_dense_block(w, p, x):
    """model.py:130-134 (note: despite the name there is no residual add)."""
    x1 = _in_relu(_conv(w, p + ".conv1.0", x, pad=1))
    x2 = _in_relu(_conv(w, p + ".conv2.0", torch.cat([x, x1], 1), pad=1))
    x3 = _in_relu(_conv(w, p + ".conv3.0", torch.cat([x, x1, x2], 1), pad=1))
    return se_block(w, p + ".se", x3)


def dual_attention(w, p, x):
    """model.py:98-101."""
    c = x.shape[1]
    local = _in_relu(_conv(w, p + ".local_attn.0", x, pad=1, groups=c))
    g = torch.sigmoid(_conv(w, p + ".global_attn.4", F.relu(_conv(w, p + ".global_attn.1", _gap(x)))))
    return _conv(w, p + ".fusion", torch.cat([local, g * x], 1))


def encoder(w, p, x):
    """model.py:149-152."""
    x = residual_dense_block(w, p + ".dense_block", x)
    x = dual_attention(w, p + ".dual_attn", x)
    return _in_relu(_conv(w, p + ".transition.0", x, pad=1))


def fpn(w, feats, p="fpn"):
    """model.py:182-205.  The two F.interpolate calls (:192-193) are exact identities
    because every level has the same spatial size, so they are elided."""
    sw = torch.softmax(_t(w, p + ".weights"), dim=0)
    outs = []
    for i, c in enumerate(feats):
        lat = _conv(w, f"{p}.lateral.{i}", c)
        outs.append(sw[i] * _conv(w, f"{p}.smooth.{i}.0", lat, pad=1))
    return torch.cat(outs, 1)


def head(w, p, x):
    """model.py:230-239."""
    x = _in_relu(_conv(w, p + ".conv1", x, pad=1))
    x = _in_relu(_conv(w, p + ".conv2", x, pad=1))
    g = torch.sigmoid(_conv(w, p + ".calibration.4", F.relu(_conv(w, p + ".calibration.1", _gap(x)))))
    return _conv(w, p + ".final", x * g)


@torch.no_grad()
def mica_forward(w, exp_map, af=None, return_intermediates=False, dtype=torch.float32):
    """model.py:331-348.  exp_map f32[B,1,D,H,W], af f32[B,24,D,H,W] or None ->
    (backbone f32[B,4,...], ca f32[B,4,...], aa f32[B,21,...]) logits, NCDHW.

    dtype=torch.float64 is the reference module after `MICA().double()` with the same float32 weights widened (what
    oracle/gen_golden_r5.py and oracle/noise_floor.py --truth run as "the exact answer"): every op below then computes in
    double.  Pinned bit for bit against the reference's own float64 run: S = 16 by tests/test_cpu_oracle.py against
    tests/golden/truth64_S16_*.npz, whole 64^3 tiles by gen_golden_r5.py (manifest.json["oracle64_vs_reference64_maxabs"])."""
    exp_map = torch.as_tensor(exp_map).to(dtype)
    af = None if af is None else torch.as_tensor(af).to(dtype)
    if dtype != torch.float32:
        w = {k: _t(w, k).to(dtype) for k in w}
    inter = {}
    x = multi_scale_input(w, exp_map, af)
    inter["stem"] = x
    feats = []
    for e in range(3):
        x = encoder(w, f"encoder.{e}", x)
        feats.append(x)
        inter[f"enc{e}"] = x
    f = fpn(w, feats)
    inter["fpn"] = f
    bb = head(w, "backbone_head", f)
    ca = head(w, "ca_head", torch.cat([f, bb], 1))
    aa = head(w, "aa_head", torch.cat([f, bb, ca], 1))
    if return_intermediates:
        return bb, ca, aa, inter
    return bb, ca, aa


@torch.no_grad()
def mica_forward_per_tile(w, exp_map, af=None):
    """Per-tile AF gating: the reference at batch size 1 (utils/predict.py:72,193,279),
    which is this build's definition of parity for batched tiles (SURVEY.md hard part 4)."""
    outs = [mica_forward(w, exp_map[b:b + 1], None if af is None else af[b:b + 1])
            for b in range(exp_map.shape[0])]
    return tuple(torch.cat([o[i] for o in outs], 0) for i in range(3))


@torch.no_grad()
def postprocess(bb, ca, aa):
    """utils/predict.py:342-349,358-363: drop class 1, softmax over the remaining 3, keep
    index 2 (true class 3); amino acids: softmax over classes 1..20, argmax."""
    bb = torch.as_tensor(bb); ca = torch.as_tensor(ca); aa = torch.as_tensor(aa)
    bb_s = torch.softmax(torch.cat((bb[:, :1], bb[:, 2:]), 1), dim=1)
    ca_s = torch.softmax(torch.cat((ca[:, :1], ca[:, 2:]), 1), dim=1)
    aa_s = torch.softmax(aa[:, 1:], dim=1)
    aa_pred = torch.max(aa_s, 1)[1]
    return bb_s[:, 2], ca_s[:, 2], aa_s, aa_pred
