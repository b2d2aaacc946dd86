"""Parameter table of the MICA network and a deterministic synthetic-weight generator.

The table restates the 125 tensors of the reference ``MICA().state_dict()``
(reference models/model.py:5-348; names follow torch's module nesting).  Trained
weights are a Zenodo download that is not available offline (reference
README.md:27-39), so tests and benchmarks use :func:`synth_state_dict`, a
counter-hash generator that is independent of the torch version and is
reproduced bit-for-bit on every host (numpy uint64 arithmetic only).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

BASE = 64
AF_CHANNELS = 24
STEM_KERNELS = (3, 5, 7, 9)
HEADS = (("backbone_head", 192, 4), ("ca_head", 196, 4), ("aa_head", 200, 21))


def param_shapes() -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor of the reference state_dict (125 entries)."""
    t: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, cout, cin, k):
        t[name + ".weight"] = (cout, cin, k, k, k)
        t[name + ".bias"] = (cout,)

    def linear(name, cout, cin):
        t[name + ".weight"] = (cout, cin)
        t[name + ".bias"] = (cout,)

    ip = "input_processing"
    for i, k in enumerate(STEM_KERNELS):                     # model.py:9-14
        conv(f"{ip}.exp_convs.{i}", BASE // 2, 1, k)
    conv(f"{ip}.feat_conv", BASE, AF_CHANNELS, 3)            # model.py:17
    conv(f"{ip}.exp_attention.1", BASE, 2 * BASE, 1)         # model.py:20-26
    conv(f"{ip}.exp_attention.3", 2 * BASE, BASE, 1)
    conv(f"{ip}.exp_downsizing", BASE, 2 * BASE, 1)          # model.py:28
    conv(f"{ip}.feat_gate.0", BASE // 4, BASE, 1)            # model.py:31-36
    conv(f"{ip}.feat_gate.2", 1, BASE // 4, 1)
    conv(f"{ip}.fusion", BASE, 3 * BASE, 1)                  # model.py:38
    for e, c in enumerate((BASE, 2 * BASE, 4 * BASE)):       # model.py:281-285
        p = f"encoder.{e}"
        conv(f"{p}.dense_block.conv1.0", c // 2, c, 3)       # model.py:106-126
        conv(f"{p}.dense_block.conv2.0", c // 2, 3 * c // 2, 3)
        conv(f"{p}.dense_block.conv3.0", c, 2 * c, 3)
        linear(f"{p}.dense_block.se.fc.0", c // 16, c)       # model.py:245-252
        linear(f"{p}.dense_block.se.fc.3", c, c // 16)
        t[f"{p}.dual_attn.local_attn.0.weight"] = (c, 1, 3, 3, 3)   # model.py:80 (groups=c)
        t[f"{p}.dual_attn.local_attn.0.bias"] = (c,)
        conv(f"{p}.dual_attn.global_attn.1", c // 4, c, 1)   # model.py:87-94
        conv(f"{p}.dual_attn.global_attn.4", c, c // 4, 1)
        conv(f"{p}.dual_attn.fusion", c, 2 * c, 1)           # model.py:96
        conv(f"{p}.transition.0", 2 * c, c, 3)               # model.py:141-142
    t["fpn.weights"] = (3,)                                  # model.py:179
    for i in range(3):
        conv(f"fpn.lateral.{i}", BASE, BASE * 2 * 2 ** i, 1)  # model.py:157-161
    for i in range(3):
        conv(f"fpn.smooth.{i}.0", BASE, BASE, 3)             # model.py:163-177
    for name, cin, ncls in HEADS:                            # model.py:291-293
        conv(f"{name}.conv1", 64, cin, 3)                    # model.py:210-225
        conv(f"{name}.conv2", 32, 64, 3)
        conv(f"{name}.calibration.1", 8, 32, 1)
        conv(f"{name}.calibration.4", 32, 8, 1)
        conv(f"{name}.final", ncls, 32, 1)
    assert len(t) == 125
    return t


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, n: int, seed: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (name, index, seed)."""
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base) & _M64
    bits = _splitmix64(ctr) >> np.uint64(11)                 # 53 random bits
    return bits.astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def synth_state_dict(seed: int = 2022, final_gain: float = 6.0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights: U(-b, b), b = sqrt(3/fan_in) (unit-variance preserving),
    biases U(-0.1, 0.1).  The three ``*.final`` convs get ``final_gain`` so that the
    softmax outputs are not near-uniform (hard part 6)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_shapes().items():
        n = int(np.prod(shape))
        u = hash_uniform(name, n, seed)
        if name == "fpn.weights":
            v = 0.5 * u
        elif name.endswith(".bias"):
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = math.sqrt(3.0 / fan_in) * u
            if ".final." in name:
                v = v * final_gain
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def load_checkpoint_state_dict(path: str) -> "OrderedDict[str, np.ndarray]":
    """Read a reference checkpoint (reference utils/predict.py:234-240): a dict with
    ``model_state_dict`` whose keys may carry DataParallel's ``module.`` prefix
    (reference train.py:234,298-304).  Loaded with ``weights_only=True``."""
    import torch

    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    shapes = param_shapes()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, v in sd.items():
        k = k.replace("module.", "")
        if k in shapes:                                      # strict=False semantics (predict.py:240)
            a = v.detach().to(torch.float32).cpu().numpy()
            if tuple(a.shape) != tuple(shapes[k]):
                raise ValueError(f"checkpoint tensor {k} has shape {a.shape}, expected {shapes[k]}")
            out[k] = np.ascontiguousarray(a)
    missing = [k for k in shapes if k not in out]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
    return out


def synth_state_dict_heavy(seed: int = 5, final_gain: float = 6.0, outlier: float = 30.0) -> "OrderedDict[str, np.ndarray]":
    """Heavy-tailed variant of :func:`synth_state_dict` (round 4: what a trained network could stress that U(-b, b) does not):
    every weight's magnitude is multiplied by a random power of two in 2^-4 .. 2^3 (a log-uniform spread of 128x; exact
    arithmetic, so the generator stays bit-reproducible), and about 3 % of the output channels of every conv / linear layer
    (at least one per layer with >= 16 outputs) are `outlier` times larger than the rest."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_shapes().items():
        n = int(np.prod(shape))
        u = hash_uniform(name, n, seed)
        if name == "fpn.weights":
            v = 0.5 * u
        elif name.endswith(".bias"):
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            e = np.floor(4.0 * hash_uniform(name + "#exp", n, seed))          # -4 .. 3
            v = math.sqrt(3.0 / fan_in) * 0.3 * u * np.exp2(e)
            cout = shape[0]
            if cout >= 16 and ".final." not in name:
                pick = hash_uniform(name + "#out", cout, seed)
                hot = pick < -0.94
                hot[int(np.argmin(pick))] = True
                v = (v.reshape(cout, -1) * np.where(hot, outlier, 1.0)[:, None]).reshape(-1)
            if ".final." in name:
                v = v * final_gain
        out[name] = v.astype(np.float32).reshape(shape)
    return out
