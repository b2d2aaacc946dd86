"""`MICA` with the reference's inner boundary (reference models/model.py:260-348): construct, load a
state_dict, call forward(exp_map, af_features) -> (backbone, ca, aa) logits.  All arithmetic runs in
libmica_hip.so on an MI355X; there is no CPU path."""
from __future__ import annotations

import torch

from .engine import AF_BATCH, Engine, MicaHipError
from .weights import param_shapes


class MICA:
    def __init__(self, base_filters: int = 64, dropout_schedule=None, max_batch: int = 8, max_cached_shapes: int = 1):
        """max_cached_shapes: engines (one per tile shape, each holding its own workspace: 37 GB at 8 tiles of 64^3, more for larger
        boxes) kept alive; the least recently used one is closed when another shape arrives.  Default 1: the predictor feeds one tile
        shape, and a caller that cycles through box shapes should not pin a workspace per shape without asking for it (a new shape
        costs a workspace allocation and a weight upload + repack, ~0.1 s)."""
        if base_filters != 64:
            raise MicaHipError("only base_filters=64 (the reference default, model.py:261) is built")
        self.max_batch = max_batch
        self.max_cached_shapes = max(1, int(max_cached_shapes))
        self._sd = None
        self._engines = {}
        self.device = None
        self.training = False

    # -- torch.nn.Module look-alikes used by the reference's predictor (utils/predict.py:233-241) ----
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise MicaHipError(f"MICA.to({device!r}): this build runs on MI355X only")
        self.device = torch.device("cuda", dev.index or 0)
        return self

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        shapes = param_shapes()
        sd = {}
        for k, v in state_dict.items():
            k = k.replace("module.", "")
            if k in shapes:
                sd[k] = v
        missing = [k for k in shapes if k not in sd]
        if missing:
            # strict=False in the reference silently keeps random init for missing keys (predict.py:240);
            # a partially initialised network is never what the caller wants, so fail loudly.
            raise MicaHipError(f"state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self._sd = sd
        for e in self._engines.values():
            e.close()
        self._engines = {}
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def _engine(self, dims) -> Engine:
        if self._sd is None:
            raise MicaHipError("MICA: load_state_dict() first (no trained weights ship with the package)")
        dims = tuple(int(v) for v in dims)
        if dims in self._engines:
            self._engines[dims] = self._engines.pop(dims)          # most recently used last (dicts keep insertion order)
        else:
            while len(self._engines) >= self.max_cached_shapes:
                self._engines.pop(next(iter(self._engines))).close()
            e = Engine(self.device or 0, max_batch=self.max_batch, tile_size=dims)
            e.load_state_dict(self._sd)
            self._engines[dims] = e
        return self._engines[dims]

    def forward(self, exp_map: torch.Tensor, af_features: torch.Tensor | None = None):
        """exp_map f32[B,1,D,H,W], af_features f32[B,24,D,H,W] or None -> three NCDHW logit tensors (any box with edges in
        [4, 128], as the reference's fully convolutional forward; one engine is kept per tile shape).
        The AF3 gate is batch-wide, exactly as model.py:60."""
        if exp_map.dim() != 5 or exp_map.shape[1] != 1:
            raise MicaHipError(f"exp_map must be [B,1,D,H,W], got {tuple(exp_map.shape)}")
        e = self._engine(exp_map.shape[2:])
        dev = e.device
        af = None if af_features is None else af_features.to(dev, torch.float32)
        return e.forward_logits(exp_map.to(dev, torch.float32), af, AF_BATCH)

    __call__ = forward
