"""Thin object wrapper over the C ABI: one Engine = one mica_ctx = one GPU.

PyTorch is used for device memory and streams only (tensor.data_ptr() is what crosses the ABI)."""
from __future__ import annotations

import ctypes as C
import functools
import threading

import numpy as np
import torch

from . import _cabi
from ._cabi import MicaHipError, AF_NONE, AF_PER_TILE, AF_BATCH, AF_ALWAYS  # noqa: F401


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or not t.is_cuda:
        raise MicaHipError(f"{name}: expected a float32 CUDA(HIP) tensor, got {t.dtype} on {t.device}")
    return t.contiguous()


class Engine:
    def __init__(self, device: int | str | torch.device = 0, max_batch: int = 1, tile_size=64, conv_variant: int | None = None):
        """tile_size: an int (cubic tiles, what the tiler / predictor use) or (D, H, W) for the inner boundary alone
        (`forward_logits`: MICA.forward is size-agnostic).
        conv_variant: None = the library default (3, or MICA_F43 from the environment), 0 = every 3x3x3 conv on the F(2,3) kernel,
        1 = encoder.2's four 3x3x3 convs on the F(4,3) kernel, 2 = those and encoder.1's transition (+0.7 % throughput, rms error
        +1-2 %), 3 = mode 1 and the late narrow layers (FPN smooth convs, the heads' conv1) on the kernel's 64-channel variant.
        Fixed before the weights are loaded."""
        if not torch.cuda.is_available():
            raise MicaHipError("no HIP device visible: mica_amd runs on MI355X (gfx950) only, there is no CPU fallback")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise MicaHipError(f"device {device!r} is not a GPU; mica_amd has no CPU path")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.lib = _cabi.load_library()
        self.call_lock = threading.RLock()        # see _serialise_engine_methods below
        self.max_batch = int(max_batch)
        self.tile_shape = tuple(int(v) for v in tile_size) if isinstance(tile_size, (tuple, list)) else (int(tile_size),) * 3
        if len(self.tile_shape) != 3:
            raise MicaHipError(f"tile_size must be an int or (D, H, W), got {tile_size!r}")
        if conv_variant is not None and int(conv_variant) not in (0, 1, 2, 3):    # before the context (tens of GB of workspace) exists
            raise MicaHipError(f"conv_variant must be None, 0, 1, 2 or 3, got {conv_variant!r}")
        self.tile_size = self.tile_shape[0] if len(set(self.tile_shape)) == 1 else None      # None: non-cubic, forward_logits only
        h = C.c_void_p()
        with torch.cuda.device(self.device):          # the ambient current device of the caller is left alone
            r = self.lib.mica_create_dims(self.device.index, self.max_batch, *self.tile_shape, C.byref(h))
        if r != 0:
            raise MicaHipError(f"mica_create failed ({r}): {self.lib.mica_last_error(None).decode()}")
        self._h = h
        try:
            if conv_variant is not None:
                self._check(self.lib.mica_set_conv_variant(self._h, int(conv_variant)), "mica_set_conv_variant")
            self.conv_variant = int(self.lib.mica_get_conv_variant(self._h))
        except Exception:
            self.close()                   # a context that cannot be configured must not keep its workspace
            raise
        self.weights_loaded = False
        self.last_forward_scale = 16.0     # lowest activation scale any chunk of the last forward_* call needed
        self.forward_retries = 0           # tiles of the last forward_* call that were repeated at a lower activation scale
        self.last_input_runs = 0           # runs of equal AF3 gate its last library call was cut into (MultiScaleInput launches per run)
        self.last_af_tiles = 0

    # -- plumbing -------------------------------------------------------------------------------
    def _stream(self):
        """torch's current stream of THIS engine's device (the C side does hipSetDevice(ctx->device))."""
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, r, what):
        if r != 0:
            raise MicaHipError(f"{what} failed ({r}): {self.lib.mica_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mica_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mica_workspace_bytes(self._h))

    # -- weights ----------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: {name: float32 array/tensor} (125 tensors, optional 'module.' prefix)."""
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.mica_load_weight(self._h, k.encode(), a.ctypes.data_as(_cabi._FP), shape, a.ndim),
                        f"mica_load_weight({k})")
        self._check(self.lib.mica_finalize_weights(self._h), "mica_finalize_weights")
        self.weights_loaded = True

    # -- forward ----------------------------------------------------------------------------------
    LOW_SCALE_WARN = 0.25      # below this the split encoding loses bits: the call still succeeds, loudly

    def _begin_forward(self):
        """A public forward_* call may be cut into several C calls (T > max_batch): the scale reported is the lowest of them."""
        self.last_forward_scale = float(self.lib.mica_get_activation_scale(self._h))
        self.forward_retries = 0

    def _check_forward(self, r, what):
        self._check(r, what)
        sc = float(self.lib.mica_get_last_forward_scale(self._h))
        self.forward_retries += int(self.lib.mica_get_last_forward_retries(self._h))
        self.last_input_runs = int(self.lib.mica_get_last_forward_input_runs(self._h))      # of the last library call of this forward
        self.last_af_tiles = int(self.lib.mica_get_last_forward_af_tiles(self._h))          # its tiles that took the AF3 branch
        self.last_forward_scale = min(self.last_forward_scale, sc)
        if sc < self.LOW_SCALE_WARN:
            import warnings
            warnings.warn(f"{what}: activations of a tile exceeded {60000 / self.LOW_SCALE_WARN:.0f}; it was computed at activation "
                          f"scale {sc} where the whole-network error can exceed 1e-4 (up to ~2e-4 measured)", RuntimeWarning, stacklevel=3)

    def _batch_mode(self, n: int, af, af_mode: int):
        """-> (af, af_mode) for the per-call chunks.  The batch-wide AF3 test (models/model.py:60: af.abs().sum() < 1e-6) is a
        property of ONE forward call of the reference; a batch beyond max_batch is cut into several library calls, so the test is
        evaluated here over the whole batch and the chunks are told its outcome (AF_ALWAYS, or no AF3 input at all)."""
        if af is None:
            return None, AF_NONE
        if af_mode == AF_BATCH and n > self.max_batch:
            # the library's own per-tile reduction (no |af| temporary of the batch's size), combined exactly as forward_impl combines
            # it for MICA_AF_BATCH: f32 per-tile sums added in double - the same decision on either side of max_batch
            sums = (C.c_float * n)()
            self._check(self.lib.mica_af_abs_sums(self._h, _ptr(af), n, sums, self._stream()), "mica_af_abs_sums")
            tot = 0.0
            for v in sums:
                tot += float(v)
            return (None, AF_NONE) if tot < 1e-6 else (af, AF_ALWAYS)
        return af, af_mode

    def _cubic(self, what):
        if self.tile_size is None:
            raise MicaHipError(f"{what}: this engine was built for non-cubic tiles {self.tile_shape}; only forward_logits takes those")
        return self.tile_size

    def forward_logits(self, exp_map: torch.Tensor, af: torch.Tensor | None, af_mode: int = AF_BATCH):
        dims = self.tile_shape
        exp_map = _f32c(exp_map, "exp_map")
        B = exp_map.shape[0]
        if tuple(exp_map.shape) != (B, 1, *dims):
            raise MicaHipError(f"exp_map must be [B,1,{dims[0]},{dims[1]},{dims[2]}], got {tuple(exp_map.shape)}")
        if af is not None:
            af = _f32c(af, "af_features")
            if tuple(af.shape) != (B, 24, *dims):
                raise MicaHipError(f"af_features must be [B,24,{dims[0]},{dims[1]},{dims[2]}], got {tuple(af.shape)}")
        af, af_mode = self._batch_mode(B, af, af_mode)
        bb = torch.empty((B, 4, *dims), dtype=torch.float32, device=self.device)
        ca = torch.empty_like(bb)
        aa = torch.empty((B, 21, *dims), dtype=torch.float32, device=self.device)
        self._begin_forward()
        for b0 in range(0, B, self.max_batch):
            b1 = min(B, b0 + self.max_batch)
            self._check_forward(self.lib.mica_forward_logits(
                self._h, _ptr(exp_map[b0:b1]), _ptr(af[b0:b1]) if af is not None else None, b1 - b0,
                af_mode, _ptr(bb[b0:b1]), _ptr(ca[b0:b1]), _ptr(aa[b0:b1]), self._stream()),
                "mica_forward_logits")
        return bb, ca, aa

    def forward_tiles(self, map_tiles: torch.Tensor, af_tiles: torch.Tensor | None, out=None, af_mode: int = AF_PER_TILE):
        """map_tiles f32[T,S^3-shaped], af_tiles f32[T,24,...] or None -> (bb_prob[T,S,S,S], ca_prob, aa_prob[T,20,S,S,S], aa_pred)."""
        S = self._cubic("forward_tiles")
        T = map_tiles.shape[0]
        map_tiles = _f32c(map_tiles, "map_tiles").view(T, 1, S, S, S)
        if af_tiles is not None:
            af_tiles = _f32c(af_tiles, "af_tiles").view(T, 24, S, S, S)
        af_tiles, af_mode = self._batch_mode(T, af_tiles, af_mode)
        if out is None:
            out = (torch.empty((T, S, S, S), dtype=torch.float32, device=self.device),
                   torch.empty((T, S, S, S), dtype=torch.float32, device=self.device),
                   torch.empty((T, 20, S, S, S), dtype=torch.float32, device=self.device),
                   torch.empty((T, S, S, S), dtype=torch.float32, device=self.device))
        bbp, cap, aap, pred = out
        self._begin_forward()
        for b0 in range(0, T, self.max_batch):
            b1 = min(T, b0 + self.max_batch)
            self._check_forward(self.lib.mica_forward_tiles(
                self._h, _ptr(map_tiles[b0:b1]), _ptr(af_tiles[b0:b1]) if af_tiles is not None else None, b1 - b0,
                af_mode, _ptr(bbp[b0:b1]), _ptr(cap[b0:b1]), _ptr(aap[b0:b1]),
                _ptr(pred[b0:b1]), self._stream()), "mica_forward_tiles")
        return out

    def forward_records(self, map_tiles: torch.Tensor, af_tiles: torch.Tensor | None, rec: torch.Tensor, af_mode: int = AF_PER_TILE):
        """As forward_tiles, written straight into rec f32[T,23,S,S,S] (bb, ca, aa_pred, aa_prob x20 per tile): the record layout
        stitch_tiles and the multi-GPU exchange use."""
        S = self._cubic("forward_records")
        T = map_tiles.shape[0]
        map_tiles = _f32c(map_tiles, "map_tiles").view(T, 1, S, S, S)
        if af_tiles is not None:
            af_tiles = _f32c(af_tiles, "af_tiles").view(T, 24, S, S, S)
        af_tiles, af_mode = self._batch_mode(T, af_tiles, af_mode)
        if rec.dtype != torch.float32 or not rec.is_cuda or not rec.is_contiguous() or tuple(rec.shape) != (T, 23, S, S, S):
            raise MicaHipError(f"rec must be a contiguous float32 CUDA(HIP) tensor [{T},23,{S},{S},{S}]")
        self._begin_forward()
        for b0 in range(0, T, self.max_batch):
            b1 = min(T, b0 + self.max_batch)
            self._check_forward(self.lib.mica_forward_records(
                self._h, _ptr(map_tiles[b0:b1]), _ptr(af_tiles[b0:b1]) if af_tiles is not None else None, b1 - b0,
                af_mode, _ptr(rec[b0:b1]), self._stream()), "mica_forward_records")
        return rec

    def postprocess(self, bb, ca, aa):
        B = bb.shape[0]
        S = self._cubic("postprocess")
        bb, ca, aa = _f32c(bb, "bb"), _f32c(ca, "ca"), _f32c(aa, "aa")
        o = (torch.empty((B, S, S, S), dtype=torch.float32, device=self.device),
             torch.empty((B, S, S, S), dtype=torch.float32, device=self.device),
             torch.empty((B, 20, S, S, S), dtype=torch.float32, device=self.device),
             torch.empty((B, S, S, S), dtype=torch.float32, device=self.device))
        self._check(self.lib.mica_postprocess(self._h, _ptr(bb), _ptr(ca), _ptr(aa), B, *[_ptr(t) for t in o], self._stream()),
                    "mica_postprocess")
        return o

    # -- tiler / stitch / normalise -------------------------------------------------------------------
    def gather_tiles(self, vol: torch.Tensor, grid: int, pad: int, first: int, count: int, out=None):
        """vol f32 or uint8 [C,N0,N1,N2] (or [N0,N1,N2]) -> tiles f32[count,C,W,W,W].  uint8: binary AF3 encodings kept at a quarter
        of the memory."""
        if vol.dtype == torch.uint8:
            if not vol.is_cuda:
                raise MicaHipError(f"vol: expected a CUDA(HIP) tensor, got {vol.device}")
            vol = vol.contiguous()
            fn = self.lib.mica_gather_tiles_u8
        else:
            vol = _f32c(vol, "vol")
            fn = self.lib.mica_gather_tiles
        if vol.dim() == 3:
            vol = vol[None]
        Cc, n0, n1, n2 = vol.shape
        W = grid + 2 * pad
        if out is None:
            out = torch.empty((count, Cc, W, W, W), dtype=torch.float32, device=self.device)
        self._check(fn(self._h, _ptr(vol), Cc, n0, n1, n2, grid, pad, first, count, _ptr(out), self._stream()), "mica_gather_tiles")
        return out

    def stitch_tiles(self, tiles: torch.Tensor, vol: torch.Tensor, grid: int, pad: int, first: int):
        """tiles f32[count,C,W,W,W] -> central regions into vol f32[C,N0,N1,N2] (in place)."""
        tiles = _f32c(tiles, "tiles")
        if not vol.is_contiguous():
            raise MicaHipError("vol must be contiguous")
        v4 = vol if vol.dim() == 4 else vol[None]
        Cc, n0, n1, n2 = v4.shape
        self._check(self.lib.mica_stitch_tiles(self._h, _ptr(tiles), Cc, n0, n1, n2, grid, pad, first, tiles.shape[0],
                                               _ptr(v4), self._stream()), "mica_stitch_tiles")
        return vol

    def normalise_map_(self, vol: torch.Tensor, map_type: int = 0, numpy_legacy: bool = False):
        """In place; returns (median, percentile).  Raises MicaHipError like the reference logs failure.
        map_type: MICA_MAP_* of include/mica_hip.h (0 float32 map; 1/2/3 = int8/int16/uint16 values held as f32).
        numpy_legacy: numpy 1.x arithmetic (float64 percentile weights, float64 clip / divide) instead of numpy 2's."""
        vol = _f32c(vol, "vol")
        st = (C.c_double * 2)()
        self._check(self.lib.mica_normalise_map_np(self._h, _ptr(vol), vol.numel(), int(map_type), int(bool(numpy_legacy)), st, self._stream()),
                    "mica_normalise_map")
        return float(st[0]), float(st[1])

    def zoom_cubic(self, vol: torch.Tensor, factors, map_type: int = 0):
        """scipy.ndimage.zoom(vol, factors, order=3) on the GPU (bit-exact); vol f32[N0,N1,N2] -> f32[round(N*f)]."""
        vol = _f32c(vol, "vol")
        n = tuple(vol.shape)
        if len(n) != 3 or len(factors) != 3:
            raise MicaHipError("zoom_cubic: 3-D volume and three factors expected")
        o = tuple(int(round(a * b)) for a, b in zip(n, factors))        # scipy's output_shape rule
        if min(o) < 1:
            raise MicaHipError(f"zoom_cubic: empty output shape {o}")
        out = torch.empty(o, dtype=torch.float32, device=self.device)
        self._check(self.lib.mica_zoom_cubic_typed(self._h, _ptr(vol), *n, *o, int(map_type), _ptr(out), self._stream()),
                    "mica_zoom_cubic")
        return out

    def rasterise_atoms(self, xyz: torch.Tensor, bb: torch.Tensor, aa: torch.Tensor, origin, shape):
        """Atoms -> f32[24, nz, ny, nx] AF3 encoding (preprocessing.py:283-298).  xyz f32[n,3], bb/aa int32[n] channel or -1;
        origin = header origin (x, y, z); shape = (nz, ny, nx).  Raises MicaHipError where the reference raises IndexError."""
        xyz = _f32c(xyz, "xyz")
        n = xyz.shape[0]
        for t, name in ((bb, "bb"), (aa, "aa")):
            if t.dtype != torch.int32 or not t.is_contiguous() or t.device != xyz.device or t.numel() != n:
                raise MicaHipError(f"{name} must be a contiguous int32 tensor of {n} entries on {xyz.device}")
        nz, ny, nx = (int(v) for v in shape)
        out = torch.empty((24, nz, ny, nx), dtype=torch.float32, device=self.device)
        org = (C.c_float * 3)(*[float(np.float32(v)) for v in origin])
        self._check(self.lib.mica_rasterise_atoms(self._h, _ptr(xyz), _ptr(bb), _ptr(aa), n, org, nz, ny, nx, _ptr(out), self._stream()),
                    "mica_rasterise_atoms")
        return out

    # -- point lists for Solver.clustering (modeler.py:762-858) -------------------------------------------------
    def threshold_points(self, vol: torch.Tensor, thr: float, capacity: int | None = None) -> torch.Tensor:
        """np.array(np.where(vol > thr)).T as ascending linear indices (int64 device tensor); modeler.py:767."""
        vol = _f32c(vol, "vol")
        n = vol.numel()
        cap = n if capacity is None else int(capacity)
        while True:
            idx = torch.empty((max(cap, 1),), dtype=torch.int64, device=self.device)
            cnt = (C.c_int64 * 1)()
            self._check(self.lib.mica_threshold_points(self._h, _ptr(vol), n, float(thr), _ptr(idx), cap, cnt, self._stream()),
                        "mica_threshold_points")
            if cnt[0] <= cap:
                return idx[:cnt[0]]
            cap = int(cnt[0])

    def gather_values(self, vol: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """vol f32[N0,N1,N2] or [C,N0,N1,N2], idx int64[n] linear voxel indices -> f32[n] or [C,n]."""
        vol = _f32c(vol, "vol")
        c = vol.shape[0] if vol.dim() == 4 else 1
        nvox = vol.numel() // c
        if idx.dtype != torch.int64 or not idx.is_cuda:
            raise MicaHipError("idx: expected an int64 CUDA(HIP) tensor")
        idx = idx.contiguous()
        out = torch.empty((c, idx.numel()), dtype=torch.float32, device=self.device)
        self._check(self.lib.mica_gather_values(self._h, _ptr(vol), c, nvox, _ptr(idx), idx.numel(), _ptr(out), self._stream()),
                    "mica_gather_values")
        return out if vol.dim() == 4 else out[0]

    def refine_candidates(self, ca: torch.Tensor, aa: torch.Tensor, cands: torch.Tensor):
        """modeler.py:836-852: cands int32[n,3] -> (coord f64[n,3], aa f32[n,20], ok bool[n])."""
        ca, aa = _f32c(ca, "ca"), _f32c(aa, "aa")
        if ca.dim() != 3 or aa.dim() != 4 or aa.shape[0] != 20 or tuple(aa.shape[1:]) != tuple(ca.shape):
            raise MicaHipError("refine_candidates: ca f32[N0,N1,N2] and aa f32[20,N0,N1,N2] expected")
        if cands.dtype != torch.int32 or not cands.is_cuda or cands.dim() != 2 or cands.shape[1] != 3:
            raise MicaHipError("cands: expected an int32 CUDA(HIP) tensor [n,3]")
        cands = cands.contiguous()
        n = cands.shape[0]
        coord = torch.empty((n, 3), dtype=torch.float64, device=self.device)
        aao = torch.empty((n, 20), dtype=torch.float32, device=self.device)
        ok = torch.empty((n,), dtype=torch.int32, device=self.device)
        self._check(self.lib.mica_refine_candidates(self._h, _ptr(ca), _ptr(aa), *ca.shape, _ptr(cands), n, _ptr(coord), _ptr(aao),
                                                    _ptr(ok), self._stream()), "mica_refine_candidates")
        return coord, aao, ok.bool()

    def segment_sums(self, vals: torch.Tensor, seg_off: torch.Tensor) -> torch.Tensor:
        """modeler.py:776-787: np.sum of every segment vals[seg_off[s]:seg_off[s+1]] in numpy's float32 summation order."""
        vals = _f32c(vals, "vals")
        if seg_off.dtype != torch.int64 or not seg_off.is_cuda or seg_off.dim() != 1 or seg_off.numel() < 1:
            raise MicaHipError("seg_off: expected an int64 CUDA(HIP) vector of nseg + 1 offsets")
        seg_off = seg_off.contiguous()
        nseg = seg_off.numel() - 1
        out = torch.empty((nseg,), dtype=torch.float32, device=self.device)
        self._check(self.lib.mica_segment_sums(self._h, _ptr(vals), _ptr(seg_off), nseg, _ptr(out), self._stream()), "mica_segment_sums")
        return out

    def nms_points(self, pts: torch.Tensor, shape, radius: float) -> torch.Tensor:
        """modeler.py:822-831 over candidates sorted by descending score: pts int32[n,3] -> bool[n] (kept)."""
        if pts.dtype != torch.int32 or not pts.is_cuda or pts.dim() != 2 or pts.shape[1] != 3:
            raise MicaHipError("pts: expected an int32 CUDA(HIP) tensor [n,3]")
        pts = pts.contiguous()
        n = pts.shape[0]
        keep = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self._check(self.lib.mica_nms_points(self._h, _ptr(pts), n, *[int(v) for v in shape], float(radius), _ptr(keep), self._stream()),
                    "mica_nms_points")
        return keep.bool()

    def neighbour_matrix(self, cands: torch.Tensor, bb: torch.Tensor, numpy_legacy: bool = False):
        """modeler.py:860-888: cands f64[n,3], bb f32[N0,N1,N2] -> (cand_self_dis f64[n,n], neigh_mat f64[n,n]) on the device.
        numpy_legacy: the promotion rules of numpy 1.x (the reference's pinned 1.19.1: float64 density sums) instead of numpy 2."""
        bb = _f32c(bb, "bb")
        if cands.dtype != torch.float64 or not cands.is_cuda or cands.dim() != 2 or cands.shape[1] != 3:
            raise MicaHipError("cands: expected a float64 CUDA(HIP) tensor [n,3]")
        cands = cands.contiguous()
        n = cands.shape[0]
        dis = torch.empty((n, n), dtype=torch.float64, device=self.device)
        mat = torch.empty((n, n), dtype=torch.float64, device=self.device)
        self._check(self.lib.mica_neighbour_matrix_np(self._h, _ptr(cands), n, _ptr(bb), *bb.shape, int(bool(numpy_legacy)), _ptr(dis),
                                                      _ptr(mat), self._stream()), "mica_neighbour_matrix")
        return dis, mat

    # -- single ops (tests) -----------------------------------------------------------------------------
    def op_conv3d(self, x, w, b, k, variant: int = 0):
        """variant 0: the F(2,3)-along-x kernel (and the 1x1 kernel for k = 1); 1: the F(4,3)-along-x kernel (k = 3, cout % 128 == 0)."""
        x = _f32c(x, "x")
        B, cin, d, h, ww = x.shape
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        cout = w.shape[0]
        y = torch.empty((B, cout, d, h, ww), dtype=torch.float32, device=self.device)
        self._check(self.lib.mica_op_conv3d_variant(self._h, _ptr(x), B, cin, d, h, ww, w.ctypes.data_as(_cabi._FP),
                                                    b.ctypes.data_as(_cabi._FP), cout, k, int(variant), _ptr(y), self._stream()), "mica_op_conv3d")
        return y

    def op_norm_conv1_conv3(self, x, w1, b1, w3, b3, variant: int = 0):
        """conv3x3x3(conv1x1x1(relu(instance_norm(x)))) through the fused 1x1 kernel (raw source, Winograd-operand epilogue)."""
        x = _f32c(x, "x")
        B, cin, d, h, ww = x.shape
        w1, b1, w3, b3 = (np.ascontiguousarray(a, dtype=np.float32) for a in (w1, b1, w3, b3))
        cmid, cout = w1.shape[0], w3.shape[0]
        y = torch.empty((B, cout, d, h, ww), dtype=torch.float32, device=self.device)
        fp = lambda a: a.ctypes.data_as(_cabi._FP)
        self._check(self.lib.mica_op_norm_conv1_conv3_variant(self._h, _ptr(x), B, cin, d, h, ww, fp(w1), fp(b1), cmid, fp(w3), fp(b3), cout,
                                                              int(variant), _ptr(y), self._stream()), "mica_op_norm_conv1_conv3")
        return y

    def op_instnorm_relu(self, x):
        x = _f32c(x, "x")
        B, c, d, h, w = x.shape
        y = torch.empty_like(x)
        self._check(self.lib.mica_op_instnorm_relu(self._h, _ptr(x), B, c, d, h, w, _ptr(y), self._stream()), "mica_op_instnorm_relu")
        return y

    def op_depthwise3(self, x, w, b):
        x = _f32c(x, "x")
        B, c, d, h, ww = x.shape
        w = np.ascontiguousarray(w, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        y = torch.empty_like(x)
        self._check(self.lib.mica_op_depthwise3(self._h, _ptr(x), B, c, d, h, ww, w.ctypes.data_as(_cabi._FP),
                                                b.ctypes.data_as(_cabi._FP), _ptr(y), self._stream()), "mica_op_depthwise3")
        return y

    def op_se_depthwise(self, x, dw_w, dw_b, fc0_w, fc0_b, fc3_w, fc3_b):
        """relu(IN(dwconv3(SE(relu(IN(x)))))) as the forward graph computes it (pool fused into the depthwise load, gate folded
        into the norm constants)."""
        x = _f32c(x, "x")
        B, c, d, h, ww = x.shape
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (dw_w, dw_b, fc0_w, fc0_b, fc3_w, fc3_b)]
        y = torch.empty_like(x)
        self._check(self.lib.mica_op_se_depthwise(self._h, _ptr(x), B, c, d, h, ww, *[a.ctypes.data_as(_cabi._FP) for a in arrs], _ptr(y),
                                                  self._stream()), "mica_op_se_depthwise")
        return y

    def op_stem(self, m):
        m = _f32c(m, "map")
        B, one, d, h, w = m.shape
        y = torch.empty((B, 128, d, h, w), dtype=torch.float32, device=self.device)
        self._check(self.lib.mica_op_stem(self._h, _ptr(m), B, d, h, w, _ptr(y), self._stream()), "mica_op_stem")
        return y

    @property
    def activation_scale(self) -> float:
        """Scale of the split-f16 operand encoding every forward call starts from (16 unless set).  A tile that overflows is
        repeated at a lower scale for that call only; `last_forward_scale` tells which."""
        return float(self.lib.mica_get_activation_scale(self._h))

    @activation_scale.setter
    def activation_scale(self, v: float):
        self._check(self.lib.mica_set_activation_scale(self._h, float(v)), "mica_set_activation_scale")

    def set_profiling(self, on: bool):
        self._check(self.lib.mica_set_profiling(self._h, int(on)), "mica_set_profiling")

    def conv_profile(self):
        return self.profile(0)

    def profile(self, kind: int):
        """(ms, launches, work) of the last profiled forward: kind 0 dense convs (FLOPs), 1 depthwise conv3d (bytes)."""
        ms, n, wk = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.mica_get_profile(self._h, kind, C.byref(ms), C.byref(n), C.byref(wk)), "mica_get_profile")
        return ms.value, n.value, wk.value


def _serialise_engine_methods():
    """Every public method of an Engine runs under that engine's lock: a context is not thread-safe (include/mica_hip.h), ctypes
    releases the GIL for the duration of a call, and an engine may be shared between the caller's thread and a background one (the
    tile-file writer of mica_amd/handoff.py gathers with the tiler's engine).  The lock spans the call AND the reading of its error
    text / result registers, so one thread's failure is never reported with another's message."""
    def serial(fn):
        @functools.wraps(fn)
        def method(self, *a, **k):
            with self.call_lock:
                return fn(self, *a, **k)
        return method

    for name, member in list(vars(Engine).items()):
        if callable(member) and not name.startswith("_"):
            setattr(Engine, name, serial(member))


_serialise_engine_methods()
