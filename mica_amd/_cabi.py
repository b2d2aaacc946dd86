"""ctypes binding of libmica_hip.so (include/mica_hip.h).  Fails loudly when the library is
missing: there is no CPU or PyTorch fallback behind this module."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmica_hip.so")

ABI_VERSION = 3          # mica_abi_version() of the library these signatures describe (bumped whenever an export or a struct changes)
MICA_OK = 0
MICA_ERR_ARG, MICA_ERR_HIP, MICA_ERR_STATE, MICA_ERR_RANGE = -1, -2, -3, -4
AF_NONE, AF_PER_TILE, AF_BATCH, AF_ALWAYS = 0, 1, 2, 3

_P = C.c_void_p
_I, _L, _F = C.c_int, C.c_int64, C.c_float
_FP = C.POINTER(C.c_float)
_LP = C.POINTER(C.c_int64)
_DP = C.POINTER(C.c_double)

# name -> (restype, argtypes): every symbol include/mica_hip.h declares
SIGNATURES = {
    "mica_abi_version": (_I, []),
    "mica_create": (_I, [_I, _I, _I, C.POINTER(_P)]),
    "mica_create_dims": (_I, [_I, _I, _I, _I, _I, C.POINTER(_P)]),
    "mica_destroy": (None, [_P]),
    "mica_last_error": (C.c_char_p, [_P]),
    "mica_workspace_bytes": (_L, [_P]),
    "mica_load_weight": (_I, [_P, C.c_char_p, _FP, _LP, _I]),
    "mica_finalize_weights": (_I, [_P]),
    "mica_forward_logits": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "mica_forward_tiles": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "mica_forward_records": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "mica_af_abs_sums": (_I, [_P, _P, _L, _FP, _P]),
    "mica_postprocess": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mica_tile_count": (_L, [_L, _L, _L, _I]),
    "mica_tile_table": (_L, [_L, _L, _L, _I, _LP, _L]),
    "mica_gather_tiles": (_I, [_P, _P, _I, _L, _L, _L, _I, _I, _L, _L, _P, _P]),
    "mica_gather_tiles_u8": (_I, [_P, _P, _I, _L, _L, _L, _I, _I, _L, _L, _P, _P]),
    "mica_stitch_tiles": (_I, [_P, _P, _I, _L, _L, _L, _I, _I, _L, _L, _P, _P]),
    "mica_normalise_map": (_I, [_P, _P, _L, _DP, _P]),
    "mica_zoom_cubic": (_I, [_P, _P, _L, _L, _L, _L, _L, _L, _P, _P]),
    "mica_normalise_map_typed": (_I, [_P, _P, _L, _I, _DP, _P]),
    "mica_normalise_map_np": (_I, [_P, _P, _L, _I, _I, _DP, _P]),
    "mica_zoom_cubic_typed": (_I, [_P, _P, _L, _L, _L, _L, _L, _L, _I, _P, _P]),
    "mica_rasterise_atoms": (_I, [_P, _P, _P, _P, _L, _FP, _L, _L, _L, _P, _P]),
    "mica_threshold_points": (_I, [_P, _P, _L, _F, _P, _L, _LP, _P]),
    "mica_gather_values": (_I, [_P, _P, _I, _L, _P, _L, _P, _P]),
    "mica_refine_candidates": (_I, [_P, _P, _P, _L, _L, _L, _P, _L, _P, _P, _P, _P]),
    "mica_segment_sums": (_I, [_P, _P, _P, _L, _P, _P]),
    "mica_nms_points": (_I, [_P, _P, _L, _L, _L, _L, C.c_double, _P, _P]),
    "mica_neighbour_matrix": (_I, [_P, _P, _L, _P, _L, _L, _L, _P, _P, _P]),
    "mica_neighbour_matrix_np": (_I, [_P, _P, _L, _P, _L, _L, _L, _I, _P, _P, _P]),
    "mica_op_conv3d": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _I, _I, _P, _P]),
    "mica_op_norm_conv1_conv3": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _I, _FP, _FP, _I, _P, _P]),
    "mica_op_conv3d_variant": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _I, _I, _I, _P, _P]),
    "mica_op_norm_conv1_conv3_variant": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _I, _FP, _FP, _I, _I, _P, _P]),
    "mica_set_conv_variant": (_I, [_P, _I]),
    "mica_get_conv_variant": (_I, [_P]),
    "mica_op_instnorm_relu": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "mica_op_depthwise3": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _P, _P]),
    "mica_op_se_depthwise": (_I, [_P, _P, _I, _I, _I, _I, _I, _FP, _FP, _FP, _FP, _FP, _FP, _P, _P]),
    "mica_op_stem": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "mica_get_activation_scale": (_F, [_P]),
    "mica_get_last_forward_scale": (_F, [_P]),
    "mica_get_last_forward_retries": (_I, [_P]),
    "mica_get_last_forward_input_runs": (_I, [_P]),
    "mica_get_last_forward_af_tiles": (_I, [_P]),
    "mica_set_activation_scale": (_I, [_P, _F]),
    "mica_set_profiling": (_I, [_P, _I]),
    "mica_get_conv_profile": (_I, [_P, _DP, _LP, _DP]),
    "mica_get_profile": (_I, [_P, _I, _DP, _LP, _DP]),
}

_lib = None


def source_hash() -> str:
    """SHA-256 over the library's sources (mica_amd/csrc/*.hip, common.h, Makefile and include/mica_hip.h, in name order).
    tools/profile.sh stamps every rocprofv3 summary it writes under profiles/ with it and bench.py only quotes counter figures whose
    stamp equals the running tree's (.git does not travel to the GPU box, so a content hash stands in for the git tree id)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    names = sorted(f for f in os.listdir(src) if f.endswith((".hip", ".h")) or f == "Makefile")
    for f in [os.path.join(src, n) for n in names] + [os.path.join(os.path.dirname(_HERE), "include", "mica_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class MicaHipError(RuntimeError):
    pass


def load_library(path: str | None = None):
    """dlopen libmica_hip.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch first: PyTorch-ROCm ships its own HIP runtime, and the process must hold ONE copy of it.  Loaded the other way round
    # (this library first, pulling the system libamdhip64, then torch with its bundled one) the second runtime reports "no
    # ROCm-capable device" to whoever got it.
    import torch  # noqa: F401

    p = path or os.environ.get("MICA_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise MicaHipError(
            f"{p} not found: build it with `make -C mica_amd/csrc` (or __graft_entry__.build()). "
            "mica_amd has no CPU fallback.")
    lib = C.CDLL(p)
    lib.mica_abi_version.restype = C.c_int
    v = lib.mica_abi_version()
    if v != ABI_VERSION:
        raise MicaHipError(f"{p} has ABI version {v}, this package binds version {ABI_VERSION}: a stale build - rebuild it with "
                           "`make -C mica_amd/csrc` (the library is git-ignored, it does not follow a checkout)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def tile_table(n0: int, n1: int, n2: int, grid: int):
    """Host-side tile index table int64[T,6] = (i,j,k,di,dj,dk) (no GPU needed)."""
    import numpy as np

    lib = load_library()
    t = lib.mica_tile_count(n0, n1, n2, grid)
    if t < 0:
        raise MicaHipError("mica_tile_count: bad argument")
    tab = np.empty((t, 6), dtype=np.int64)
    r = lib.mica_tile_table(n0, n1, n2, grid, tab.ctypes.data_as(_LP), t)
    if r != t:
        raise MicaHipError("mica_tile_table failed")
    return tab
