"""CPU restatement (numpy) of the reference's tiler, stitcher and map normaliser.

TEST INFRASTRUCTURE - never imported by the product path.

PINNED: tiler (transpose_axes, tile_volume) and normaliser (normalise_map) against the reference's own GridCreator /
DataPreprocessor / training tiler run unmodified by oracle/gen_golden_r3.py (tests/golden/tiler_ref.json, normaliser_ref.json);
the stitcher through the reference CryoEMPredictor goldens (oracle/gen_golden.py); zoom_cubic against the installed scipy.
"""
from __future__ import annotations

import numpy as np


def transpose_axes(vol, mapc, mapr, maps, nstart_zyx):
    """utils/create_grids.py:67-87,119-122.  ``vol`` is indexed as stored in the MRC
    ([section, row, column]); returns the array re-indexed to (x, y, z) order and the
    permuted start offsets."""
    axis_order = [int(maps) - 1, int(mapr) - 1, int(mapc) - 1]
    trans_offset, trans_order = [], []
    for i in range(3):
        for j in range(3):
            if axis_order[j] == i:
                trans_offset.append(float(nstart_zyx[j]))
                trans_order.append(j)
    return np.transpose(vol, trans_order), trans_offset


def tile_volume(vol, grid_size=48, padding=8):
    """utils/create_grids.py:124-176 without the disk: returns (tiles f32[T,W,W,W],
    index table int64[T,6] of (i,j,k,di,dj,dk)) in the reference's lexicographic order."""
    shape = vol.shape
    win = grid_size + 2 * padding
    pads = [(padding, win - (shape[a] % grid_size)) for a in range(3)]     # :129-139
    padded = np.pad(vol, pads, "constant")
    tiles, idx = [], []
    for i in range(0, shape[0], grid_size):                                 # :143-145
        for j in range(0, shape[1], grid_size):
            for k in range(0, shape[2], grid_size):
                g = padded[i:i + win, j:j + win, k:k + win]
                if g.shape == (win, win, win):                              # :157
                    tiles.append(g)
                    idx.append((i, j, k, min(grid_size, shape[0] - i),
                                min(grid_size, shape[1] - j), min(grid_size, shape[2] - k)))
    return np.stack(tiles).astype(vol.dtype, copy=False), np.asarray(idx, dtype=np.int64)


def stitch_volume(tile_data, idx, orig_shape, padding=8):
    """utils/predict.py:459-501: scatter the central region of every tile.
    tile_data is [T,W,W,W] or [T,C,W,W,W]."""
    tile_data = np.asarray(tile_data)
    p = padding
    if tile_data.ndim == 5:
        vol = np.zeros((tile_data.shape[1], *orig_shape), dtype=np.float32)
        for t, (i, j, k, di, dj, dk) in enumerate(idx):
            vol[:, i:i + di, j:j + dj, k:k + dk] = tile_data[t][:, p:p + di, p:p + dj, p:p + dk]
    else:
        vol = np.zeros(tuple(orig_shape), dtype=np.float32)
        for t, (i, j, k, di, dj, dk) in enumerate(idx):
            vol[i:i + di, j:j + dj, k:k + dk] = tile_data[t][p:p + di, p:p + dj, p:p + dk]
    return vol


SPLINE_POLE = float.fromhex("-0x1.126145e9ecd56p-2")     # correctly rounded sqrt(3) - 2 (what scipy uses)


def _spline_prefilter_axis(c, axis):
    c = np.moveaxis(c, axis, -1)
    n = c.shape[-1]
    if n >= 2:
        z = SPLINE_POLE
        c *= (1.0 - z) * (1.0 - 1.0 / z)
        z_n_1 = z ** (n - 1)
        c0 = c[..., 0] + z_n_1 * c[..., n - 1]
        zi = z
        for i in range(1, n - 1):
            c0 = c0 + zi * (c[..., i] + z_n_1 * c[..., n - 1 - i])
            zi *= z
        c[..., 0] = c0 / (1.0 - z_n_1 * z_n_1)
        for i in range(1, n):
            c[..., i] += z * c[..., i - 1]
        c[..., n - 1] = (z * c[..., n - 2] + c[..., n - 1]) * z / (z * z - 1.0)
        for i in range(n - 2, -1, -1):
            c[..., i] = z * (c[..., i + 1] - c[..., i])
    return np.moveaxis(c, -1, axis)


def zoom_cubic(data, factors):
    """Restatement of scipy.ndimage.zoom(data, factors, order=3) (mode='constant', cval=0, prefilter=True,
    grid_mode=False) - the call at utils/preprocessing.py:117.  scipy is a third-party dependency of the reference
    (environment.yml:13 pins 1.5.2, this image has 1.15.3) whose C source is not in the tree; this follows the published
    algorithm and is pinned bit-exact against scipy 1.15.3 by tests/test_cpu_oracle.py.  float64 internally."""
    import math
    data = np.asarray(data)
    n = data.shape
    out_shape = tuple(int(round(a * b)) for a, b in zip(n, factors))
    f = data.astype(np.float64)
    for ax in range(3):
        f = _spline_prefilter_axis(f, ax)
    idx, wts, ok = [], [], []
    for a in range(3):
        zf = (n[a] - 1) / (out_shape[a] - 1) if out_shape[a] > 1 else 1.0
        cc = np.arange(out_shape[a], dtype=np.float64) * zf
        ok.append(~(cc > n[a] - 1))                       # scipy: a coordinate past the edge yields cval
        fl = np.floor(cc)
        x = cc - fl
        y, zc = x, 1.0 - x
        w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w2 = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0
        w0 = zc * zc * zc / 6.0
        w3 = 1.0 - w0 - w1 - w2
        wts.append(np.stack([w0, w1, w2, w3]))
        st = fl.astype(np.int64) - 1
        ia = []
        for k in range(4):
            i = st + k
            if n[a] == 1:
                i = np.zeros_like(i)
            else:
                p = 2 * (n[a] - 1)
                i = np.mod(i, p)
                i = np.where(i < n[a], i, p - i)
            ia.append(i)
        idx.append(np.stack(ia))
    out = np.zeros(out_shape, dtype=np.float64)
    for k0 in range(4):
        for k1 in range(4):
            for k2 in range(4):
                c = f[np.ix_(idx[0][k0], idx[1][k1], idx[2][k2])]
                c = c * wts[0][k0][:, None, None]
                c = c * wts[1][k1][None, :, None]
                c = c * wts[2][k2][None, None, :]
                out = out + c
    out = out * (ok[0][:, None, None] & ok[1][None, :, None] & ok[2][None, None, :])
    return out.astype(data.dtype if data.dtype.kind == "f" else np.float64)


def normalise_map(data, voxel_size=(1.0, 1.0, 1.0), target_voxel_size=1.0, numpy_legacy=False):
    """utils/preprocessing.py:111-133 (second witness:
    scripts_for_training_data/create_normalized_map.py:37-79).  Returns (normalised f32 map,
    median, percentile) or raises ValueError where the reference logs an error.

    numpy_legacy=True states what the SAME lines compute under numpy 1.x (the reference pins 1.19.1, environment.yml:8; this
    container has 2.2.6, so the statement below is written out explicitly and is NOT pinned by a run - parity unpinned):
      * np.percentile, 'linear' (numpy 1.19 lib/function_base.py, _quantile_ureduce_func): indices = q * (Nx - 1) in float64,
        weights_above = indices - floor(indices), r = x_below * weights_below + x_above * weights_above - a float64 for every
        input dtype (0-d operands promote like scalars), not numpy 2's float32 _lerp;
      * value-based casting: `map < p` and `map >= p` compare in float32 with p rounded to float32 (same kind: the array wins);
        `(map >= p) * p` is bool array x float64 scalar = a float64 array (the scalar's kind is higher), so the sum :131-132 and
        the division :133 are float64, rounded once at astype(float32) :139."""
    from scipy.ndimage import zoom

    zf = [voxel_size[0] / target_voxel_size, voxel_size[1] / target_voxel_size,
          voxel_size[2] / target_voxel_size]
    res = zoom(data, zf, order=3)                                           # :117
    norm = np.nan_to_num(res)                                               # :122
    median = np.median(norm)                                                # :123
    m = (norm > median) * (norm - median)                                   # :124
    pos = m[np.where(m > 0)]
    if len(pos) == 0:
        raise ValueError("No positive values found after thresholding")     # :163-165
    if numpy_legacy:
        srt = np.sort(pos.astype(np.float64) if pos.dtype.kind != "f" else pos)
        nx = len(srt)
        ind = (99.9 / 100.0) * (nx - 1)
        below = int(np.floor(ind))
        above = min(below + 1, nx - 1)
        w_above = ind - below
        w_below = 1.0 - w_above
        pct = np.float64(srt[below]) * w_below + np.float64(srt[above]) * w_above
        if pct == 0:
            raise ValueError("Percentile value is zero - cannot normalize")
        if m.dtype == np.float32:
            p32 = np.float32(pct)
            out = ((m < p32) * m).astype(np.float64) + (m >= p32) * np.float64(pct)
        else:
            out = (m < pct) * m + (m >= pct) * pct
        out = out / np.float64(pct)
        return out.astype(np.float32), float(median), float(pct)
    pct = np.percentile(pos, 99.9)                                          # :128
    if pct == 0:
        raise ValueError("Percentile value is zero - cannot normalize")     # :159-161
    m = (m < pct) * m + (m >= pct) * pct                                    # :131-132
    m = m / pct                                                             # :133
    return m.astype(np.float32), float(median), float(pct)                  # :139 astype
