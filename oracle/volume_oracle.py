"""CPU restatement (numpy) of the reference's tiler, stitcher and map normaliser.

TEST INFRASTRUCTURE - never imported by the product path.
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


def normalise_map(data, voxel_size=(1.0, 1.0, 1.0), target_voxel_size=1.0):
    """utils/preprocessing.py:111-133 (second witness:
    scripts_for_training_data/create_normalized_map.py:37-79).  Returns (normalised f32 map,
    median, percentile) or raises ValueError where the reference logs an error."""
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
    pct = np.percentile(pos, 99.9)                                          # :128
    if pct == 0:
        raise ValueError("Percentile value is zero - cannot normalize")     # :159-161
    m = (m < pct) * m + (m >= pct) * pct                                    # :131-132
    m = m / pct                                                             # :133
    return m.astype(np.float32), float(median), float(pct)                  # :139 astype
