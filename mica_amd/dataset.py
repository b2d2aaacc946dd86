"""Tile-file reader mirroring CryoEMTestDataset (reference dataset/dataset.py:179-224) for callers
that still hand tiles over as .npz files (the on-disk "wire" format of SURVEY.md section 8b)."""
from __future__ import annotations

import numpy as np

AF3_TYPES = ['CA', 'N', 'C', 'O', 'ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU', 'MET',
             'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR']      # dataset.py:184-188


def read_npz_grid(path):
    """The `grid` array of a tile file, fast: np.savez stores members uncompressed in keyword order (reference
    utils/create_grids.py:163 passes grid first), so the array sits behind one ZIP local header and one .npy header at the
    start of the file and can be read with a single np.fromfile - no zipfile / pickle machinery under the GIL (25 files per
    tile: the generic np.load path caps a reader thread pool at ~40 tiles/s).  Anything unexpected (compressed member, other
    first member, Fortran order, object dtype) falls back to np.load."""
    import ast
    import struct
    try:
        with open(path, "rb") as f:
            head = f.read(1024)
        sig, _ver, _flag, method, _t, _d, _crc, _cs, _us, nlen, xlen = struct.unpack("<IHHHHHIIIHH", head[:30])
        if sig != 0x04034B50 or method != 0 or head[30:30 + nlen] != b"grid.npy":
            raise ValueError("not a stored grid.npy first member")
        o = 30 + nlen + xlen
        if head[o:o + 6] != b"\x93NUMPY":
            raise ValueError("no npy magic")
        major = head[o + 6]
        if major == 1:
            hlen, o2 = struct.unpack("<H", head[o + 8:o + 10])[0], o + 10
        else:
            hlen, o2 = struct.unpack("<I", head[o + 8:o + 12])[0], o + 12
        meta = ast.literal_eval(head[o2:o2 + hlen].decode("latin1"))
        dt = np.dtype(meta["descr"])
        if meta["fortran_order"] or dt.hasobject:
            raise ValueError("unsupported layout")
        shape = tuple(meta["shape"])
        a = np.fromfile(path, dtype=dt, count=int(np.prod(shape)), offset=o2 + hlen)
        if a.size != int(np.prod(shape)):
            raise ValueError("truncated")
        return a.reshape(shape)
    except Exception:
        return np.load(path)['grid']


class CryoEMTestDataset:
    def __init__(self, data_dir, transform=None):
        self.data_dir = list(data_dir)
        self.transform = transform
        self.types = list(AF3_TYPES)

    def __len__(self):
        return len(self.data_dir)

    def __getitem__(self, idx):
        """-> (map f32[1,W,W,W], af f32[24,W,W,W], metadata) as numpy arrays; any failure loading the
        24 AF3 tiles yields zeros, like the reference (dataset.py:209-219)."""
        map_path = self.data_dir[idx]
        data = np.load(map_path)
        grid = data['grid']
        metadata = {k: data[k] for k in ('i', 'j', 'k', 'di', 'dj', 'dk', 'orig_shape')}
        metadata['filename'] = map_path.split("/")[-1].split(".")[0]
        try:
            feats = []
            for t in self.types:
                p = map_path.replace('normalized_map_grids', f"AF3_encoding_grids/{t}_grids")
                p = p.replace('normalized_map', f"{t}")
                feats.append(read_npz_grid(p))
            af = np.stack(feats, axis=0)
        except Exception:
            af = np.zeros((24,) + grid.shape)
        return grid[None].astype(np.float32, copy=False), af.astype(np.float32, copy=False), metadata
