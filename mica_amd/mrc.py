"""Minimal MRC2014 reader/writer (the `mrcfile` package is not available offline).

Covers what the hot path touches (reference utils/create_grids.py:108-117,
utils/preprocessing.py:98-107,138-148): data, voxel_size, origin, mapc/mapr/maps, n[xyz]start.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

_MODES = {0: np.int8, 1: np.int16, 2: np.float32, 6: np.uint16, 12: np.float16}
_MODE_OF = {np.dtype(v): k for k, v in _MODES.items()}


@dataclass
class MrcHeader:
    nx: int = 0
    ny: int = 0
    nz: int = 0
    mode: int = 2
    nxstart: int = 0
    nystart: int = 0
    nzstart: int = 0
    mx: int = 0
    my: int = 0
    mz: int = 0
    cella: tuple = (0.0, 0.0, 0.0)
    cellb: tuple = (90.0, 90.0, 90.0)
    mapc: int = 1
    mapr: int = 2
    maps: int = 3
    dmin: float = 0.0
    dmax: float = 0.0
    dmean: float = 0.0
    ispg: int = 1
    nsymbt: int = 0
    origin: tuple = (0.0, 0.0, 0.0)
    rms: float = 0.0
    extra: bytes = field(default=b"", repr=False)

    @property
    def voxel_size(self):
        """(x, y, z) Angstrom per voxel = cella / m[xyz] (mrcfile semantics)."""
        return tuple(float(c) / m if m else 0.0 for c, m in zip(self.cella, (self.mx, self.my, self.mz)))


def read_mrc(path: str):
    """-> (data ndarray [nz,ny,nx] in file dtype, MrcHeader)."""
    from . import handoff
    handoff.wait_file(path)                  # a file a stage of this process is still writing in the background: join its writer
    with open(path, "rb") as f:
        h = f.read(1024)
        if len(h) < 1024:
            raise ValueError(f"{path}: truncated MRC header")
        # as mrcfile in its default (non-permissive) mode, which the reference uses (create_grids.py:108, preprocessing.py:98): a file
        # without the map ID or with an unknown machine stamp is refused, not parsed as whatever its bytes happen to say
        if h[208:212] != b"MAP ":
            raise ValueError(f"{path}: map ID string 'MAP ' not found: not an MRC file, or the file is corrupt")
        if h[212] == 0x44 and h[213] in (0x44, 0x41):
            end = "<"
        elif h[212] == 0x11 and h[213] == 0x11:
            end = ">"
        else:
            raise ValueError(f"{path}: unrecognised machine stamp 0x{h[212]:02x} 0x{h[213]:02x}")
        ints = struct.unpack(end + "10i", h[0:40])
        cella = struct.unpack(end + "3f", h[40:52])
        cellb = struct.unpack(end + "3f", h[52:64])
        mapc, mapr, maps = struct.unpack(end + "3i", h[64:76])
        dmin, dmax, dmean = struct.unpack(end + "3f", h[76:88])
        ispg, nsymbt = struct.unpack(end + "2i", h[88:96])
        origin = struct.unpack(end + "3f", h[196:208])
        rms = struct.unpack(end + "f", h[216:220])[0]
        hd = MrcHeader(ints[0], ints[1], ints[2], ints[3], ints[4], ints[5], ints[6], ints[7], ints[8], ints[9],
                       cella, cellb, mapc, mapr, maps, dmin, dmax, dmean, ispg, nsymbt, origin, rms)
        if hd.mode not in _MODES:
            raise ValueError(f"{path}: unsupported MRC mode {hd.mode}")
        if min(hd.nx, hd.ny, hd.nz) < 1 or sorted((mapc, mapr, maps)) != [1, 2, 3]:
            raise ValueError(f"{path}: bad MRC header (dims {hd.nx},{hd.ny},{hd.nz}; axes {mapc},{mapr},{maps})")
        hd.extra = f.read(max(nsymbt, 0))
        dt = np.dtype(_MODES[hd.mode]).newbyteorder(end)
        n = hd.nx * hd.ny * hd.nz
        data = np.fromfile(f, dtype=dt, count=n)
        if data.size != n:
            raise ValueError(f"{path}: truncated MRC data")
    return data.reshape(hd.nz, hd.ny, hd.nx).astype(dt.newbyteorder("="), copy=False), hd


def write_mrc(path: str, data: np.ndarray, voxel_size=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), mapc=1, mapr=2, maps=3,
              nxstart=0, nystart=0, nzstart=0, stats=None):
    """Little-endian MRC2014 with header statistics filled in (what mrcfile.new + set_data +
    update_header_stats produce for the fields the path reads).  `stats` = (dmin, dmax, dmean, rms) when the caller already
    has them (the 0/1 encoding channels: one count on the GPU instead of four float64 passes over 67 MB per file)."""
    data = np.ascontiguousarray(data)
    if data.dtype not in _MODE_OF:
        raise ValueError(f"unsupported dtype {data.dtype}")
    nz, ny, nx = data.shape
    h = bytearray(1024)
    struct.pack_into("<10i", h, 0, nx, ny, nz, _MODE_OF[data.dtype], int(nxstart), int(nystart), int(nzstart), nx, ny, nz)
    struct.pack_into("<3f", h, 40, nx * float(voxel_size[0]), ny * float(voxel_size[1]), nz * float(voxel_size[2]))
    struct.pack_into("<3f", h, 52, 90.0, 90.0, 90.0)
    struct.pack_into("<3i", h, 64, int(mapc), int(mapr), int(maps))
    if stats is None:
        d64 = data.astype(np.float64)
        stats = (float(d64.min()), float(d64.max()), float(d64.mean()), float(d64.std()))
        del d64
    struct.pack_into("<3f", h, 76, float(stats[0]), float(stats[1]), float(stats[2]))
    struct.pack_into("<2i", h, 88, 1, 0)
    struct.pack_into("<3f", h, 196, float(origin[0]), float(origin[1]), float(origin[2]))
    h[208:212] = b"MAP "
    h[212:216] = bytes([0x44, 0x44, 0x00, 0x00])
    struct.pack_into("<f", h, 216, float(stats[3]))
    struct.pack_into("<i", h, 220, 0)
    le = np.ascontiguousarray(data.astype(data.dtype.newbyteorder("<"), copy=False))
    # written under a hidden temporary name and renamed into place: a reader that is not one of this package's mirrors (which join
    # the writer, handoff.wait_file) sees either no file or the complete one, never a truncated map
    part = os.path.join(os.path.dirname(path) or ".", f".{os.path.basename(path)}.{os.getpid()}.part")
    try:
        with open(part, "wb", buffering=0) as f:
            f.write(bytes(h))
            body = memoryview(le.reshape(-1)).cast("B")      # the array's own memory: no second copy of a 67-MB volume on its way out
            done = 0
            while done < body.nbytes:
                done += f.write(body[done:])
        os.replace(part, path)
    except BaseException:
        try:
            os.remove(part)
        except OSError:
            pass
        raise


def transpose_to_xyz(data: np.ndarray, hd: MrcHeader):
    """Axis-order transform of GridCreator.transpose (reference utils/create_grids.py:67-87,119-122):
    returns the array indexed (x, y, z) and the offsets [n?start] permuted the same way."""
    axis_order = [hd.maps - 1, hd.mapr - 1, hd.mapc - 1]
    offset = [float(hd.nzstart), float(hd.nystart), float(hd.nxstart)]
    trans_offset, trans_order = [], []
    for i in range(3):
        for j in range(3):
            if axis_order[j] == i:
                trans_offset.append(offset[j])
                trans_order.append(j)
    return np.transpose(data, trans_order), trans_offset
