"""DataPreprocessor.resample_and_normalize_map with the reference's interface and side effect
(reference utils/preprocessing.py:30-170): reads the MRC, resamples to 1 Angstrom, normalises on the
GPU (exact median / 99.9th-percentile select, libmica_hip.so) and writes
<dirname(AF3_results)>/resampled_normalized_map.mrc.

The cubic-spline resample (scipy.ndimage.zoom(order=3), :117) also runs on the GPU (`mica_zoom_cubic`, f64, bit-exact
against scipy).  `create_AF3_encodings` (:225-347) rasterises the docked model's atoms on the GPU (`mica_rasterise_atoms`)
and writes the 24 `<CH>_encoding.mrc` files; the PDB is read by a fixed-column reader (af3_encoding.py) because Bio.PDB
is not installed - that parser step has no reference to be checked against here."""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from . import handoff, mrc
from .engine import Engine, MicaHipError


_MAP_TYPES = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.uint16): 3}


class DataPreprocessor:
    def __init__(self, map_path, AF3_results, quiet=False, engine: Engine | None = None, device=0, write_files: str | None = None):
        """write_files: "sync" (default; also the environment's MICA_MRC_FILES when the argument is None) - the MRC files this stage
        is called for (the normalised map, the 24 encoding channels) exist when its methods return, as at the reference's call sites
        (utils/modeler.py:675-683); "background" - they are written behind the caller's back (joined by `GridCreator.load_volume`,
        by the predictor mirror before it returns and at process exit): for the Solver flow, whose next stage is the `GridCreator`
        mirror of this process, which takes the volumes from the GPU either way (mica_amd/handoff.py; mica_amd/solver_mirrors.py
        selects it).  Either way a file appears under its name only when it is complete (hidden temporary name + rename)."""
        if write_files is None:
            write_files = os.environ.get("MICA_MRC_FILES", "sync")
        if write_files not in ("sync", "background"):
            raise ValueError("write_files must be 'sync' or 'background'")
        self.write_files = write_files
        self.map_path = map_path
        self.AF3_results = AF3_results
        self.quiet = quiet
        self.normalized_map_path = None
        self.logger = logging.getLogger(__name__)
        self._engine = engine
        self._device = device

    def print_clean(self, message):
        if not self.quiet:
            print(message)

    def normalize_array(self, data: np.ndarray, voxel_size=(1.0, 1.0, 1.0), target_voxel_size=1.0):
        """preprocessing.py:111-133 on an array: returns (float32 map in [0,1], median, percentile).
        Raises MicaHipError where the reference logs 'Normalization failed'."""
        t, med, pct = self.normalize_on_device(data, voxel_size, target_voxel_size)
        return t.cpu().numpy(), med, pct

    def normalize_on_device(self, data: np.ndarray, voxel_size=(1.0, 1.0, 1.0), target_voxel_size=1.0):
        """The same, the result left on the GPU: (float32 device tensor [nz', ny', nx'], median, percentile)."""
        zf = [voxel_size[0] / target_voxel_size, voxel_size[1] / target_voxel_size, voxel_size[2] / target_voxel_size]
        eng = self._engine or Engine(self._device, max_batch=1, tile_size=64)
        self._engine = eng
        # the reference passes the array in the dtype mrcfile returns (:98-117): MRC modes 0/1/6 arrive as integers, stay
        # integers through scipy's zoom and are promoted to float64 by numpy at :124 (include/mica_hip.h, MICA_MAP_*);
        # mode 12 (float16) is refused by scipy.ndimage ("array type dtype('float16') not supported") and the reference fails
        map_type = _MAP_TYPES.get(np.dtype(data.dtype))
        if map_type is None:
            raise MicaHipError(f"array type {np.dtype(data.dtype)!r} not supported")
        t = torch.from_numpy(np.ascontiguousarray(data).astype(np.float32, copy=False)).to(eng.device)
        # the reference zooms unconditionally (:117); the kernel reproduces scipy bit for bit, including factor 1.0
        # (identity on finite data, NaN spreading through the recursive prefilter otherwise)
        t = eng.zoom_cubic(t, zf, map_type)
        med, pct = eng.normalise_map_(t, map_type)
        return t, med, pct

    def resample_and_normalize_map(self, target_voxel_size=1.0):
        """Side effect and messages as preprocessing.py:80-170; returns None."""
        success = False
        try:
            data, hd = mrc.read_mrc(self.map_path)
            t, _, _ = self.normalize_on_device(np.asarray(data), hd.voxel_size, target_voxel_size)
            self.normalized_map_path = path = os.path.join(os.path.dirname(self.AF3_results), 'resampled_normalized_map.mrc')
            nz, ny, nx = (int(v) for v in t.shape)
            tv = float(target_voxel_size)
            # the header the written file will carry (mrc.write_mrc below), for the stage that takes the map from the GPU instead
            hdn = mrc.MrcHeader(nx=nx, ny=ny, nz=nz, mode=2, nxstart=hd.nxstart, nystart=hd.nystart, nzstart=hd.nzstart, mx=nx, my=ny, mz=nz,
                                cella=(float(np.float32(nx * tv)), float(np.float32(ny * tv)), float(np.float32(nz * tv))),
                                mapc=hd.mapc, mapr=hd.mapr, maps=hd.maps, origin=tuple(float(np.float32(v)) for v in hd.origin))

            # the header statistics (mrcfile's update_header_stats: min, max, mean, population standard deviation) as float64
            # reductions on the GPU: exact min / max, mean and deviation accurate to ~1e-15 before the header rounds them to float32
            # - the float64 passes over the host copy were most of this file's write time (0.56 s of 0.65 s at 512^3)
            d64 = t.double()
            stats = (float(d64.min()), float(d64.max()), float(d64.mean()), float(d64.std(unbiased=False)))
            del d64

            def write():
                mrc.write_mrc(path, t.cpu().numpy(), voxel_size=(target_voxel_size,) * 3, origin=hd.origin, mapc=hd.mapc, mapr=hd.mapr,
                              maps=hd.maps, nxstart=hd.nxstart, nystart=hd.nystart, nzstart=hd.nzstart, stats=stats)
            # the normalised map stays on the GPU for GridCreator (mica_amd/handoff.py); the file the reference's call site expects
            # (utils/modeler.py:684-690) is written behind the caller's back and joined by whoever reads it
            handoff.drop_file(path)                # a writer of an earlier run still on its way to this very path is joined first
            if os.path.exists(path):
                os.remove(path)
            handoff.register_file(path, t, hdn, writer=write)
            if self.write_files == "sync":
                handoff.wait_file(path)
            success = True
        except Exception as e:
            self.logger.error(f"Map processing failed: {e}")
        self.print_clean("Map successfully resampled and normalized." if success else "Map Resampling and Normalization Failed")

    def encode_AF3_volume(self, combined_docked_model_path, origin, shape) -> torch.Tensor:
        """The atom loop of create_AF3_encodings (:268-298) without the disk: float32 [24, nz, ny, nx] on the device."""
        from . import af3_encoding

        eng = self._engine or Engine(self._device, max_batch=1, tile_size=64)
        self._engine = eng
        coords, names, resnames = af3_encoding.read_pdb_atoms(combined_docked_model_path)
        return af3_encoding.rasterise(eng, coords, names, resnames, origin, shape)

    def create_AF3_encodings(self, combined_docked_model_path):
        """Side effects and return value as preprocessing.py:225-347: reads the normalised map's header, writes
        <dirname(AF3_results)>/AF3_encodings/<CH>_encoding.mrc for the 24 channels, returns True on success."""
        from . import af3_encoding

        success = False
        try:
            fe = handoff.lookup_file(self.normalized_map_path)
            if fe is not None:
                hd, shape = fe.header, tuple(fe.tensor.shape)
            else:
                handoff.wait_file(self.normalized_map_path)
                data, hd = mrc.read_mrc(self.normalized_map_path)
                shape = data.shape
            vol = self.encode_AF3_volume(combined_docked_model_path, hd.origin, shape)          # f32 [24, nz, ny, nx] on the device
            u8 = vol.to(torch.uint8)                                                            # 0 / 1 by construction (:288-298)
            del vol
            self.AF3_encodings = os.path.join(os.path.dirname(self.AF3_results), 'AF3_encodings')
            os.makedirs(self.AF3_encodings, exist_ok=True)
            nz, ny, nx = shape
            hdn = mrc.MrcHeader(nx=nx, ny=ny, nz=nz, mode=2, nxstart=hd.nxstart, nystart=hd.nystart, nzstart=hd.nzstart, mx=nx, my=ny, mz=nz,
                                cella=(float(nx), float(ny), float(nz)), mapc=hd.mapc, mapr=hd.mapr, maps=hd.maps,
                                origin=tuple(float(np.float32(v)) for v in hd.origin))
            # header statistics of a 0/1 volume from one count per channel: min / max / mean are exactly numpy's float64 results
            # (an integer sum), the standard deviation sqrt(p (1 - p)) equals numpy's two-pass value after the header's float32 rounding
            ones = u8.sum(dim=(1, 2, 3), dtype=torch.int64).cpu().tolist()
            nvox = float(nz * ny * nx)

            def stats01(k):
                p1 = k / nvox
                return (0.0 if k < nvox else 1.0, 1.0 if k > 0 else 0.0, p1, float(np.sqrt(p1 * (1.0 - p1))))
            # the 24 channel files are written in the background, each channel staying resident (uint8) for GridCreator
            for ch, name in enumerate(af3_encoding.CHANNEL_NAMES):
                p = os.path.join(self.AF3_encodings, f"{name}_encoding.mrc")
                handoff.drop_file(p)               # joins a writer of an earlier run (a second map in the same AF3_results directory)
                if os.path.exists(p):
                    os.remove(p)

                def write(p=p, ch=ch):
                    mrc.write_mrc(p, u8[ch].cpu().numpy().astype(np.float32), voxel_size=(1.0, 1.0, 1.0), origin=hd.origin, mapc=hd.mapc,
                                  mapr=hd.mapr, maps=hd.maps, nxstart=hd.nxstart, nystart=hd.nystart, nzstart=hd.nzstart, stats=stats01(ones[ch]))
                handoff.register_file(p, u8[ch], hdn, writer=write)
            if self.write_files == "sync":                    # the 24 writers run on the file pool's threads; all joined here
                for name in af3_encoding.CHANNEL_NAMES:
                    handoff.wait_file(os.path.join(self.AF3_encodings, f"{name}_encoding.mrc"))
            success = True
        except Exception as e:
            self.print_clean(f"   Encoding failed: AF3 encoding failed: {e}")
        return success
