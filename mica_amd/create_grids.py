"""GridCreator with the reference's interface (reference utils/create_grids.py:25-397): same method
names, arguments, result dicts and .npz tile files, but the windows are cut by the HIP gather kernel
from a volume resident on the GPU, and an in-memory path skips the files altogether."""
from __future__ import annotations

import logging
import os
import time
from glob import glob

import numpy as np
import torch

from . import mrc
from .engine import Engine


class GridCreator:
    def __init__(self, quiet=False, engine: Engine | None = None, device=0):
        self.quiet = quiet
        self.logger = logging.getLogger(__name__)
        self.processed_count = 0
        self.failed_count = 0
        self.failed_entries = []
        self._engine = engine
        self._device = device

    def _eng(self, window: int) -> Engine:
        if self._engine is None or self._engine.tile_size != window:
            self._engine = Engine(self._device, max_batch=1, tile_size=window)
        return self._engine

    def print_clean(self, message):
        if not self.quiet:
            print(message)

    def transpose(self, numpy_image, axis_order, offset):
        """create_grids.py:67-87."""
        trans_offset, trans_order = [], []
        for i in range(3):
            for j in range(len(axis_order)):
                if axis_order[j] == i:
                    trans_offset.append(offset[j])
                    trans_order.append(j)
        return np.transpose(numpy_image, trans_order), trans_offset

    # ---- in-memory form ---------------------------------------------------------------------------
    def load_volume(self, mrc_file):
        """-> (float32 volume indexed (x,y,z), offset list, header) exactly as create_grids.py:108-122."""
        data, hd = mrc.read_mrc(mrc_file)
        vol, offset = mrc.transpose_to_xyz(data, hd)
        return np.ascontiguousarray(vol), offset, hd

    def tiles_on_device(self, volume: np.ndarray | torch.Tensor, grid_size=48, padding=8, chunk=64):
        """Generator of (first, tiles f32[count,1,W,W,W] on the GPU) over the reference's tile order."""
        eng = self._eng(grid_size + 2 * padding)
        v = volume if isinstance(volume, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(volume, dtype=np.float32))
        v = v.to(eng.device, torch.float32).contiguous()
        T = int(eng.lib.mica_tile_count(*v.shape, grid_size))
        for first in range(0, T, chunk):
            count = min(chunk, T - first)
            yield first, eng.gather_tiles(v, grid_size, padding, first, count)

    # ---- file-writing form (same artefacts as the reference) ------------------------------------------
    def create_and_save_grids(self, mrc_file, output_dir, grid_size=48, padding=8, min_grid_max=None):
        """The training-data tilers (scripts_for_training_data/create_grids_for_normalized_map.py:18-101 and its
        four siblings): same pad/loop as the inference tiler but WITHOUT the axis transpose, file prefix
        `grid_`; the normalised-map variant skips tiles whose maximum is below 0.01 (:78) - pass
        min_grid_max=0.01 for that one, None for the mask/encoding variants.  Returns the number of files."""
        return self.create_grids_from_mrc(mrc_file, output_dir, grid_size, padding, "grid", transpose=False,
                                          min_grid_max=min_grid_max)[0]

    def create_grids_from_mrc(self, mrc_file, output_dir, grid_size=48, padding=8, file_prefix="grid", transpose=True,
                              min_grid_max=None):
        """create_grids.py:89-184: returns (grid_count, offset); (0, None) on failure."""
        try:
            os.makedirs(output_dir, exist_ok=True)
            if transpose:
                vol, offset, hd = self.load_volume(mrc_file)
            else:
                data, hd = mrc.read_mrc(mrc_file)
                vol, offset = np.ascontiguousarray(data), [float(hd.nzstart), float(hd.nystart), float(hd.nxstart)]
            orig_shape = vol.shape
            from ._cabi import tile_table
            table = tile_table(*orig_shape, grid_size)
            voxel_size = np.rec.array(hd.voxel_size, dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
            origin = np.rec.array(tuple(hd.origin), dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
            count = 0
            src_dtype = vol.dtype
            for first, tiles in self.tiles_on_device(vol.astype(np.float32), grid_size, padding):
                host = tiles.cpu().numpy()[:, 0]
                for t in range(host.shape[0]):
                    if min_grid_max is not None and host[t].max() < min_grid_max:
                        continue
                    i, j, k, di, dj, dk = (int(x) for x in table[first + t])
                    np.savez(os.path.join(output_dir, f"{file_prefix}_i{i}_j{j}_k{k}.npz"),
                             grid=host[t].astype(src_dtype, copy=False), i=i, j=j, k=k, di=di, dj=dj, dk=dk,
                             orig_shape=orig_shape, grid_size=grid_size, padding=padding, voxel_size=voxel_size,
                             origin=origin, mapc=np.int32(hd.mapc), mapr=np.int32(hd.mapr), maps=np.int32(hd.maps))
                    count += 1
            self.logger.info(f"Created {count} grids from {os.path.basename(mrc_file)}")
            return count, offset
        except Exception as e:
            self.logger.error(f"Grid creation failed for {os.path.basename(mrc_file)}: {e}")
            return 0, None

    def create_normalized_map_grids(self, normalized_map_path, output_dir, grid_size=48, padding=8):
        """create_grids.py:205-267."""
        start = time.time()
        if not os.path.exists(normalized_map_path):
            msg = f"Normalized map not found: {normalized_map_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        n, offset = self.create_grids_from_mrc(normalized_map_path, output_dir, grid_size, padding, "normalized_map_grid")
        return {"success": n > 0, "grid_count": n, "offset": offset, "output_directory": output_dir,
                "processing_time": time.time() - start, "input_file": normalized_map_path}

    def create_AF3_encodings_grids(self, AF3_encodings_path, output_dir, grid_size=48, padding=8, parallel=True):
        """create_grids.py:269-397 (the process pool is gone: one GPU gather per channel)."""
        start = time.time()
        if not os.path.exists(AF3_encodings_path):
            msg = f"AF3 encodings directory not found: {AF3_encodings_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        files = glob(os.path.join(AF3_encodings_path, "*_encoding.mrc"))
        if not files:
            msg = f"No AF3 encoding files found in {AF3_encodings_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        ok = bad = total = 0
        errors = []
        for f in files:
            ch = os.path.basename(f).split("_encoding.mrc")[0]
            n, _ = self.create_grids_from_mrc(f, os.path.join(output_dir, f"{ch}_grids"), grid_size, padding, f"{ch}_grid")
            if n > 0:
                ok += 1
                total += n
            else:
                bad += 1
                errors.append(f"Failed {os.path.basename(f)}")
        return {"success": ok > 0, "successful_channels": ok, "failed_channels": bad, "total_channels": len(files),
                "total_grids": total, "output_directory": output_dir, "processing_time": time.time() - start,
                "processing_errors": errors, "input_directory": AF3_encodings_path}
