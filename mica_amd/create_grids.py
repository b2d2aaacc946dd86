"""GridCreator with the reference's interface (reference utils/create_grids.py:25-397): same method
names, arguments, result dicts and .npz tile files, but the windows are cut by the HIP gather kernel
from a volume resident on the GPU, and the volume STAYS resident: it is registered under the directory
the caller asked to fill (mica_amd/handoff.py), where a `CryoEMPredictor(grids_path=...)` of the same
process finds it and predicts without reading a single tile file.  The files themselves - the reference's
on-disk wire format, what a predictor in ANOTHER process reads - are complete when a wrapper returns
(`write_files="sync"`, the default: the reference's contract, utils/create_grids.py:159-174), written by a
background pool behind the caller (`"background"`; mica_amd/solver_mirrors.py, the Solver-flow import shim
of INTEGRATION.md, selects it) or not at all (`False`)."""
from __future__ import annotations

import logging
import os
import time
from concurrent.futures import ThreadPoolExecutor
from glob import glob

import numpy as np
import torch

from . import handoff, mrc
from .dataset import AF3_TYPES
from .engine import Engine


def _axis_perm(hd):
    """The permutation GridCreator.transpose applies to [section, row, column] data, and the permuted start offsets
    (create_grids.py:67-87, 119-122)."""
    axis_order = [hd.maps - 1, hd.mapr - 1, hd.mapc - 1]
    offset = [float(hd.nzstart), float(hd.nystart), float(hd.nxstart)]
    order = [j for i in range(3) for j in range(3) if axis_order[j] == i]
    return order, [offset[j] for j in order]


class GridCreator:
    def __init__(self, quiet=False, engine: Engine | None = None, device=0, write_files: bool | str | None = None):
        """write_files: produce the reference's `.npz` tile files (create_grids.py:159-174).
          "sync"         (default; also the environment's MICA_TILE_FILES when the argument is None) every file is complete when a
                         wrapper returns, as at the reference's call sites (utils/modeler.py:684-706): any consumer - the reference's
                         own CryoEMPredictor / CryoEMTestDataset behind a swapped GridCreator, a subprocess, an os.listdir - finds them;
          "background"   (or True) the wrappers `create_normalized_map_grids` / `create_AF3_encodings_grids` return as soon as the
                         volume is resident and registered; the files follow from a background writer (`wait_for_files()` joins it;
                         so do process exit and the `CryoEMPredictor` mirror of this process, whichever route it takes).  For the
                         Solver flow (getData -> nnPred -> rmtree), where nothing but the mirrors looks at the directory in between:
                         mica_amd/solver_mirrors.py.  A file appears under its final name only when it is complete (written under a
                         hidden name and renamed), so a foreign reader can find FEWER files than it expects, never a truncated one;
          False / "none" no files - for a caller that knows its predictor is the mirror in this process."""
        self.quiet = quiet
        self.logger = logging.getLogger(__name__)
        self.processed_count = 0
        self.failed_count = 0
        self.failed_entries = []
        self._engine = engine
        self._device = device
        if write_files is None:
            write_files = os.environ.get("MICA_TILE_FILES", "sync")
        if write_files is True:
            write_files = "background"
        if write_files in (False, "none", "false", "0"):
            write_files = "none"
        if write_files not in ("sync", "background", "none"):
            raise ValueError("write_files must be 'sync', 'background' (True) or 'none' (False)")
        self.write_files = write_files != "none"
        self.sync_files = write_files == "sync"
        self._writers = []

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

    def wait_for_files(self):
        """Join the background tile-file writers this creator started; returns the number of files written."""
        n = 0
        for w in self._writers:
            n += w.wait()
        self._writers = []
        return n

    # ---- in-memory form ---------------------------------------------------------------------------
    def load_volume(self, mrc_file):
        """-> (float32 volume indexed (x,y,z), offset list, header) exactly as create_grids.py:108-122."""
        handoff.wait_file(mrc_file)
        data, hd = mrc.read_mrc(mrc_file)
        vol, offset = mrc.transpose_to_xyz(data, hd)
        return np.ascontiguousarray(vol), offset, hd

    def _device_volume(self, mrc_file, device, transpose=True, as_u8=False):
        """The MRC's data on the GPU, indexed as the tiles are: -> (tensor f32 [N0,N1,N2] - or uint8 when `as_u8` and every value
        is a small non-negative integer, as binary encodings are -, offset, header, file dtype).  Taken from the stage that wrote
        the file if it is still resident (handoff.lookup_file), read from disk otherwise."""
        fe = handoff.lookup_file(mrc_file)
        if fe is not None:
            t, hd = fe.tensor, fe.header
            file_dtype = np.dtype(np.float32)
            if as_u8 and t.dtype != torch.uint8:
                u = t.to(torch.uint8)
                t = u if bool((u.to(torch.float32) == t).all()) else t
            elif not as_u8 and t.dtype != torch.float32:
                t = t.to(torch.float32)
        else:
            handoff.wait_file(mrc_file)
            data, hd = mrc.read_mrc(mrc_file)
            file_dtype = data.dtype
            host = None
            if as_u8 and data.dtype == np.float32:
                u = data.astype(np.uint8)
                if np.array_equal(u, data):
                    host = u
            if host is None:
                host = np.ascontiguousarray(data).astype(np.float32, copy=False)
            t = torch.from_numpy(host).to(device)
        if transpose:
            order, offset = _axis_perm(hd)
            t = t.permute(*order)
        else:
            offset = [float(hd.nzstart), float(hd.nystart), float(hd.nxstart)]
        return t.contiguous(), offset, hd, file_dtype

    def tiles_on_device(self, volume: np.ndarray | torch.Tensor, grid_size=48, padding=8, chunk=64):
        """Generator of (first, tiles f32[count,1,W,W,W] on the GPU) over the reference's tile order."""
        eng = self._eng(grid_size + 2 * padding)
        v = volume if isinstance(volume, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(volume, dtype=np.float32))
        v = v.to(eng.device, torch.float32).contiguous()
        T = int(eng.lib.mica_tile_count(*v.shape, grid_size))
        for first in range(0, T, chunk):
            count = min(chunk, T - first)
            yield first, eng.gather_tiles(v, grid_size, padding, first, count)

    @staticmethod
    def _constant_members(orig_shape, grid_size, padding, hd):
        """The per-map members of a tile file as np.savez stores them (create_grids.py:163-174)."""
        voxel_size = np.rec.array(hd.voxel_size, dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
        origin = np.rec.array(tuple(hd.origin), dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
        return dict(orig_shape=tuple(int(v) for v in orig_shape), grid_size=grid_size, padding=padding, voxel_size=voxel_size, origin=origin,
                    mapc=np.int32(hd.mapc), mapr=np.int32(hd.mapr), maps=np.int32(hd.maps))

    def _start_writer(self, volume4, table, targets, grid_size, padding, hd, file_dtype, min_grid_max=None):
        eng = self._eng(grid_size + 2 * padding)
        # a long-lived creator (one instance tiling many maps) does not keep every finished writer: failed ones stay for wait_for_files
        self._writers = [w for w in self._writers if not (w.done() and w.error is None)]
        w = handoff.TileFileWriter(eng, volume4, table, targets, grid_size, padding,
                                   self._constant_members(volume4.shape[1:], grid_size, padding, hd), file_dtype=file_dtype,
                                   min_grid_max=min_grid_max).start()
        self._writers.append(w)
        return w

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
        """create_grids.py:89-184: returns (grid_count, offset); (0, None) on failure.  Synchronous like the reference: the files
        exist when it returns (this entry always writes them - they are what it is called for)."""
        try:
            os.makedirs(output_dir, exist_ok=True)
            eng = self._eng(grid_size + 2 * padding)
            vol, offset, hd, file_dtype = self._device_volume(mrc_file, eng.device, transpose=transpose)
            from ._cabi import tile_table
            table = tile_table(*vol.shape, grid_size)
            w = self._start_writer(vol[None], table, [(output_dir, file_prefix)], grid_size, padding, hd, file_dtype, min_grid_max)
            count = w.wait()
            self._writers.remove(w)
            self.logger.info(f"Created {count} grids from {os.path.basename(mrc_file)}")
            return count, offset
        except Exception as e:
            self.logger.error(f"Grid creation failed for {os.path.basename(mrc_file)}: {e}")
            return 0, None

    def create_normalized_map_grids(self, normalized_map_path, output_dir, grid_size=48, padding=8):
        """create_grids.py:205-267.  The transposed map stays on the GPU, registered under `output_dir`; the tile files follow in
        the background when `write_files`."""
        start = time.time()
        if not os.path.exists(normalized_map_path) and handoff.lookup_file(normalized_map_path) is None:
            msg = f"Normalized map not found: {normalized_map_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        try:
            handoff.drop_grids(output_dir, cancel_files=True)       # an earlier volume registered for this directory, and its writer
            os.makedirs(output_dir, exist_ok=True)
            eng = self._eng(grid_size + 2 * padding)
            vol, offset, hd, file_dtype = self._device_volume(normalized_map_path, eng.device)
            from ._cabi import tile_table
            table = tile_table(*vol.shape, grid_size)
            n = len(table)
            writer = None
            if self.write_files:
                writer = self._start_writer(vol[None], table, [(output_dir, "normalized_map_grid")], grid_size, padding, hd, file_dtype)
            handoff.register_grids(output_dir, handoff.GridEntry("map", vol, grid_size, padding, offset=offset, writer=writer,
                                                                 sources=(normalized_map_path,)))
            if self.sync_files:
                self.wait_for_files()
            self.logger.info(f"Created {n} grids from {os.path.basename(normalized_map_path)}")
        except Exception as e:
            self.logger.error(f"Grid creation failed for {os.path.basename(normalized_map_path)}: {e}")
            n, offset = 0, None
        return {"success": n > 0, "grid_count": n, "offset": offset, "output_directory": output_dir,
                "processing_time": time.time() - start, "input_file": normalized_map_path}

    def create_AF3_encodings_grids(self, AF3_encodings_path, output_dir, grid_size=48, padding=8, parallel=True):
        """create_grids.py:269-397 (the process pool is gone: the channel files are read on a few threads, the 24 channels become
        ONE uint8 volume on the GPU - binary encodings, a quarter of the float32 bytes - registered under `output_dir`, and one
        background writer cuts all channels' tile files)."""
        start = time.time()
        if not os.path.exists(AF3_encodings_path):
            msg = f"AF3 encodings directory not found: {AF3_encodings_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        files = glob(os.path.join(AF3_encodings_path, "*_encoding.mrc"))
        # channel files this process is still writing (DataPreprocessor.create_AF3_encodings) count as present
        seen = {os.path.realpath(f) for f in files}
        files += [f for f in handoff.files_under(AF3_encodings_path, "_encoding.mrc") if f not in seen]
        if not files:
            msg = f"No AF3 encoding files found in {AF3_encodings_path}"
            self.logger.error(msg)
            return {"success": False, "error": msg}
        handoff.drop_grids(output_dir, cancel_files=True)           # an earlier volume registered for this directory, and its writer
        names = [os.path.basename(f).split("_encoding.mrc")[0] for f in files]
        rank = {n: i for i, n in enumerate(AF3_TYPES)}
        order = sorted(range(len(files)), key=lambda q: (rank.get(names[q], len(rank)), names[q]))     # the 24 known channels first
        eng = self._eng(grid_size + 2 * padding)

        def load(q):
            try:
                return self._device_volume(files[q], eng.device, as_u8=True)
            except Exception as e:
                self.logger.error(f"Grid creation failed for {os.path.basename(files[q])}: {e}")
                return None

        with ThreadPoolExecutor(max_workers=min(8, len(files)) if parallel else 1) as pool:
            loaded = list(pool.map(load, order))
        ok = [(names[q], r) for q, r in zip(order, loaded) if r is not None]
        errors = [f"Failed {os.path.basename(files[q])}" for q, r in zip(order, loaded) if r is None]
        bad = len(errors)
        total = 0
        if ok:
            from ._cabi import tile_table
            shapes = {tuple(r[0].shape) for _, r in ok}
            groups = [ok] if len(shapes) == 1 else [[it] for it in ok]            # channel files of different shapes: one by one
            for grp in groups:
                u8 = all(r[0].dtype == torch.uint8 for _, r in grp)                  # binary encodings: a quarter of the float32 bytes
                vol = torch.stack([r[0] if u8 else r[0].to(torch.float32) for _, r in grp])
                hd, file_dtype = grp[0][1][2], grp[0][1][3]
                table = tile_table(*vol.shape[1:], grid_size)
                total += len(table) * len(grp)
                writer = None
                if self.write_files:
                    writer = self._start_writer(vol, table, [(os.path.join(output_dir, f"{n}_grids"), f"{n}_grid") for n, _ in grp],
                                                grid_size, padding, hd, file_dtype)
                if len(groups) == 1:
                    chans = tuple(n for n, _ in grp)
                    # the predictor needs exactly the 24 channels of dataset.py:184-188 in that order; anything else in the directory
                    # is tiled to files like the reference does, but is not part of the resident hand-off
                    if chans[:len(AF3_TYPES)] == tuple(AF3_TYPES):
                        os.makedirs(output_dir, exist_ok=True)
                        handoff.register_grids(output_dir, handoff.GridEntry("af3", vol[:len(AF3_TYPES)], grid_size, padding,
                                                                             channels=chans[:len(AF3_TYPES)], writer=writer,
                                                                             sources=tuple(files[q] for q, r in zip(order, loaded) if r is not None)))
        if self.sync_files:
            self.wait_for_files()
        return {"success": len(ok) > 0, "successful_channels": len(ok), "failed_channels": bad, "total_channels": len(files),
                "total_grids": total, "output_directory": output_dir, "processing_time": time.time() - start,
                "processing_errors": errors, "input_directory": AF3_encodings_path}
