"""Disk-free tile -> forward -> stitch pipeline on one GPU (and its sharded multi-GPU form).

Replaces the npz round trips between GridCreator, CryoEMTestDataset, CryoEMPredictor.run_inference and
reconstruct_volume (reference utils/create_grids.py:143-176, dataset/dataset.py:194-224,
utils/predict.py:334-376,439-512) with device-resident tensors; the four output volumes are the
ones run_prediction() returns (utils/predict.py:589-634).
"""
from __future__ import annotations

import torch

from . import _cabi
from .engine import AF_PER_TILE, Engine

KEYS = ("backbone_probability", "carbon_alpha_probability", "amino_acid_prediction", "amino_acid_probability")


def volume_dict(full):
    """The four volumes of utils/predict.py:459-462 as views of one [23, N0, N1, N2] array / tensor (record channel order)."""
    return {"backbone_probability": full[0], "carbon_alpha_probability": full[1], "amino_acid_prediction": full[2],
            "amino_acid_probability": full[3:]}


class SlabDownloader:
    """Brings the stitched volumes to the host WHILE the map is still being computed.

    Tiles are processed in the reference's order (utils/create_grids.py:143-145: x outermost), so once the tiles of an x block
    have been stitched the slab out[:, i:i+grid] is final.  It is copied into one of two pinned staging buffers on a copy stream,
    behind an event of the stitching stream, and a helper thread unpacks the staging buffer into the result array (ordinary
    pageable memory: page-locking a whole 12.3-GB result costs more than the first slab takes to compute, and the first download
    would wait for it) while the GPU works on the next slabs.  What is left when the last tile is done is the last slab only - at
    512^3, 1.1 of 12.3 GB."""

    def __init__(self, out: torch.Tensor, grid: int, tiles_per_slab: int):
        import queue
        import threading

        import numpy as np
        self.out, self.grid, self.per = out, grid, tiles_per_slab
        self.C, self.n0 = out.shape[0], out.shape[1]
        self.nslabs = -(-self.n0 // grid)
        self.stream = torch.cuda.Stream(device=out.device)
        self.next = 0
        self.host = np.empty(tuple(out.shape), dtype=np.float32)        # the result: views of it are what the caller gets
        self.stage = [None, None]
        self.free = [threading.Event(), threading.Event()]               # staging buffer k has been unpacked
        self.jobs = queue.Queue()
        self.error = None

        def alloc():
            torch.cuda.set_device(out.device)        # page-locking registers with the CURRENT device's context: not device 0's on rank r
            for k in range(2):
                self.stage[k] = torch.empty((self.C, min(grid, self.n0), *out.shape[2:]), dtype=out.dtype).pin_memory()
                self.free[k].set()

        def unpack():
            from concurrent.futures import ThreadPoolExecutor
            torch.cuda.set_device(out.device)
            with ThreadPoolExecutor(max_workers=4, thread_name_prefix="mica-slab-copy") as pool:      # numpy's copy releases the GIL
                def piece(k, lo, hi, c0, c1):
                    self.host[c0:c1, lo:hi] = self.stage[k][c0:c1, :hi - lo].numpy()
                while True:
                    job = self.jobs.get()
                    if job is None:
                        return
                    k, lo, hi, ev = job
                    try:
                        ev.synchronize()
                        step = -(-self.C // 4)
                        for f in [pool.submit(piece, k, lo, hi, c0, min(c0 + step, self.C)) for c0 in range(0, self.C, step)]:
                            f.result()
                    except Exception as ex:      # reported by finish()
                        self.error = ex
                    finally:
                        self.free[k].set()
        # daemon threads: a caller that fails half way through a map (and never reaches finish()) must not leave a thread behind that
        # keeps the interpreter from exiting; close() ends them in the orderly case
        self._alloc = threading.Thread(target=alloc, name="mica-pinned-alloc", daemon=True)
        self._alloc.start()
        self._unpack = threading.Thread(target=unpack, name="mica-slab-unpack", daemon=True)
        self._unpack.start()
        self._closed = False

    def _download(self, s: int):
        k = s & 1
        if self.stage[k] is None:
            self._alloc.join()
        self.free[k].wait()                      # slab s - 2 has left this staging buffer (long ago: a slab computes for far longer)
        self.free[k].clear()
        lo, hi = s * self.grid, min((s + 1) * self.grid, self.n0)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.out.device))
        done = torch.cuda.Event()
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            for c in range(self.C):              # each channel's slab is one contiguous run of the volume
                self.stage[k][c, :hi - lo].copy_(self.out[c, lo:hi], non_blocking=True)
            done.record(self.stream)
        self.jobs.put((k, lo, hi, done))

    def tiles_done(self, n_tiles: int):
        """`n_tiles` tiles (in table order) have been stitched on the current stream: download every slab they complete."""
        complete = min(n_tiles // self.per, self.nslabs)
        while self.next < complete:
            self._download(self.next)
            self.next += 1

    def close(self):
        """End the helper threads (idempotent); what is still queued is unpacked first."""
        if not self._closed:
            self._closed = True
            self.jobs.put(None)
            self._unpack.join()
            self._alloc.join()

    def finish(self):
        """-> the host array [23, N0, N1, N2] (numpy), complete."""
        try:
            self.tiles_done(self.per * self.nslabs)
        finally:
            self.close()
        if self.error is not None:
            raise self.error
        return self.host


class VolumePredictor:
    def __init__(self, engine: Engine, grid_size: int = 48, padding: int = 8, batch: int | None = None):
        if grid_size + 2 * padding != engine.tile_size:
            raise _cabi.MicaHipError(f"grid_size + 2*padding must equal the engine's tile size {engine.tile_size}")
        self.e, self.grid, self.pad = engine, grid_size, padding
        self.batch = batch or engine.max_batch
        S = engine.tile_size
        dev = engine.device
        B = self.batch
        self._map_tiles = torch.empty((B, 1, S, S, S), dtype=torch.float32, device=dev)
        self._af_tiles = torch.empty((B, 24, S, S, S), dtype=torch.float32, device=dev)
        # one record per tile: [bb, ca, aa_pred, aa_prob x20] = 23 channels
        self._rec = torch.empty((B, 23, S, S, S), dtype=torch.float32, device=dev)

    def run_batch(self, vol, af_vol, first: int, count: int):
        """tiles first..first+count-1 -> record tensor view [count,23,S,S,S] (valid until the next call)."""
        e = self.e
        mt = e.gather_tiles(vol, self.grid, self.pad, first, count, out=self._map_tiles[:count])
        at = None
        if af_vol is not None:
            at = e.gather_tiles(af_vol, self.grid, self.pad, first, count, out=self._af_tiles[:count])
        S = e.tile_size
        return e.forward_records(mt.view(count, S, S, S), at, self._rec[:count], af_mode=AF_PER_TILE)

    def predict_volume(self, vol: torch.Tensor, af_vol: torch.Tensor | None = None, to_host: bool = False):
        """vol f32[N0,N1,N2] on the GPU (already normalised, (x,y,z) order), af_vol f32 or uint8 [24,N0,N1,N2] or None.
        Returns the dict of four device volumes with the shapes/dtypes of utils/predict.py:459-462 - or, with `to_host`, the same
        dict as numpy arrays (views of one host array), downloaded slab by slab while later tiles compute (SlabDownloader)."""
        e = self.e
        n0, n1, n2 = vol.shape
        T = int(e.lib.mica_tile_count(n0, n1, n2, self.grid))
        out = torch.zeros((23, n0, n1, n2), dtype=torch.float32, device=e.device)
        dl = SlabDownloader(out, self.grid, (-(-n1 // self.grid)) * (-(-n2 // self.grid))) if to_host else None
        try:
            for first in range(0, T, self.batch):
                count = min(self.batch, T - first)
                rec = self.run_batch(vol, af_vol, first, count)
                e.stitch_tiles(rec, out, self.grid, self.pad, first)
                if dl is not None:
                    dl.tiles_done(first + count)
            if dl is not None:
                return volume_dict(dl.finish())
            return volume_dict(out)
        finally:
            if dl is not None:
                dl.close()

    def predict_maps_streamed(self, maps, afs=None):
        """Several independent maps back to back (BASELINE.json configs[4]) with host buffers on both sides.
        Map i+1 is staged (host copy into pinned memory on a helper thread, then hipMemcpyAsync on a copy stream) while
        map i computes; the four volumes of map i travel back through a pinned buffer on the copy stream while map i+1
        computes.  `maps`: list of float32 [N0,N1,N2] numpy arrays, `afs`: optional list of float32 [24,N0,N1,N2] (or
        None entries); returns a list of dicts of host arrays."""
        import threading

        import numpy as np
        e = self.e
        dev = e.device
        copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        afs = afs or [None] * len(maps)
        slots = [None, None]          # per parity: (pinned map, pinned af, device map, device af, ready event)

        def stage(i):
            torch.cuda.set_device(dev)
            m = np.ascontiguousarray(maps[i], dtype=np.float32)
            a = None if afs[i] is None else np.ascontiguousarray(afs[i], dtype=np.float32)
            k = i & 1
            prev = slots[k]
            pm = prev[0] if prev is not None and prev[0].shape == m.shape else torch.empty(m.shape, dtype=torch.float32, pin_memory=True)
            pm.copy_(torch.from_numpy(m))
            pa = None
            if a is not None:
                pa = prev[1] if prev is not None and prev[1] is not None and prev[1].shape == a.shape else \
                    torch.empty(a.shape, dtype=torch.float32, pin_memory=True)
                pa.copy_(torch.from_numpy(a))
            with torch.cuda.stream(copy_stream):
                dm = pm.to(dev, non_blocking=True)
                da = None if pa is None else pa.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            slots[k] = (pm, pa, dm, da, ev)

        def collect(pending):
            """finish the download of a previous map: (pinned [23,...] buffer, event) -> dict of host arrays"""
            buf, ev = pending
            ev.synchronize()
            h = buf.numpy()
            return {"backbone_probability": h[0].copy(), "carbon_alpha_probability": h[1].copy(),
                    "amino_acid_prediction": h[2].copy(), "amino_acid_probability": h[3:].copy()}

        results, pending = [None] * len(maps), None
        hbuf = [None, None]           # pinned download buffers, by parity

        def helper(i, pend):
            """runs beside map i's compute: stage map i+1, unpack map i-1"""
            if i + 1 < len(maps):
                stage(i + 1)
            if pend is not None:
                results[i - 1] = collect(pend)

        if maps:
            stage(0)
        for i in range(len(maps)):
            pm, pa, dm, da, ev = slots[i & 1]
            # the other slot's device tensors were consumed by map i-1, whose kernels are already enqueued on `main`;
            # the copy stream waits for them before the buffers are overwritten
            copy_stream.wait_stream(main)
            th = threading.Thread(target=helper, args=(i, pending))
            th.start()
            main.wait_event(ev)
            dm.record_stream(main)
            if da is not None:
                da.record_stream(main)
            out = self.predict_volume(dm, da)                      # dict of views of one [23, N0, N1, N2] tensor
            full = out["backbone_probability"]._base
            done = torch.cuda.Event()
            done.record(main)
            th.join()                                              # map i-1 is unpacked: its pinned buffer (other parity) is free
            k = i & 1
            if hbuf[k] is None or hbuf[k].shape != full.shape:
                hbuf[k] = torch.empty(full.shape, dtype=torch.float32, pin_memory=True)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                hbuf[k].copy_(full, non_blocking=True)
                full.record_stream(copy_stream)
                dl = torch.cuda.Event()
                dl.record(copy_stream)
            pending = (hbuf[k], dl)
        if pending is not None:
            results[len(maps) - 1] = collect(pending)
        return results

    def predict_volume_sharded(self, vol: torch.Tensor, af_vol: torch.Tensor | None = None, group=None,
                               force_collective: bool = False, stats: dict | None = None, to_host: bool = False,
                               gather_to_root: bool = False):
        """Multi-GPU form: every rank holds the (normalised) volume (the encodings as uint8: a quarter of the bytes on every rank),
        runs its share of the tile batches and the cropped records are all-gathered (RCCL) - or, with `gather_to_root`, gathered into
        the stitching rank alone; rank 0 returns the dict, the other ranks return None.
        force_collective: run the collective even in a process group of one rank (executes the RCCL branch on one GPU);
        stats (optional dict) receives the number of collectives issued, the backend and the world size;
        to_host: rank 0 returns numpy arrays, downloaded slab by slab while later rounds compute (SlabDownloader)."""
        import torch.distributed as dist
        from .dist import sharded_records
        e = self.e
        n0, n1, n2 = vol.shape
        T = int(e.lib.mica_tile_count(n0, n1, n2, self.grid))
        g, p = self.grid, self.pad
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        out = torch.zeros((23, n0, n1, n2), dtype=torch.float32, device=e.device) if rank == 0 else None

        def run(first, count):
            return self.run_batch(vol, af_vol, first, count)[:, :, p:p + g, p:p + g, p:p + g]

        dl = SlabDownloader(out, g, (-(-n1 // g)) * (-(-n2 // g))) if to_host and rank == 0 else None
        done = [0]

        def stitch(rec, first):
            e.stitch_tiles(rec.contiguous(), out, g, 0, first)      # cropped records: window = grid, no halo
            if dl is not None:
                # batches reach the stitcher in tile order (round by round, rank by rank): everything before first + count is final
                done[0] = max(done[0], first + int(rec.shape[0]))
                dl.tiles_done(done[0])

        try:
            sharded_records(run, stitch, T, self.batch, (23, g, g, g), e.device, group=group, stitch_rank=0,
                            force_collective=force_collective, stats=stats, gather_to_root=gather_to_root)
            if rank != 0:
                return None
            if dl is not None:
                return volume_dict(dl.finish())
            return volume_dict(out)
        finally:
            if dl is not None:
                dl.close()
