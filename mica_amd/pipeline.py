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

    def predict_volume(self, vol: torch.Tensor, af_vol: torch.Tensor | None = None):
        """vol f32[N0,N1,N2] on the GPU (already normalised, (x,y,z) order), af_vol f32[24,N0,N1,N2] or None.
        Returns the dict of four device volumes with the shapes/dtypes of utils/predict.py:459-462."""
        e = self.e
        n0, n1, n2 = vol.shape
        T = int(e.lib.mica_tile_count(n0, n1, n2, self.grid))
        out = torch.zeros((23, n0, n1, n2), dtype=torch.float32, device=e.device)
        for first in range(0, T, self.batch):
            count = min(self.batch, T - first)
            rec = self.run_batch(vol, af_vol, first, count)
            e.stitch_tiles(rec, out, self.grid, self.pad, first)
        return {"backbone_probability": out[0], "carbon_alpha_probability": out[1],
                "amino_acid_prediction": out[2], "amino_acid_probability": out[3:]}

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
                               force_collective: bool = False, stats: dict | None = None):
        """Multi-GPU form: every rank holds the (normalised) volume, runs its share of the tile batches and
        the cropped records are all-gathered (RCCL); rank 0 returns the dict, the other ranks return None.
        force_collective: run the all-gather even in a process group of one rank (executes the RCCL branch on one GPU);
        stats (optional dict) receives the number of collectives issued, the backend and the world size."""
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

        def stitch(rec, first):
            e.stitch_tiles(rec.contiguous(), out, g, 0, first)      # cropped records: window = grid, no halo

        sharded_records(run, stitch, T, self.batch, (23, g, g, g), e.device, group=group, stitch_rank=0,
                        force_collective=force_collective, stats=stats)
        if rank != 0:
            return None
        return {"backbone_probability": out[0], "carbon_alpha_probability": out[1],
                "amino_acid_prediction": out[2], "amino_acid_probability": out[3:]}
