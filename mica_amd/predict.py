"""CryoEMPredictor with the reference's outer boundary (reference utils/predict.py:40-634): same
constructor, `run_prediction() -> (success, volumes)` with the four arrays Solver.nnPred consumes
(utils/modeler.py:735-738), same "(False, {}) + log" failure convention - but tiles go
file -> GPU -> network -> softmax/argmax -> stitch without the per-tile .npz spill, and nothing runs
on the CPU except file reading."""
from __future__ import annotations

import glob
import logging
import os
import time

import numpy as np
import torch

from . import handoff
from .dataset import CryoEMTestDataset
from .engine import AF_BATCH, AF_PER_TILE, Engine
from .weights import load_checkpoint_state_dict

KEYS = ("backbone_probability", "carbon_alpha_probability", "amino_acid_prediction", "amino_acid_probability")


class CryoEMPredictor:
    def __init__(self, model_path, grids_path, output_path, save_output=True, device="cuda", quiet=False, batch_size=8,
                 reference_batching=False, gpus=None):
        """Same positional arguments as the reference (utils/predict.py:48).  Three additions:
        `gpus` - (default: the environment's MICA_GPUS, else 1) shard the tiles of the map over that many GPUs of the node
        (BASELINE.json configs[2]): this process is rank 0, ranks 1 .. N-1 are persistent child processes that mica_amd/multi.py starts
        (once, kept for the next map, gone with the process); the volumes GridCreator left on this GPU are broadcast to them over
        RCCL, every rank runs its share of the tile batches and the cropped records come back to this rank, which stitches.  The
        volumes are bit-identical to gpus=1.  When the tiler ran elsewhere, the volumes are rebuilt from its tile files first (a
        complete set: the tiler is a pure copy) and sharded the same way; an incomplete set is read tile by tile on one GPU, as the
        reference does;
        `batch_size` - tiles per forward call (results do not depend on it with per-tile gating);
        `reference_batching` - reproduce the reference's batching exactly (utils/predict.py:176-215, 278-286): batch 1
        up to `batch_threshold` tiles (= per-tile AF3 gating), above it batches of `optimal_batch_size` (at most 8) tiles in
        glob order with the AF3 test of models/model.py:60 taken over the WHOLE batch, so that a tile without atoms that
        shares a batch with one that has atoms goes through feat_conv/fusion as it does in the reference.  The default
        (False) gates per tile: results independent of file order and batch size."""
        self.model_path = model_path
        self.grids_path = grids_path
        self.output_path = output_path
        self.reconstruction_path = os.path.join(output_path, "results", os.path.normpath(self.grids_path).split(os.sep)[-1])
        self.save_output = save_output.lower() == 'true' if isinstance(save_output, str) else bool(save_output)
        self.device = device
        self.quiet = quiet
        self.batch_size = batch_size
        self.reference_batching = bool(reference_batching)
        self.batch_threshold = 200          # utils/predict.py:72 (an instance attribute there too)
        self.loader_threads = 4             # tile-file readers beside the GPU: 66.0 sub-grids/s with 4, 55.1 with 2 (profiles/r04_file_predictor.txt)
        self.use_optimized_batching = False
        self.optimal_batch_size = 1
        self.use_resident_volumes = True    # take the volumes a GridCreator of this process left on the GPU (handoff.py) instead of its files
        self.keep_resident_volumes = False  # True: leave them registered after the prediction (a second predictor on the same grids_path)
        from .multi import configured_gpus
        self.gpus = configured_gpus(gpus)
        self.rank_backend = None            # None: MICA_RANK_BACKEND, else "nccl" (= RCCL over xGMI); "gloo" rehearses N > 1 on one card
        self.rank_devices = None            # None: MICA_RANK_DEVICES, else 0 .. gpus-1 (rank 0's entry should be this predictor's device)
        self.gather_to_root = False         # records to the stitching rank alone (dist.gather) instead of the all-gather north_star names
        self.rank_pool = None               # the pool of the last multi-GPU run (its start-up report: profiles/r06_rank_startup.txt)
        self.resident = None                # (map entry, AF3 entry or None) once select_processing_strategy found them
        self.engine = None
        self.sample_count = 0
        self.timing_stats = {k: 0 for k in ('strategy_selection', 'model_loading', 'data_loading', 'inference',
                                            'reconstruction', 'saving', 'total')}
        self.logger = logging.getLogger(__name__)

    def _print_status(self, m):
        if not self.quiet:
            print(m)

    # ---- the in-process hand-off ---------------------------------------------------------------------------
    def _find_resident(self):
        """(map entry, AF3 entry or None) when the `GridCreator` mirror of THIS process cut `grids_path` and its volumes are still on
        the GPU (mica_amd/handoff.py); None -> the tile files are read, as the reference does.  `reference_batching` always reads the
        files: its batches are runs of FILES in glob order (utils/predict.py:278-286)."""
        if not self.use_resident_volumes or self.reference_batching:
            return None
        m = handoff.lookup_grids(os.path.join(self.grids_path, "normalized_map_grids"))
        if m is None or m.kind != "map":
            return None
        dev = torch.device(self.device if ":" in str(self.device) else "cuda:0")
        if m.volume.device != dev:
            return None                      # the volumes live on another GPU than the one this predictor was given: read the files
        afdir = os.path.join(self.grids_path, "AF3_encoding_grids")
        a = handoff.lookup_grids(afdir)
        if a is not None and (a.kind != "af3" or a.shape != m.shape or (a.grid_size, a.padding) != (m.grid_size, m.padding) or
                              a.volume.device != dev):
            a = None
        if a is None and os.path.isdir(afdir) and glob.glob(os.path.join(afdir, "*_grids", "*.npz")):
            return None                      # encodings tiled by somebody else: their files are the only copy
        return m, a

    def _wait_for_tile_files(self):
        """Before the files under grids_path are read: join the background writers of this process that are still producing them."""
        for d in ("normalized_map_grids", "AF3_encoding_grids"):
            e = handoff.lookup_grids(os.path.join(self.grids_path, d))
            if e is not None and e.writer is not None:
                e.writer.wait()

    # ---- steps (names as in the reference) ----------------------------------------------------------
    def select_processing_strategy(self):
        t0 = time.time()
        self.resident = self._find_resident()
        if self.resident is not None:
            from ._cabi import tile_table
            m = self.resident[0]
            self.sample_count = len(tile_table(*m.shape, m.grid_size))
            self.use_optimized_batching = self.sample_count > self.batch_threshold
            self.optimal_batch_size = 8 if self.use_optimized_batching else 1
            self.timing_stats['strategy_selection'] = time.time() - t0
            return True
        self._wait_for_tile_files()
        files = glob.glob(f"{self.grids_path}/normalized_map_grids/*.npz")
        self.sample_count = len(files)
        self.timing_stats['strategy_selection'] = time.time() - t0
        if not files:
            self.logger.error(f"No grid files found in: {self.grids_path}/normalized_map_grids/")
            return False
        # utils/predict.py:193-199.  _calculate_optimal_batch_size (:156-174) = min(8, 70 % of device memory minus 3x the
        # model over 31 MB per sample): with 288 GB of HBM that is always the cap of 8
        self.use_optimized_batching = self.sample_count > self.batch_threshold
        self.optimal_batch_size = 8 if self.use_optimized_batching else 1
        return True

    def load_model(self, tile_size=None):
        t0 = time.time()
        if tile_size is None:
            tile_size = self.resident[0].grid_size + 2 * self.resident[0].padding if self.resident is not None else 64
        try:
            if not os.path.exists(self.model_path):
                self.logger.error(f"Model file not found: {self.model_path}")
                return False
            if str(self.device).startswith("cpu"):
                raise RuntimeError("device='cpu': this build runs on MI355X only (no CPU path)")
            sd = load_checkpoint_state_dict(self.model_path)
            dev = torch.device(self.device if ":" in str(self.device) else "cuda:0")
            self.engine = Engine(dev, max_batch=max(self.batch_size, 8) if self.reference_batching else self.batch_size, tile_size=tile_size)
            self.engine.load_state_dict(sd)
            self.timing_stats['model_loading'] = time.time() - t0
            return True
        except Exception as e:
            self.logger.error(f"Model loading failed: {e}")
            self.timing_stats['model_loading'] = time.time() - t0
            return False

    def prepare_data(self):
        files = glob.glob(f"{self.grids_path}/normalized_map_grids/*.npz")
        if not files:
            self.logger.error(f"No grid files found in: {self.grids_path}/normalized_map_grids/")
            return False, None
        return True, CryoEMTestDataset(data_dir=files, transform=None)

    def run_inference(self, dataset):
        """Forward over all tiles and stitch (predict.py:307-398 + 439-512 fused).  Returns the volume dict or None."""
        t0 = time.time()
        try:
            e = self.engine
            S = e.tile_size
            meta = []
            for idx in range(len(dataset)):
                d = np.load(dataset.data_dir[idx])
                meta.append((int(d['i']), int(d['j']), int(d['k']), tuple(int(x) for x in np.asarray(d['orig_shape']).flatten()),
                             int(d['grid_size']) if 'grid_size' in d else 48, int(d['padding']) if 'padding' in d else 8))
            shape, grid, pad = meta[0][3], meta[0][4], meta[0][5]
            if grid + 2 * pad != S:
                raise RuntimeError(f"tile window {grid + 2 * pad} does not match the engine's {S}")
            nt1, nt2 = -(-shape[1] // grid), -(-shape[2] // grid)
            order = sorted(range(len(meta)), key=lambda t: ((meta[t][0] // grid) * nt1 + meta[t][1] // grid) * nt2 + meta[t][2] // grid)
            tindex = lambda t: ((meta[t][0] // grid) * nt1 + meta[t][1] // grid) * nt2 + meta[t][2] // grid
            out = torch.zeros((23, *shape), dtype=torch.float32, device=e.device)
            from .pipeline import SlabDownloader, volume_dict
            dl = SlabDownloader(out, grid, nt1 * nt2)        # finished x slabs travel to the host while later tiles compute
            B = e.max_batch
            rec = torch.empty((B, 23, S, S, S), dtype=torch.float32, device=e.device)
            pos = 0
            if self.reference_batching and self.use_optimized_batching:
                # the reference's DataLoader: consecutive files in glob order, batch-wide AF3 gate (model.py:60)
                bs = self.optimal_batch_size
                for g0 in range(0, len(meta), bs):
                    grp = list(range(g0, min(g0 + bs, len(meta))))
                    items = [dataset[t] for t in grp]
                    x = torch.from_numpy(np.stack([it[0] for it in items])).to(e.device)
                    af = torch.from_numpy(np.stack([it[1] for it in items])).to(e.device)
                    n = len(grp)
                    e.forward_records(x.view(n, S, S, S), af, rec[:n], af_mode=AF_BATCH)
                    for q, t in enumerate(grp):
                        e.stitch_tiles(rec[q:q + 1], out, grid, pad, tindex(t))
                pos = len(order)
            # runs of consecutive tiles; their files are read on a small thread pool a few runs ahead of the GPU (25 .npz files
            # = 26 MB per tile: the reference's DataLoader reads them on the main thread between forwards, predict.py:70,334)
            runs = []
            while pos < len(order):
                run = [order[pos]]
                while len(run) < B and pos + len(run) < len(order) and tindex(order[pos + len(run)]) == tindex(run[-1]) + 1:
                    run.append(order[pos + len(run)])
                runs.append(run)
                pos += len(run)
            from concurrent.futures import ThreadPoolExecutor
            ahead = 3
            copy_stream = torch.cuda.Stream(device=e.device)
            main_stream = torch.cuda.current_stream(e.device)
            pins = {}
            uploaded = {}        # buffer parity -> event of the last upload that read the pinned pair

            def stage(r, futs):
                """host side of run r, on a helper thread: wait for its tile files, assemble the batch in pinned memory (two
                alternating buffers) and start the upload on the copy stream; -> (map tiles, AF3 tiles or None, event)"""
                torch.cuda.set_device(e.device)
                items = [f.result() for f in futs]
                n, k = len(items), r & 1
                if k not in pins:
                    pins[k] = (torch.empty((B, 1, S, S, S), dtype=torch.float32).pin_memory(),
                               torch.empty((B, 24, S, S, S), dtype=torch.float32).pin_memory())
                px, pa = pins[k]
                if k in uploaded:
                    uploaded[k].synchronize()      # host-side: the non-blocking upload of run r - 2 has finished reading this pair
                any_af = False
                for q, it in enumerate(items):
                    px[q].copy_(torch.from_numpy(it[0]))
                    pa[q].copy_(torch.from_numpy(it[1]))
                    any_af = any_af or bool(np.any(it[1]))
                with torch.cuda.stream(copy_stream):
                    x = px[:n].to(e.device, non_blocking=True)
                    af = pa[:n].to(e.device, non_blocking=True) if any_af else None
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                uploaded[k] = ev
                return x, af, ev

            with ThreadPoolExecutor(max_workers=self.loader_threads) as pool, ThreadPoolExecutor(max_workers=1) as stager:
                loads = [[pool.submit(dataset.__getitem__, t) for t in run] for run in runs[:ahead]]
                staged = [stager.submit(stage, 0, loads.pop(0))] if runs else []
                for r, run in enumerate(runs):
                    if r + ahead < len(runs):
                        loads.append([pool.submit(dataset.__getitem__, t) for t in runs[r + ahead]])
                    if r + 1 < len(runs):
                        # the pinned buffers of parity (r + 1) & 1 were last read by the upload of run r - 1: stage() waits for that
                        # upload's event on the host before it rewrites them (it does not rely on the forward being synchronous)
                        staged.append(stager.submit(stage, r + 1, loads.pop(0)))
                    x, af, ev = staged.pop(0).result()
                    main_stream.wait_event(ev)
                    n = len(run)
                    e.forward_records(x.view(n, S, S, S), af, rec[:n], af_mode=AF_PER_TILE)
                    e.stitch_tiles(rec[:n], out, grid, pad, tindex(run[0]))
                    dl.tiles_done(tindex(run[-1]) + 1)       # runs come in table order: every tile before this index that exists is done
                    x.record_stream(main_stream)
                    if af is not None:
                        af.record_stream(main_stream)
            vols = volume_dict(dl.finish())
            self.timing_stats['inference'] = time.time() - t0
            return vols
        except Exception as e:
            self.logger.error(f"Inference failed: {e}")
            self.timing_stats['inference'] = time.time() - t0
            return None
        finally:
            if 'dl' in locals():
                dl.close()

    def run_inference_resident(self):
        """The hot loop on the volumes GridCreator left on the GPU: gather -> forward -> softmax / argmax -> stitch per batch of
        tiles, per-tile AF3 gating (= the reference at batch size 1 and = the file route of this class), the four volumes
        downloaded slab by slab while later tiles compute.  A directory without encodings, or with fewer than the 24 channels,
        means zeros for every tile (dataset/dataset.py:218-219) = the exp_downsizing branch."""
        m, a = self.resident
        for ent in self.resident:            # the tile-file writers held back for the tiler's and the weight load's sake: go
            if ent is not None and ent.writer is not None:
                ent.writer.release()
        af = a.volume if a is not None and len(a.channels) == 24 else None
        return self._predict_volumes(m.volume, af, m.grid_size, m.padding)

    def _predict_volumes(self, vol, af, grid, pad):
        """vol f32 [N0,N1,N2] / af u8 or f32 [24,N0,N1,N2] (or None) on this predictor's GPU -> the dict of four host volumes, on one GPU
        or - `gpus` > 1 - sharded over the ranks of mica_amd/multi.py; None (+ log) on failure, like run_inference."""
        t0 = time.time()
        try:
            from .pipeline import VolumePredictor
            if self.gpus > 1:
                from . import multi
                self.rank_pool = pool = multi.get_pool(self.gpus, tile=self.engine.tile_size, batch=self.engine.max_batch, backend=self.rank_backend,
                                                       devices=self.rank_devices, conv_variant=None)
                runner = multi.EngineRunner(None, self.engine.tile_size, self.engine.max_batch, engine=self.engine, loaded_model=self.model_path)
                try:
                    vols = pool.predict(runner, self.model_path, vol, af, grid, pad, gather_to_root=self.gather_to_root, to_host=True)
                finally:
                    if runner.engine is not self.engine:      # the runner had to build a fresh context (the checkpoint changed under it)
                        self.engine.close()
                        self.engine = runner.engine
            else:
                vols = VolumePredictor(self.engine, grid, pad, self.engine.max_batch).predict_volume(vol, af, to_host=True)
            self.timing_stats['inference'] = time.time() - t0
            return vols
        except Exception as e:
            self.logger.error(f"Inference failed: {e}")
            self.timing_stats['inference'] = time.time() - t0
            return None

    def _volumes_from_tile_files(self):
        """The map (and the 24 encoding channels) rebuilt from a COMPLETE set of tile files - the tiler is a pure copy and the tiles'
        central regions cover the volume exactly once (utils/create_grids.py:143-157, utils/predict.py:494-501) - so that `gpus` > 1
        also serves a predictor whose tiler ran elsewhere: -> (vol f32 device, af u8 / f32 device or None, grid, pad), or None when the
        set is not one this shortcut reproduces exactly (a tile of the table missing, an encoding tile missing or unreadable - the
        reference then feeds zeros for that tile's 24 channels, dataset/dataset.py:218-219 -, windows of another size than the engine's):
        the caller falls back to reading tile by tile on one GPU."""
        from concurrent.futures import ThreadPoolExecutor
        from ._cabi import tile_table
        from .dataset import AF3_TYPES, read_npz_grid
        files = glob.glob(f"{self.grids_path}/normalized_map_grids/*.npz")
        if not files:
            return None
        metas = []
        for f in files:
            d = np.load(f)
            metas.append((int(d['i']), int(d['j']), int(d['k']), int(d['di']), int(d['dj']), int(d['dk']),
                          tuple(int(x) for x in np.asarray(d['orig_shape']).flatten()), int(d['grid_size']) if 'grid_size' in d else 48,
                          int(d['padding']) if 'padding' in d else 8))
        shape, grid, pad = metas[0][6], metas[0][7], metas[0][8]
        table = {tuple(int(v) for v in row[:3]) for row in tile_table(*shape, grid)}
        if any(m[6:] != (shape, grid, pad) for m in metas) or {m[:3] for m in metas} != table or len(metas) != len(table):
            return None
        afdir = os.path.join(self.grids_path, "AF3_encoding_grids")
        with_af = os.path.isdir(afdir)
        vol = np.zeros(shape, np.float32)
        af = np.zeros((24, *shape), np.uint8) if with_af else None
        state = {"af": af, "ok": True}

        def place(q):
            i, j, k, di, dj, dk = metas[q][:6]
            core = (slice(pad, pad + di), slice(pad, pad + dj), slice(pad, pad + dk))
            dst = (slice(i, i + di), slice(j, j + dj), slice(k, k + dk))
            vol[dst] = read_npz_grid(files[q])[core]
            if with_af:
                tiles = []
                for t in AF3_TYPES:
                    p = files[q].replace('normalized_map_grids', f"AF3_encoding_grids/{t}_grids").replace('normalized_map', f"{t}")
                    tiles.append(read_npz_grid(p)[core])
                return q, dst, tiles
            return q, dst, None
        try:
            with ThreadPoolExecutor(max_workers=self.loader_threads) as pool:
                for q, dst, tiles in pool.map(place, range(len(files))):
                    if tiles is None:
                        continue
                    for c, g in enumerate(tiles):
                        a = state["af"]
                        if a.dtype == np.uint8:
                            u = g.astype(np.uint8)
                            if not np.array_equal(u, g):                 # not a binary encoding: keep the values as they are
                                state["af"] = a = a.astype(np.float32)
                                a[(c, *dst)] = g
                            else:
                                a[(c, *dst)] = u
                        else:
                            a[(c, *dst)] = g
        except Exception as e:
            self.logger.warning(f"tile files under {self.grids_path} are not a complete set ({e}): reading tile by tile on one GPU")
            return None
        dev = torch.device(self.device if ":" in str(self.device) else "cuda:0")
        return torch.from_numpy(vol).to(dev), None if state["af"] is None else torch.from_numpy(state["af"]).to(dev), grid, pad

    @staticmethod
    def close_ranks():
        """End the worker processes of `gpus` > 1 now (they otherwise stay for the next map and leave at interpreter exit)."""
        from . import multi
        multi.shutdown()

    def run_prediction(self):
        t0 = time.time()
        try:
            if not self.select_processing_strategy():
                return False, {}
            if self.gpus > 1 and self.resident is not None:
                # the workers' python + torch import and 37-GB workspaces come up beside this rank's weight load (if the Solver-flow
                # shim has not started them already, beside the whole of getData)
                from . import multi
                multi.get_pool(self.gpus, tile=self.resident[0].grid_size + 2 * self.resident[0].padding, batch=self.batch_size,
                               backend=self.rank_backend, devices=self.rank_devices, conv_variant=None).spawn()
            from_files = None
            if self.gpus > 1 and self.resident is None and not self.reference_batching:
                # the tiler ran elsewhere: rebuild the volumes from its (complete) tile files and shard those
                from_files = self._volumes_from_tile_files()
                if from_files is None:
                    self.logger.warning("gpus > 1: the tile files are not a complete set this route reproduces exactly; reading tile by tile on one GPU")
                else:
                    from . import multi
                    multi.get_pool(self.gpus, tile=from_files[2] + 2 * from_files[3], batch=self.batch_size, backend=self.rank_backend,
                                   devices=self.rank_devices, conv_variant=None).spawn()
            if not self.load_model(tile_size=None if from_files is None else from_files[2] + 2 * from_files[3]):
                return False, {}
            if self.resident is not None:
                vols = self.run_inference_resident()
            elif from_files is not None:
                vols = self._predict_volumes(*from_files)
            else:
                ok, dataset = self.prepare_data()
                if not ok:
                    return False, {}
                vols = self.run_inference(dataset)
            if vols is None:
                return False, {}
            if self.save_output:
                os.makedirs(self.reconstruction_path, exist_ok=True)
                for k, v in vols.items():
                    np.save(f'{self.reconstruction_path}/{k}.npy', v)
            self.timing_stats['total'] = time.time() - t0
            return True, vols
        except Exception as e:
            self.logger.error(f"Prediction pipeline failed: {e}")
            return False, {}
        finally:
            if self.resident is not None:
                # the tile files the caller asked GridCreator for are complete when this returns - on every way out (utils/modeler.py:755
                # deletes grids_path right after nnPred: nothing may still be writing into it)
                for e in self.resident:
                    if e is not None and e.writer is not None:
                        handoff.join_writer(e.writer)
                handoff.flush()              # ... nor into the normalised map / the encodings, which it deletes as well (:756-757)
                if not self.keep_resident_volumes:
                    # the grids are consumed: give back the HBM behind them (0.5 + 3.2 GB of transposed volumes and as much again in
                    # the MRC stage's copies at 512^3) instead of waiting for 32 newer entries to evict them.  The FILES stay: a second
                    # predictor on the same grids_path reads them, like the reference.
                    for d, e in zip(("normalized_map_grids", "AF3_encoding_grids"), self.resident):
                        if e is not None:
                            for f in e.sources:
                                handoff.drop_file(f)
                            handoff.drop_grids(os.path.join(self.grids_path, d))
                            e.release()
            if self.engine is not None:
                self.engine.close()
                self.engine = None
