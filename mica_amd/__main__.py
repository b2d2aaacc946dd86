"""Thin command-line entry for the hot path alone (the reference's run.py drives the whole MICA pipeline; this produces the four
volumes Solver.nnPred consumes, utils/modeler.py:735-738, and stops there):

    python -m mica_amd --map emd_1234.mrc --model MICA_best_model.pth [--docked-model 1234_af3_docked.pdb] --out results/

MRC -> resample to 1 A + normalise (GPU) -> optional AF3 encodings rasterised from the docked model (GPU) -> 64^3 windows
(grid 48 + 2 x 8 halo) -> network -> softmax / argmax -> stitched volumes, written as <out>/<key>.npy like the reference's
save_output (utils/predict.py:555-558), indexed (x, y, z); the tiler's offset is printed.  Nothing touches the disk in between.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m mica_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("--map", required=True, help="cryo-EM density map (MRC mode 2)")
    ap.add_argument("--model", required=True, help="checkpoint with 'model_state_dict' (reference train.py:298-304)")
    ap.add_argument("--docked-model", default=None, help="docked AlphaFold3 model (PDB) for the 24 encoding channels")
    ap.add_argument("--out", required=True, help="output directory for the four .npy volumes")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--batch", type=int, default=8, help="tiles per forward call")
    ap.add_argument("--grid-size", type=int, default=48)
    ap.add_argument("--padding", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=None, help="shard the tiles over this many GPUs of the node (default: MICA_GPUS, else 1): this process "
                    "is rank 0, the others are child processes it starts (mica_amd/multi.py)")
    ap.add_argument("--rank-backend", default=None, help="nccl (= RCCL over xGMI, default) | gloo (rehearsal of N > 1 on one card, host-staged)")
    ap.add_argument("--rank-devices", default=None, help="comma-separated device index per rank (default 0 .. gpus-1)")
    args = ap.parse_args(argv)

    from . import mrc
    from .engine import Engine
    from .pipeline import VolumePredictor
    from .preprocessing import DataPreprocessor
    from .weights import load_checkpoint_state_dict

    from . import multi
    gpus = multi.configured_gpus(args.gpus)
    pool = None
    if gpus > 1:            # the workers start first: their import and workspace allocation run beside this rank's own start-up and normaliser
        pool = multi.get_pool(gpus, tile=args.grid_size + 2 * args.padding, batch=args.batch, backend=args.rank_backend,
                              devices=None if args.rank_devices is None else [int(v) for v in args.rank_devices.split(",")]).spawn()
    t0 = time.time()
    eng = Engine(args.device, max_batch=args.batch, tile_size=args.grid_size + 2 * args.padding)
    eng.load_state_dict(load_checkpoint_state_dict(args.model))
    data, hd = mrc.read_mrc(args.map)
    dp = DataPreprocessor(args.map, os.path.join(args.out, "AF3_results"), quiet=True, engine=eng)
    norm, med, pct = dp.normalize_array(np.asarray(data), hd.voxel_size, 1.0)
    # the encodings are rasterised on the resampled grid with the map's origin (preprocessing.py:236-252), then both go through
    # the tiler's axis transform (create_grids.py:119-122)
    hdn = mrc.MrcHeader(nx=norm.shape[2], ny=norm.shape[1], nz=norm.shape[0], mapc=hd.mapc, mapr=hd.mapr, maps=hd.maps,
                        nxstart=hd.nxstart, nystart=hd.nystart, nzstart=hd.nzstart, origin=hd.origin)
    vol, offset = mrc.transpose_to_xyz(norm, hdn)
    d_vol = torch.from_numpy(np.ascontiguousarray(vol)).to(eng.device)
    d_af = None
    if args.docked_model:
        enc = dp.encode_AF3_volume(args.docked_model, hd.origin, norm.shape)            # [24, nz, ny, nx] on the device
        perm = _axis_perm(hdn)
        d_af = enc.permute(0, *[1 + p for p in perm]).contiguous()
    if pool is not None:
        runner = multi.EngineRunner(None, eng.tile_size, args.batch, engine=eng, loaded_model=args.model)
        vols = pool.predict(runner, args.model, d_vol, None if d_af is None else d_af.to(torch.uint8), args.grid_size, args.padding, to_host=True)
        print(pool.startup_report())
        pool.close()
    else:
        vols = VolumePredictor(eng, args.grid_size, args.padding, args.batch).predict_volume(d_vol, d_af, to_host=True)
    os.makedirs(args.out, exist_ok=True)
    for k, v in vols.items():
        np.save(os.path.join(args.out, f"{k}.npy"), v)
    T = int(eng.lib.mica_tile_count(*d_vol.shape, args.grid_size))
    print(f"map {tuple(data.shape)} -> {tuple(d_vol.shape)} (x, y, z), median {med:.6g}, 99.9th percentile {pct:.6g}, offset {offset}, "
          f"{T} sub-grids, {time.time() - t0:.1f} s; volumes in {args.out}")
    eng.close()
    return 0


def _axis_perm(hd):
    """The permutation GridCreator.transpose applies to [section, row, column] data (create_grids.py:67-87)."""
    axis_order = [hd.maps - 1, hd.mapr - 1, hd.mapc - 1]
    return [j for i in range(3) for j in range(3) if axis_order[j] == i]


if __name__ == "__main__":
    sys.exit(main())
