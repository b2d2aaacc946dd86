"""Host-side logic without a GPU: MRC I/O, tile-file dataset, predictor failure conventions and the
multi-rank sharding over gloo (world_size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mica_amd import mrc
from mica_amd.dist import batch_plan, rank_batches, sharded_records
from mica_amd.synth import synth_density
from oracle import volume_oracle as vo


def test_mrc_round_trip_and_axis_orders(tmp_path):
    data = synth_density((5, 7, 9), 1)            # [nz, ny, nx]
    p = str(tmp_path / "a.mrc")
    mrc.write_mrc(p, data, voxel_size=(1.5, 1.25, 0.8), origin=(3.0, -2.0, 1.0), nxstart=5, nystart=6, nzstart=7)
    got, hd = mrc.read_mrc(p)
    assert np.array_equal(got, data) and (hd.nx, hd.ny, hd.nz) == (9, 7, 5) and hd.mode == 2
    assert np.allclose(hd.voxel_size, (1.5, 1.25, 0.8)) and hd.origin == (3.0, -2.0, 1.0)
    vol, off = mrc.transpose_to_xyz(got, hd)
    rv, roff = vo.transpose_axes(data, 1, 2, 3, [7, 6, 5])
    assert np.array_equal(vol, rv) and off == roff == [5.0, 6.0, 7.0] and vol.shape == (9, 7, 5)
    for axes in ((3, 2, 1), (2, 1, 3)):
        mrc.write_mrc(p, data, mapc=axes[0], mapr=axes[1], maps=axes[2], nxstart=5, nystart=6, nzstart=7)
        got, hd = mrc.read_mrc(p)
        vol, off = mrc.transpose_to_xyz(got, hd)
        rv, roff = vo.transpose_axes(data, *axes, [7, 6, 5])
        assert np.array_equal(vol, rv) and off == roff
    with open(p, "r+b") as f:
        f.truncate(1024 + 10)
    with pytest.raises(ValueError, match="truncated"):
        mrc.read_mrc(p)


def _write_tiles(root, shape, with_af):
    vol = synth_density(shape, 3)
    tiles, idx = vo.tile_volume(vol, 48, 8)
    d = os.path.join(root, "normalized_map_grids")
    os.makedirs(d)
    for t, (i, j, k, di, dj, dk) in enumerate(idx):
        np.savez(os.path.join(d, f"normalized_map_grid_i{i}_j{j}_k{k}.npz"), grid=tiles[t], i=i, j=j, k=k, di=di, dj=dj, dk=dk,
                 orig_shape=shape, grid_size=48, padding=8)
        if with_af:
            for ch in ("CA", "N"):
                dd = os.path.join(root, "AF3_encoding_grids", f"{ch}_grids")
                os.makedirs(dd, exist_ok=True)
                np.savez(os.path.join(dd, f"{ch}_grid_i{i}_j{j}_k{k}.npz"), grid=np.ones((64, 64, 64), np.float32))
    return idx


def test_tile_dataset_reads_reference_layout_and_zero_fallback(tmp_path):
    from mica_amd.dataset import CryoEMTestDataset
    import glob
    idx = _write_tiles(str(tmp_path), (50, 40, 40), with_af=True)      # only 2 of 24 AF3 channels present
    files = sorted(glob.glob(str(tmp_path / "normalized_map_grids" / "*.npz")))
    ds = CryoEMTestDataset(files, None)
    assert len(ds) == len(idx) == 2
    x, af, meta = ds[0]
    assert x.shape == (1, 64, 64, 64) and x.dtype == np.float32
    assert af.shape == (24, 64, 64, 64) and not af.any()               # any missing channel => all zeros (dataset.py:218-219)
    assert meta["filename"].startswith("normalized_map_grid_i") and int(meta["di"]) in (48, 2)


def test_predictor_failure_conventions(tmp_path):
    """(False, {}) + log, never an exception (utils/predict.py:212-215,632-634)."""
    from mica_amd.predict import CryoEMPredictor
    p = CryoEMPredictor(str(tmp_path / "none.pth"), str(tmp_path) + "/", str(tmp_path / "out"), save_output=False, quiet=True)
    assert p.run_prediction() == (False, {})                            # no grid files
    _write_tiles(str(tmp_path), (50, 40, 40), with_af=False)
    assert p.run_prediction() == (False, {})                            # model file not found
    assert CryoEMPredictor("m", "g/", "o", save_output="False").save_output is False
    assert CryoEMPredictor("m", "g/", "o", save_output="true").save_output is True


def test_batch_plan_covers_every_tile_once():
    for T, B, W in ((1331, 8, 8), (216, 8, 2), (5, 4, 3), (12, 4, 2), (1, 8, 4)):
        plan = batch_plan(T, B)
        assert sum(c for _, c in plan) == T and [f for f, _ in plan] == list(range(0, T, B))
        seen = []
        rounds = None
        for r in range(W):
            mine, rd = rank_batches(T, B, r, W)
            rounds = rd if rounds is None else rounds
            assert rd == rounds == len(mine)
            seen += [(f, c) for _, f, c in mine if c]
        assert sorted(seen) == plan


def _producer(first, count):
    # record = f(global tile index): [count, 2, 4, 4, 4]
    t = torch.arange(first, first + count, dtype=torch.float32).view(count, 1, 1, 1, 1)
    return t + torch.arange(2 * 64, dtype=torch.float32).view(1, 2, 4, 4, 4) / 1000.0


def _worker(rank, world, port, T, B, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = {}
    sharded_records(_producer, lambda rec, first: got.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"),
                    stitch_rank=0)
    if rank == 0:
        torch.save(got, out_path)
    else:
        assert not got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,B", [(13, 4), (8, 4), (3, 4)])
def test_sharded_records_gloo_world2(tmp_path, T, B):
    port = 29500 + (os.getpid() + T) % 2000
    out = str(tmp_path / "got.pt")
    mp.spawn(_worker, args=(2, port, T, B, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    single = {}
    sharded_records(_producer, lambda rec, first: single.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"))
    assert sorted(got) == sorted(single) == [f for f, _ in batch_plan(T, B)]
    for f in single:
        assert torch.equal(got[f], single[f])
