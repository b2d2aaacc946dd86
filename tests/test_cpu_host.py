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


def test_mrc_header_stats_of_a_binary_volume_from_its_count(tmp_path):
    """DataPreprocessor.create_AF3_encodings hands write_mrc the statistics of a 0/1 channel computed from one count
    (mica_amd/preprocessing.py); the file must be byte-identical to the one whose statistics come from the four float64 passes."""
    rng = np.random.default_rng(11)
    for dens, shape in ((1e-3, (40, 36, 44)), (0.37, (16, 20, 12)), (0.0, (8, 8, 8)), (1.0, (8, 8, 8))):
        v = (rng.random(shape) < dens).astype(np.float32)
        k, n = float(v.sum(dtype=np.float64)), float(v.size)
        p1 = k / n
        st = (0.0 if k < n else 1.0, 1.0 if k > 0 else 0.0, p1, float(np.sqrt(p1 * (1.0 - p1))))
        a, b = str(tmp_path / "a.mrc"), str(tmp_path / "b.mrc")
        mrc.write_mrc(a, v, origin=(1.5, -2.0, 3.0), nxstart=4)
        mrc.write_mrc(b, v, origin=(1.5, -2.0, 3.0), nxstart=4, stats=st)
        assert open(a, "rb").read() == open(b, "rb").read(), dens


def test_mrc_reader_against_hand_packed_mrc2014_headers(tmp_path):
    """The reader against files packed here word by word from the MRC2014 layout (words 1-10 nx ny nz mode n[xyz]start m[xyz], 11-16
    cell, 17-19 mapc mapr maps, 20-22 dmin dmax dmean, 23 ispg, 24 nsymbt, 50-52 origin, 53 'MAP ', 54 machine stamp, 55 rms), i.e.
    independent of write_mrc: a big-endian int16 file with an extended header and permuted axes, a little-endian uint16 file with the
    0x44 0x41 stamp, sampling m[xyz] != n[xyz]; files without the map ID or with an unknown stamp are refused as mrcfile refuses them
    in its default mode (create_grids.py:108).  mrcfile itself is absent here: header semantics stay 'parity unpinned'."""
    import struct
    nx, ny, nz = 4, 3, 2
    vals = np.arange(nx * ny * nz, dtype=np.int16).reshape(nz, ny, nx) * 7 - 30

    def header(end, mode, stamp, nsymbt=0, axes=(1, 2, 3), m=(nx, ny, nz), cell=(8.0, 4.5, 3.0), origin=(1.5, -2.0, 0.25), map_id=b"MAP "):
        h = bytearray(1024)
        struct.pack_into(end + "10i", h, 0, nx, ny, nz, mode, -3, 5, 11, *m)
        struct.pack_into(end + "6f", h, 40, *cell, 90.0, 90.0, 90.0)
        struct.pack_into(end + "3i", h, 64, *axes)
        struct.pack_into(end + "3f", h, 76, -30.0, 131.0, 50.5)
        struct.pack_into(end + "2i", h, 88, 1, nsymbt)
        struct.pack_into(end + "3f", h, 196, *origin)
        h[208:212] = map_id
        h[212:216] = bytes(stamp)
        struct.pack_into(end + "f", h, 216, 48.3)
        return bytes(h)

    p = str(tmp_path / "be.mrc")
    with open(p, "wb") as f:
        f.write(header(">", 1, (0x11, 0x11, 0, 0), nsymbt=80, axes=(2, 1, 3), m=(8, 6, 2)))
        f.write(b"\x5a" * 80)                                # extended header: skipped, kept in hd.extra
        f.write(vals.astype(">i2").tobytes())
    got, hd = mrc.read_mrc(p)
    assert got.dtype == np.int16 and got.dtype.isnative and np.array_equal(got, vals)
    assert (hd.nx, hd.ny, hd.nz, hd.mode, hd.nxstart, hd.nystart, hd.nzstart) == (4, 3, 2, 1, -3, 5, 11)
    assert (hd.mapc, hd.mapr, hd.maps) == (2, 1, 3) and hd.nsymbt == 80 and hd.extra == b"\x5a" * 80
    assert hd.voxel_size == (1.0, 0.75, 1.5) and hd.origin == (1.5, -2.0, 0.25)          # cella / m[xyz], not / n[xyz]
    vol, off = mrc.transpose_to_xyz(got, hd)
    rv, roff = vo.transpose_axes(vals, 2, 1, 3, [11, 5, -3])
    assert np.array_equal(vol, rv) and off == roff

    p2 = str(tmp_path / "le.mrc")
    u = (vals.astype(np.int32) + 40000).astype(np.uint16)
    with open(p2, "wb") as f:
        f.write(header("<", 6, (0x44, 0x41, 0, 0)))
        f.write(u.astype("<u2").tobytes())
    got, hd = mrc.read_mrc(p2)
    assert got.dtype == np.uint16 and np.array_equal(got, u) and hd.voxel_size == (2.0, 1.5, 1.5)

    for bad in (header("<", 2, (0x44, 0x44, 0, 0), map_id=b"\0\0\0\0"), header("<", 2, (0x00, 0x00, 0, 0)), header("<", 3, (0x44, 0x44, 0, 0))):
        with open(p2, "wb") as f:
            f.write(bad + b"\0" * 96)
        with pytest.raises(ValueError):
            mrc.read_mrc(p2)


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


@pytest.mark.parametrize("T,B,world", [(13, 4, 2), (8, 4, 2), (3, 4, 2), (29, 4, 4), (5, 2, 3), (133, 8, 8)])
def test_sharded_records_gloo_world2(tmp_path, T, B, world):
    """world 2 ... 4: more ranks than batches in the last round, ranks that get no batch at all ((3, 4, 2), (5, 2, 3)); (133, 8, 8) is
    the rank count and batch size of BASELINE.json's configs[2] with a tenth of the default tiling's 1331 windows (ragged last batch,
    a last round that only five of the eight ranks take part in)."""
    port = 29500 + (os.getpid() + 7 * T + world) % 2000
    out = str(tmp_path / "got.pt")
    mp.spawn(_worker, args=(world, port, T, B, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    single = {}
    sharded_records(_producer, lambda rec, first: single.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"))
    assert sorted(got) == sorted(single) == [f for f, _ in batch_plan(T, B)]
    for f in single:
        assert torch.equal(got[f], single[f])


def _bench_worker(rank, world, port, out_path):
    """bench.py's N > 1 step loop in miniature: RecordExchange.post per step with a wrapping tile layout."""
    from mica_amd.dist import RecordExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, nb, got = 2, 3, []                                     # 3 whole batches per "map": step k of rank r takes batch (k*world + r) % nb
    first_of = lambda k, r: ((k * world + r) % nb) * B
    ex = RecordExchange(B, (2, 4, 4, 4), torch.device("cpu"), lambda rec, first: got.append((first, rec.clone())), stitch_rank=0)
    for k in range(5):
        ex.post(k, _producer(first_of(k, rank), B) + 100.0 * k, [(first_of(k, r), B) for r in range(world)])
        assert len(got) == (2 * k if rank == 0 else 0)        # step k-1 is stitched while step k is in flight
    ex.flush()
    if rank == 0:
        torch.save(got, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_record_exchange_step_loop_gloo_world2(tmp_path):
    port = 29500 + (os.getpid() + 77) % 2000
    out = str(tmp_path / "steps.pt")
    mp.spawn(_bench_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert len(got) == 10
    for i, (first, rec) in enumerate(got):
        k, r = divmod(i, 2)                                   # stitched in step order, rank order within a step
        assert first == ((k * 2 + r) % 3) * 2
        assert torch.equal(rec, _producer(first, 2) + 100.0 * k)


def _force_worker(rank, world, port, T, B, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got, stats = {}, {}
    sharded_records(_producer, lambda rec, first: got.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"),
                    force_collective=True, stats=stats)
    torch.save({"got": got, "stats": stats}, out_path)
    dist.destroy_process_group()


def test_force_collective_in_a_group_of_one_rank(tmp_path):
    """`force_collective`: a process group of ONE rank still takes the collective path of RecordExchange (receive buffers, an
    all_gather_into_tensor per round, work.wait(), slot reuse) - the switch that lets the RCCL branch run on a one-GPU box
    (tests/test_gpu_configs.py::test_config2_rccl_branch_single_rank_equals_plain_pipeline; here over gloo on CPU tensors)."""
    T, B = 13, 4
    port = 29500 + (os.getpid() + 311) % 2000
    out = str(tmp_path / "force.pt")
    mp.spawn(_force_worker, args=(1, port, T, B, out), nprocs=1, join=True)
    r = torch.load(out, weights_only=True)
    assert r["stats"]["collectives"] == len(batch_plan(T, B)) == 4 and r["stats"]["world"] == 1 and r["stats"]["backend"] == "gloo"
    single = {}
    st = {}
    sharded_records(_producer, lambda rec, first: single.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"), stats=st)
    assert st["collectives"] == 0                            # no group, no switch: the plain path
    assert sorted(r["got"]) == sorted(single)
    for f in single:
        assert torch.equal(r["got"][f], single[f])


def _pdb_line(rec, serial, name, altloc, resname, chain, resseq, x, y, z, occ=1.0):
    nm = name if len(name) == 4 else " " + name.ljust(3)
    return "%-6s%5d %4s%1s%3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % (
        rec, serial, nm, altloc, resname, chain, resseq, x, y, z, occ, 20.0, name[0])


def test_af3_pdb_reader_and_oracle_rasteriser(tmp_path):
    """PDB reader of the docked model (fixed columns as PDBIO writes them; Bio is absent, so this step is pinned only by
    files written here) and the oracle restatement of preprocessing.py:172-178,283-298."""
    from mica_amd import af3_encoding as ae
    from oracle import af3_oracle as ao

    assert ae.CHANNEL_NAMES == ao.CHANNEL_NAMES and len(ae.CHANNEL_NAMES) == 24
    lines = ["MODEL        1\n",
             _pdb_line("ATOM", 1, "N", " ", "ALA", "A", 1, 1.2, 2.5, 3.5),        # 2.5 and 3.5 round half to even: 2, 4
             _pdb_line("ATOM", 2, "CA", " ", "ALA", "A", 1, 2.49, 2.51, -0.4),
             _pdb_line("ATOM", 3, "CB", "A", "ALA", "A", 1, 5.0, 5.0, 5.0, occ=0.3),
             _pdb_line("ATOM", 4, "CB", "B", "ALA", "A", 1, 6.0, 6.0, 6.0, occ=0.7),  # higher occupancy altloc wins
             _pdb_line("ATOM", 5, "O", " ", "MSE", "A", 2, 7.0, 1.0, 1.0),        # non-standard residue: backbone channel only
             _pdb_line("HETATM", 6, "O", " ", "HOH", "A", 3, 3.0, 3.0, 3.0),      # hetero flag -> skipped
             _pdb_line("ATOM", 7, "HD11", " ", "LEU", "B", 9, 100.0, -50.0, 4.4),  # hydrogen, clipped on both sides
             "ENDMDL\n"]
    pdb = tmp_path / "x_af3_docked.pdb"
    pdb.write_text("".join(lines))
    coords, names, res = ae.read_pdb_atoms(str(pdb))
    assert names == ["N", "CA", "CB", "O", "HD11"] and res == ["ALA", "ALA", "ALA", "MSE", "LEU"]
    assert coords.dtype == np.float32 and np.allclose(coords[2], [6, 6, 6])
    bb, aa = ae.channel_indices(names, res)
    assert bb.tolist() == [1, 0, -1, 3, -1] and aa.tolist() == [4, 4, 4, -1, 4 + 9]
    # models: two MODEL records with the same serial, then atoms behind ENDMDL without a MODEL record - three models, the same
    # (chain, residue, atom) in each of them is kept three times
    ca = _pdb_line("ATOM", 1, "CA", " ", "GLY", "A", 1, 1.0, 1.0, 1.0)
    pdb.write_text("MODEL        1\n" + ca + "ENDMDL\nMODEL        1\n" + ca + "ENDMDL\n" + ca + "END\n")
    assert len(ae.read_pdb_atoms(str(pdb))[1]) == 3

    shape = (9, 9, 9)
    vol = ao.rasterise_atoms(coords, names, res, (0.0, 0.0, 0.0), shape)
    assert vol.shape == (24, 9, 9, 9) and vol.dtype == np.float32
    assert vol[1, 4, 2, 1] == 1 and vol[4, 4, 2, 1] == 1            # N of ALA at (x=1, y=2 (2.5 -> 2), z=4 (3.5 -> 4))
    assert vol[0, 0, 3, 2] == 1                                       # CA: z = -0.4 -> 0
    assert vol[4, 6, 6, 6] == 1 and vol[4, 5, 5, 5] == 0              # the altloc with occupancy 0.7
    assert vol[3, 1, 1, 7] == 1 and vol[4:, 1, 1, 7].sum() == 0       # MSE: only the O channel
    assert vol[13, 4, 0, 8] == 1                                      # LEU hydrogen, clipped to (8, 0, 4)
    assert vol.sum() == 2 + 2 + 1 + 1 + 1
    # the reference clips x against shape[0] (nz) and z against shape[2] (nx): on a non-cubic map a far-out z raises
    with pytest.raises(IndexError):
        ao.rasterise_atoms(np.array([[0, 0, 30]], np.float32), ["CA"], ["GLY"], (0, 0, 0), (4, 5, 40))
    # ... and x beyond nz-1 is silently clipped to nz-1
    v2 = ao.rasterise_atoms(np.array([[30, 0, 0]], np.float32), ["CA"], ["GLY"], (0, 0, 0), (4, 5, 40))
    assert v2[0, 0, 0, 3] == 1


@pytest.mark.parametrize("name,atoms", [("model.pdb", 209), ("model.rebuilt.pdb", None)])
def test_pdb_reader_on_the_pdb_files_the_reference_ships(name, atoms):
    """The only PDB files inside the reference (modules/pulchra304/examples: a CA trace without occupancy / B-factor columns
    and a rebuilt all-atom model with a blank chain id) through the fixed-column reader, against an independent
    whitespace-split reading of the same records.  Bio.PDB is absent, so this pins the reader's handling of real short
    records, not Bio's semantics (that part stays 'parity unpinned'); skipped where /root/reference is absent."""
    import os
    from mica_amd import af3_encoding as ae
    path = os.path.join("/root/reference/modules/pulchra304/examples", name)
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    ref_names, ref_res, ref_xyz = [], [], []
    for line in open(path):
        if line.startswith("ATOM"):
            tok = line.split()
            has_chain = not tok[4].lstrip("-").isdigit()
            ref_names.append(tok[2])
            ref_res.append(tok[3])
            ref_xyz.append([float(v) for v in tok[(6 if has_chain else 5):(9 if has_chain else 8)]])
    coords, names, res = ae.read_pdb_atoms(path)
    assert names == ref_names and res == ref_res and len(names) > 100
    if atoms is not None:
        assert len(names) == atoms and set(names) == {"CA"}
    assert coords.dtype == np.float32 and np.array_equal(coords, np.asarray(ref_xyz, dtype=np.float32))
    bb, aa = ae.channel_indices(names, res)
    assert (aa >= 4).all() and ((bb >= 0) == np.isin(names, ae.BACKBONE_ATOMS)).all()


def test_fast_npz_grid_reader_equals_numpy(tmp_path):
    """dataset.read_npz_grid (one fromfile behind the ZIP + npy headers) against np.load, incl. the fallbacks."""
    from mica_amd.dataset import read_npz_grid
    rng = np.random.default_rng(3)
    for dt, shape in ((np.float32, (64, 64, 64)), (np.float64, (5, 6, 7)), (np.int16, (3, 4, 5))):
        g = (rng.random(shape) * 100).astype(dt)
        p = str(tmp_path / f"t_{np.dtype(dt).name}.npz")
        np.savez(p, grid=g, i=1, j=2, k=3, orig_shape=shape, voxel_size=np.rec.array((1.0, 1.0, 1.0), dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')]))
        got = read_npz_grid(p)
        assert got.dtype == g.dtype and np.array_equal(got, g) and np.array_equal(got, np.load(p)['grid'])
    g = rng.random((8, 8, 8)).astype(np.float32)
    np.savez_compressed(str(tmp_path / "c.npz"), grid=g)                 # compressed member: falls back to np.load
    assert np.array_equal(read_npz_grid(str(tmp_path / "c.npz")), g)
    np.savez(str(tmp_path / "o.npz"), i=1, grid=g)                       # grid is not the first member: falls back
    assert np.array_equal(read_npz_grid(str(tmp_path / "o.npz")), g)
    np.savez(str(tmp_path / "f.npz"), grid=np.asfortranarray(rng.random((4, 5, 6))))   # Fortran order: falls back
    assert np.array_equal(read_npz_grid(str(tmp_path / "f.npz")), np.load(str(tmp_path / "f.npz"))['grid'])


def test_tile_table_random_shapes_vs_oracle():
    """mica_tile_table (host, pure integer) against the pinned tiler restatement on random shapes and tilings, including axes shorter
    than one grid step, exact multiples (the reference still emits the last window, create_grids.py:130) and 1-voxel axes."""
    from mica_amd._cabi import tile_table
    from oracle import volume_oracle as vo
    rng = np.random.default_rng(11)
    cases = [(1, 1, 1, 48), (48, 96, 144, 48), (32, 64, 33, 32), (47, 49, 95, 48)]
    cases += [tuple(int(v) for v in rng.integers(1, 140, size=3)) + (int(rng.choice([16, 32, 48])),) for _ in range(25)]
    for n0, n1, n2, grid in cases:
        pad = (64 - grid) // 2
        _, idx = vo.tile_volume(np.zeros((n0, n1, n2), np.float32), grid, pad)
        got = tile_table(n0, n1, n2, grid)
        assert np.array_equal(got, idx), (n0, n1, n2, grid)
        assert len(idx) == -(-n0 // grid) * -(-n1 // grid) * -(-n2 // grid)


# ---- round 5: multi-GPU readiness that can be proven without the node ------------------------------------------------------------
def _root_worker(rank, world, port, T, B, out_path, sleepy):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got, stats, calls = {}, {}, [0]

    def producer(first, count):
        # a rank that is late for EVERY round (sleepy): the others reach post(r + 1) - which reuses the slot of round r - 1 and
        # finishes round r - long before it has contributed round r
        calls[0] += 1
        if rank == sleepy:
            time.sleep(0.15)
        return _producer(first, count) + 1000.0 * calls[0] * 0      # records depend on the tile index only

    sharded_records(producer, lambda rec, first: got.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"),
                    stitch_rank=0, gather_to_root=True, stats=stats)
    if rank == 0:
        torch.save({"got": got, "stats": stats}, out_path)
    else:
        assert not got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,B,world,sleepy", [(13, 4, 2, 1), (29, 4, 4, 2), (45, 4, 3, 0)])
def test_gather_to_root_and_slot_reuse_with_a_late_rank(tmp_path, T, B, world, sleepy):
    """VERDICT r4 next #4 (iii) + (iv): `gather_to_root` brings the records to the stitching rank alone (`dist.gather`; no receive
    buffers on the other ranks) with the same result as the all-gather; and a rank that is late for every round cannot make a fast
    rank overwrite a send slot whose exchange is still in flight or stitch a round before it is complete: post(r) only reuses
    slot r & 1 after round r - 2 was finished (waited for) in post(r - 1)."""
    port = 29500 + (os.getpid() + 13 * T + world + 500) % 2000
    out = str(tmp_path / "root.pt")
    mp.spawn(_root_worker, args=(world, port, T, B, out, sleepy), nprocs=world, join=True)
    r = torch.load(out, weights_only=True)
    assert r["stats"]["collective"] == "gather" and r["stats"]["collectives"] == r["stats"]["rounds"]
    single = {}
    sharded_records(_producer, lambda rec, first: single.__setitem__(first, rec.clone()), T, B, (2, 4, 4, 4), torch.device("cpu"))
    assert sorted(r["got"]) == sorted(single)
    for f in single:
        assert torch.equal(r["got"][f], single[f])


def test_gather_to_root_allocates_receive_buffers_on_the_root_only():
    from mica_amd.dist import RecordExchange
    ex = RecordExchange(4, (2, 4, 4, 4), torch.device("cpu"), lambda rec, first: None, gather_to_root=True)      # no group: plain path
    assert ex.recv is None and ex.to_root and not ex.collective


def test_spawn_ranks_environment_for_eight_gpus(monkeypatch):
    """`python bench.py --gpus 8` without a launcher starts eight rank processes with RANK = LOCAL_RANK = 0..7
    (-> cuda:LOCAL_RANK in main), WORLD_SIZE = 8, one rendezvous on 127.0.0.1 and the dmabuf IPC switch RCCL needs on this pool."""
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, env))

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    args = type("A", (), {"gpus": 8})()
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(args, ["--gpus", "8", "--steps", "4"])
    assert e.value.code == 0 and len(started) == 8
    ports = {env["MASTER_PORT"] for _, env in started}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r, (cmd, env) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "8", "--steps", "4"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["LOCAL_WORLD_SIZE"]) == (str(r), str(r), "8", "8")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ---- round 5: the in-process hand-off (mica_amd/handoff.py) -------------------------------------------------------------------------
def test_npz_layout_equals_np_savez(tmp_path):
    """The tile files the background writer lays out by hand are .npz files like np.savez's (reference utils/create_grids.py:163-174):
    same member names in the same order, same dtypes, shapes and values under np.load; CRCs valid; and the fast reader of
    mica_amd/dataset.py finds the grid."""
    import zipfile
    from mica_amd.dataset import read_npz_grid
    from mica_amd.handoff import SCALAR_ORDER, NpzLayout
    voxel_size = np.rec.array((1.0, 1.5, 2.0), dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
    origin = np.rec.array((-3.0, 0.5, 7.0), dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4')])
    const = dict(orig_shape=(100, 70, 50), grid_size=48, padding=8, voxel_size=voxel_size, origin=origin, mapc=np.int32(1), mapr=np.int32(2),
                 maps=np.int32(3))
    for dt in (np.float32, np.int16):
        g = (np.random.default_rng(3).random((64, 64, 64)) * 100).astype(dt)
        sc = dict(i=48, j=0, k=96, di=2, dj=22, dk=4)
        head, body, tail = NpzLayout(dt, (64, 64, 64), const).pieces(g, sc, SCALAR_ORDER)
        a, b = str(tmp_path / "a.npz"), str(tmp_path / "b.npz")
        open(a, "wb").write(head + bytes(body) + tail)
        np.savez(b, grid=g, **sc, **const)
        A, B = np.load(a), np.load(b)
        assert A.files == B.files
        for k in A.files:
            assert A[k].dtype == B[k].dtype and A[k].shape == B[k].shape and np.array_equal(A[k], B[k]), k
        assert zipfile.ZipFile(a).testzip() is None
        assert np.array_equal(read_npz_grid(a), g)


def test_handoff_registry_validity(tmp_path):
    """A resident volume is found under the directory it was registered for - until that directory is deleted or replaced (the
    reference's nnPred removes grids_path after every map, utils/modeler.py:755), and the marker sits BESIDE the directory so that the
    directory lists exactly the reference's files; a registered file is found until somebody rewrites it."""
    from mica_amd import handoff
    handoff.clear()
    d = str(tmp_path / "grids" / "normalized_map_grids")
    e = handoff.GridEntry("map", torch.zeros(4, 5, 6), 48, 8, offset=[0.0, 0.0, 0.0])
    handoff.register_grids(d, e)
    assert os.listdir(d) == [] and handoff.lookup_grids(d + "/") is e and handoff.lookup_grids(str(tmp_path / "grids" / "x")) is None
    assert e.shape == (4, 5, 6) and os.path.dirname(e.marker) == str(tmp_path / "grids")
    import shutil
    shutil.rmtree(str(tmp_path / "grids"))
    os.makedirs(d)                                            # same path, new directory: not the one the volume belongs to
    assert handoff.lookup_grids(d) is None and handoff.lookup_grids(d) is None
    e2 = handoff.GridEntry("map", torch.zeros(4, 5, 6), 48, 8)
    handoff.register_grids(d, e2)
    os.remove(e2.marker)
    assert handoff.lookup_grids(d) is None
    # files
    p = str(tmp_path / "m.mrc")
    wrote = []

    def writer():
        open(p, "wb").write(b"x" * 10)
        wrote.append(1)
    fe = handoff.register_file(p, torch.zeros(3), header="hd", writer=writer)
    handoff.wait_file(p)
    assert wrote == [1] and fe.done.is_set() and handoff.lookup_file(p) is fe and handoff.files_under(str(tmp_path), ".mrc") == [os.path.realpath(p)]
    open(p, "wb").write(b"y" * 11)                            # rewritten by somebody else: the resident copy no longer describes it
    assert handoff.lookup_file(p) is None and handoff.files_under(str(tmp_path)) == []

    def bad():
        raise OSError("disk full")
    handoff.register_file(p, torch.zeros(3), header="hd", writer=bad)
    with pytest.raises(OSError, match="disk full"):
        handoff.wait_file(p)
    assert handoff.lookup_file(p) is None
    handoff.clear()


def test_handoff_second_write_of_a_path_joins_the_first_and_deleted_files_are_not_listed(tmp_path):
    """Advisor findings of round 5: (1) `register_file` for a path that still has a background writer (a re-run, a second map in the same
    AF3_results directory) joins the older writer before the new one starts - two writers never stream into one path - and
    `drop_file`, which DataPreprocessor calls before it removes the old file, joins it too; (2) `files_under` does not list entries
    whose file the caller has deleted since (create_AF3_encodings_grids then says "No AF3 encoding files found" like the reference,
    utils/create_grids.py:300-306)."""
    import threading
    import time
    from mica_amd import handoff
    handoff.clear()
    p = str(tmp_path / "CA_encoding.mrc")
    gate, log = threading.Event(), []

    def slow():
        gate.wait(5.0)
        open(p, "wb").write(b"first")
        log.append("first closed")

    def second():
        log.append("second started")
        open(p, "wb").write(b"second!")
    handoff.register_file(p, torch.zeros(3), header="hd", writer=slow)
    threading.Timer(0.3, gate.set).start()
    t0 = time.time()
    handoff.register_file(p, torch.ones(3), header="hd", writer=second)      # returns only when the first writer has closed its file
    assert log[0] == "first closed" and time.time() - t0 >= 0.25
    handoff.wait_file(p)
    assert log == ["first closed", "second started"] and open(p, "rb").read() == b"second!"
    fe = handoff.lookup_file(p)
    assert fe is not None and float(fe.tensor.sum()) == 3.0
    # drop_file joins as well (what preprocessing.py does before os.remove)
    gate.clear()
    handoff.register_file(p, torch.zeros(3), header="hd", writer=slow)
    threading.Timer(0.2, gate.set).start()
    handoff.drop_file(p)
    assert log[-1] == "first closed" and handoff.lookup_file(p) is None
    # a file deleted by the caller is not listed any more
    handoff.register_file(p, torch.zeros(3), header="hd", writer=second)
    handoff.wait_file(p)
    assert handoff.files_under(str(tmp_path), "_encoding.mrc") == [os.path.realpath(p)]
    os.remove(p)
    assert handoff.files_under(str(tmp_path), "_encoding.mrc") == []
    handoff.clear()


def test_file_modes_of_the_mirrors_and_of_the_solver_shim(monkeypatch):
    """The drop-in classes keep the reference's contract by default ("sync": files complete when the call returns, advisor finding
    of round 5); the environment or the Solver-flow shim selects the background writers."""
    from mica_amd import solver_mirrors
    from mica_amd.create_grids import GridCreator
    from mica_amd.preprocessing import DataPreprocessor
    monkeypatch.delenv("MICA_TILE_FILES", raising=False)
    monkeypatch.delenv("MICA_MRC_FILES", raising=False)
    g = GridCreator(quiet=True)
    assert g.write_files and g.sync_files
    assert DataPreprocessor("m.mrc", "x/AF3_results", quiet=True).write_files == "sync"
    for arg, want in ((True, (True, False)), ("background", (True, False)), ("sync", (True, True)), (False, (False, False)), ("none", (False, False))):
        g = GridCreator(quiet=True, write_files=arg)
        assert (g.write_files, g.sync_files) == want, arg
    with pytest.raises(ValueError):
        GridCreator(quiet=True, write_files="later")
    monkeypatch.setenv("MICA_TILE_FILES", "background")
    monkeypatch.setenv("MICA_MRC_FILES", "background")
    g = GridCreator(quiet=True)
    assert g.write_files and not g.sync_files and DataPreprocessor("m.mrc", "x/AF3_results", quiet=True).write_files == "background"
    assert GridCreator(quiet=True, write_files="sync").sync_files                       # an explicit argument wins
    monkeypatch.delenv("MICA_TILE_FILES")
    monkeypatch.delenv("MICA_MRC_FILES")
    s = solver_mirrors.GridCreator(quiet=True)
    assert isinstance(s, GridCreator) and s.write_files and not s.sync_files
    d = solver_mirrors.DataPreprocessor("m.mrc", "x/AF3_results", quiet=True)
    assert isinstance(d, DataPreprocessor) and d.write_files == "background"
    from mica_amd.predict import CryoEMPredictor
    assert solver_mirrors.CryoEMPredictor is CryoEMPredictor


def _fake_pool(world=2, **kw):
    from mica_amd.multi import RankPool
    return RankPool(world, tile=16, batch=2, backend="gloo", devices=[0] * world, runner="tests.multi_fake:FakeRunner", timeout_s=60, **kw)


def test_rank_pool_eight_ranks_as_on_a_node(tmp_path):
    """The pool at the node's size: rank 0 + seven workers over gloo with the stand-in runner; 30 batches over 8 ranks = 4 rounds, the last
    with idle ranks; every worker reports its start-up once and leaves with exit code 0."""
    from tests.multi_fake import FakeRunner
    ck = tmp_path / "ckpt.txt"
    ck.write_text("0.5")
    pool = _fake_pool(8).spawn()
    try:
        procs = list(pool.procs)
        assert len(procs) == 7
        vol = torch.arange(40 * 24 * 17, dtype=torch.float32).reshape(40, 24, 17)            # 5 x 3 x 3 = 45 tiles of 8^3 -> 23 batches of 2
        out = pool.predict(FakeRunner(0, 16, 2), str(ck), vol, None, grid=8, pad=0)
        assert np.array_equal(out["volume"], (vol * 0.5).numpy())
        assert sorted(s["rank"] for s in pool.startup) == list(range(1, 8)) and [s["rank"] for s in pool.last_status] == list(range(8))
        assert pool.last_status[0]["stats"]["world"] == 8 and pool.last_status[0]["stats"]["collectives"] == 3      # ceil(23 / 8) rounds
    finally:
        pool.close()
    assert [p.returncode for p in procs] == [0] * 7 and not torch.distributed.is_initialized()


def test_rank_pool_two_ranks_persistent_workers_and_clean_shutdown(tmp_path):
    """mica_amd/multi.py (BASELINE configs[2] behind the single-process call site, reference utils/modeler.py:722-738): rank 0 is the
    calling process, rank 1 a fresh child that stays for the next map.  Two maps through one pool over gloo with the stand-in runner:
    the sharded result equals the expected volume, the child is the same process for both maps, loads the 'checkpoint' once, reports its
    start-up marks, and has exited with code 0 when close() returns."""
    from tests.multi_fake import FakeRunner
    ck = tmp_path / "ckpt.txt"
    ck.write_text("3.0")
    pool = _fake_pool().spawn()
    try:
        child = pool.procs[0]
        r0 = FakeRunner(0, 16, 2)
        for n, shape in enumerate(((20, 9, 17), (33, 16, 5))):
            vol = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape) / 7.0
            af = (torch.arange(24 * int(np.prod(shape))).reshape(24, *shape) % 11 == 0).to(torch.uint8) if n else None
            out = pool.predict(r0, str(ck), vol, af, grid=8, pad=0, gather_to_root=bool(n))
            want = vol * 3.0 + (0.0 if af is None else float(af.sum()))
            assert np.array_equal(out["volume"], want.numpy())
            assert pool.procs[0] is child and child.poll() is None and pool.maps == n + 1
        assert len(pool.startup) == 1 and pool.startup[0]["rank"] == 1
        s = pool.startup[0]
        assert s["spawned"] <= s["imported"] <= s["engine"] <= s["joined"] <= s["first_map_done"]
        assert [st["rank"] for st in pool.last_status] == [0, 1] and all(st["ok"] for st in pool.last_status)
        assert "rank 1: python + torch import" in pool.startup_report()
    finally:
        pool.close()
    assert child.returncode == 0 and pool.procs == [] and not torch.distributed.is_initialized()
    pool.close()                                                            # idempotent
    with pytest.raises(Exception, match="closed"):
        pool.predict(r0, str(ck), torch.zeros(8, 8, 8), None, 8, 0)


@pytest.mark.parametrize("where", ["init", "load", "predict", "killed"])
def test_rank_pool_worker_failures_end_loudly_and_leave_no_process(tmp_path, monkeypatch, where):
    """A worker that cannot start (its runner raises), cannot load the checkpoint, fails inside the map or was killed between two maps
    takes the call down with an exception that names the rank - within seconds, not at a collective's time-out - and the pool is closed:
    every child has been joined (or killed by its exact PID) when the exception reaches the caller."""
    import time
    from mica_amd.engine import MicaHipError
    from tests.multi_fake import FakeRunner
    ck = tmp_path / "ckpt.txt"
    ck.write_text("2.0")
    if where != "killed":
        monkeypatch.setenv("MICA_FAKE_FAIL", f"{where}:1")
    pool = _fake_pool().spawn()
    child = pool.procs[0]
    r0 = FakeRunner(0, 16, 2)
    vol = torch.ones(16, 16, 16)
    t0 = time.time()
    try:
        if where == "killed":
            assert np.array_equal(pool.predict(r0, str(ck), vol, None, 8, 0)["volume"], 2.0 * vol.numpy())
            child.kill()
            child.wait()
            with pytest.raises(MicaHipError, match=r"rank\(s\) \[\(1, -9\)\] have exited"):
                pool.predict(r0, str(ck), vol, None, 8, 0)
        elif where == "init":
            with pytest.raises(MicaHipError, match="exited during start-up"):
                pool.predict(r0, str(ck), vol, None, 8, 0)
        elif where == "load":
            with pytest.raises(MicaHipError, match="rank 1: FileNotFoundError: injected"):
                pool.predict(r0, str(ck), vol, None, 8, 0)
        else:
            with pytest.raises(Exception):
                pool.predict(r0, str(ck), vol, None, 8, 0)
    finally:
        pool.close()
    assert time.time() - t0 < 45 and pool.closed and pool.procs == [] and child.poll() is not None
    assert not torch.distributed.is_initialized()


def test_solver_shim_starts_the_worker_ranks_when_getdata_begins(monkeypatch, tmp_path):
    """With MICA_GPUS > 1 the Solver-flow shim starts the worker ranks when `DataPreprocessor` is constructed - the first line of
    getData (reference utils/modeler.py:675) - so that their python + torch import and their engines come up beside the normaliser and
    the tilers; the predictor later asks for the pool of the same configuration and finds THOSE processes.  (Stand-in runner: no GPU.)"""
    from mica_amd import multi, solver_mirrors
    multi.shutdown()
    monkeypatch.setenv("MICA_GPUS", "2")
    monkeypatch.setenv("MICA_RANK_BACKEND", "gloo")
    monkeypatch.setenv("MICA_RANK_RUNNER", "tests.multi_fake:FakeRunner")
    try:
        solver_mirrors.DataPreprocessor(str(tmp_path / "emd.mrc"), str(tmp_path / "AF3_results"), quiet=True)
        pool = multi.get_pool(2, tile=64, batch=8)                   # what CryoEMPredictor.run_prediction asks for with the defaults
        assert len(pool.procs) == 1 and pool.procs[0].poll() is None and not pool.joined and pool.t_spawn is not None
        worker = pool.procs[0]
        assert multi.get_pool(2, tile=64, batch=8) is pool
        other = multi.get_pool(2, tile=64, batch=4)                  # another configuration replaces the pool: the old worker is gone first
        assert other is not pool and pool.closed and worker.poll() is not None
    finally:
        multi.shutdown()
    assert multi._POOLS == {}


def test_engine_methods_run_under_the_engines_lock():
    """A context is not thread-safe (include/mica_hip.h) and the tile-file writer shares the tiler's engine from its own thread:
    every public Engine method takes the engine's re-entrant lock (mica_amd/engine.py)."""
    import threading
    from mica_amd.engine import Engine

    public = [n for n, m in vars(Engine).items() if callable(m) and not n.startswith("_")]
    assert {"gather_tiles", "forward_tiles", "forward_records", "zoom_cubic", "close", "load_state_dict"} <= set(public)
    assert all(hasattr(getattr(Engine, n), "__wrapped__") for n in public)
    e = object.__new__(Engine)                 # no context: close() is the one method that needs nothing but the library handle
    e.call_lock = threading.RLock()
    seen = []

    class Lib:
        @staticmethod
        def mica_destroy(h):
            seen.append((h, e.call_lock._is_owned()))
    e.lib, e._h = Lib, 1234
    e.close()
    assert seen == [(1234, True)] and e._h is None and not e.call_lock._is_owned()
