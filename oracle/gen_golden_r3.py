"""Round-3 goldens: the reference's OWN tiler, normaliser and AF3 rasteriser, run unmodified (build container only).

Run:  python oracle/gen_golden_r3.py [--check] [tiler] [normaliser] [af3] [cluster]   (needs /root/reference; about a minute)
      --check regenerates into a scratch directory and compares bit for bit with tests/golden/ (oracle/_check.py);
      tests/test_cpu_oracle.py::test_golden_generator_r3_check runs it whenever /root/reference is present.

TEST INFRASTRUCTURE.  utils/create_grids.py, utils/preprocessing.py and the training tilers import two
packages this image lacks, `mrcfile` and `Bio`.  Both are I/O only on this path, so this script
registers adapters for them in `sys.modules` and then imports the reference modules as they are:

* `mrcfile`  -> an adapter over this repo's own MRC2014 reader/writer (mica_amd/mrc.py): `open()` yields
  `.data`, `.voxel_size` and `.header.{origin,mapc,mapr,maps,n[xyz]start}`; `new()` collects `set_data`,
  `voxel_size` and the header fields and writes a real MRC file on exit.  No arithmetic.
* `Bio.PDB`  -> `PDBParser.get_structure` hands back the atoms of a JSON atom list as
  model/chain/residue/atom containers (get_id, get_resname, get_name, get_coord); `PDBIO` is an empty
  class.  No arithmetic either: everything that computes (transpose, pad, window loop, zoom, median,
  percentile, clip, the coordinate transform and scatter) is the reference's code, executed unmodified.

Outputs (tests/golden/):  tiler_ref.json, normaliser_ref.json, normaliser_ref_*.npy, af3_ref.json, cluster_ref.json
(Solver.clustering behind a scikit-learn DBSCAN standing in for open3d's - see install_adapters).
Every case also asserts that oracle/volume_oracle.py and oracle/af3_oracle.py reproduce the reference
bit for bit, which is what pins those restatements.  Only data travels; no reference source.
"""
from __future__ import annotations

import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"

from mica_amd import mrc as _mrc                          # noqa: E402
from mica_amd.synth import synth_density                  # noqa: E402
from oracle import volume_oracle as vo                    # noqa: E402
from oracle import af3_oracle as ao                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
_XYZ = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ----------------------------------------------------------------------------------------------
# I/O adapters for the two absent packages
# ----------------------------------------------------------------------------------------------
class _Header:
    pass


class _MrcAdapter:
    """What the reference touches of mrcfile.MrcFile (create_grids.py:108-117, preprocessing.py:98-107,138-148,
    196-206, 241-251)."""

    def __init__(self, path, mode):
        self._path, self._mode = path, mode
        self.header = _Header()
        if mode == "r":
            data, hd = _mrc.read_mrc(path)
            self.data = data
            self._voxel = np.rec.array(tuple(np.float32(v) for v in hd.voxel_size), dtype=_XYZ)
            self.header.origin = np.rec.array(tuple(np.float32(v) for v in hd.origin), dtype=_XYZ)
            for k in ("mapc", "mapr", "maps", "nxstart", "nystart", "nzstart"):
                setattr(self.header, k, np.int32(getattr(hd, k)))
        else:
            self.data = None
            self._voxel = np.rec.array((0.0, 0.0, 0.0), dtype=_XYZ)
            self.header.origin = np.rec.array((0.0, 0.0, 0.0), dtype=_XYZ)
            self.header.mapc, self.header.mapr, self.header.maps = 1, 2, 3
            self.header.nxstart = self.header.nystart = self.header.nzstart = 0

    @property
    def voxel_size(self):
        return self._voxel

    @voxel_size.setter
    def voxel_size(self, v):
        v = (v, v, v) if np.isscalar(v) else tuple(v)
        self._voxel = np.rec.array(tuple(np.float32(a) for a in v), dtype=_XYZ)

    def set_data(self, data):
        self.data = np.asarray(data)

    def update_header_stats(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._mode == "w" and exc[0] is None and self.data is not None:
            o = self.header.origin
            _mrc.write_mrc(self._path, self.data, voxel_size=(self._voxel.x, self._voxel.y, self._voxel.z),
                           origin=(float(o.x), float(o.y), float(o.z)), mapc=int(self.header.mapc),
                           mapr=int(self.header.mapr), maps=int(self.header.maps), nxstart=int(self.header.nxstart),
                           nystart=int(self.header.nystart), nzstart=int(self.header.nzstart))
        return False


class _Atom:
    def __init__(self, name, xyz):
        self._n, self._c = name, np.asarray(xyz, dtype=np.float32)        # Bio.PDB stores float32 coordinates

    def get_name(self):
        return self._n

    def get_coord(self):
        return self._c


class _Residue(list):
    def __init__(self, resname, hetflag, atoms):
        super().__init__(atoms)
        self._r, self._h = resname, hetflag

    def get_id(self):
        return (self._h, 0, " ")

    def get_resname(self):
        return self._r


class _Parser:
    """get_structure(name, path): `path` is a JSON file [[chain, resname, hetflag, [[atom, x, y, z], ...]], ...]."""

    def __init__(self, QUIET=True):
        pass

    def get_structure(self, _name, path):
        chains = {}
        for chain, resname, het, atoms in json.load(open(path)):
            chains.setdefault(chain, []).append(_Residue(resname, het, [_Atom(a[0], a[1:4]) for a in atoms]))
        return [list(chains.values())]                                      # one model -> chains -> residues -> atoms


class _PDBIO:
    pass


def install_adapters():
    m = types.ModuleType("mrcfile")
    m.open = lambda path, mode="r", **kw: _MrcAdapter(path, "r")
    m.new = lambda path, overwrite=False, **kw: _MrcAdapter(path, "w")
    sys.modules["mrcfile"] = m
    bio, pdb = types.ModuleType("Bio"), types.ModuleType("Bio.PDB")
    pdb.PDBParser, pdb.PDBIO = _Parser, _PDBIO
    bio.PDB = pdb
    sys.modules["Bio"], sys.modules["Bio.PDB"] = bio, pdb
    # utils/modeler.py (Solver.clustering) additionally imports open3d (only for DBSCAN, :768-770), superpose3d and three more
    # Bio.PDB modules it does not use on this path.  DBSCAN is outside the scope of this repo (the caller's step): scikit-learn's
    # DBSCAN, which IS installed, labels the points instead.  Any labelling serves: what the golden pins is the reference's
    # arithmetic AFTER it (cluster scores, sorted greedy NMS, refinement, distances, neighbour matrix), run on these labels.
    o3d, geo, util = types.ModuleType("open3d"), types.ModuleType("open3d.geometry"), types.ModuleType("open3d.utility")

    class PointCloud:
        points = None

        def cluster_dbscan(self, eps, min_points):
            from sklearn.cluster import DBSCAN
            return DBSCAN(eps=eps, min_samples=min_points).fit(np.asarray(self.points, dtype=np.float64)).labels_.tolist()

    geo.PointCloud, util.Vector3dVector = PointCloud, (lambda a: np.asarray(a))
    o3d.geometry, o3d.utility = geo, util
    sys.modules["open3d"], sys.modules["open3d.geometry"], sys.modules["open3d.utility"] = o3d, geo, util
    sys.modules["superpose3d"] = types.ModuleType("superpose3d")
    for name, cls in (("PDBParser", "PDBParser"), ("Structure", "Structure"), ("Model", "Model")):
        m2 = types.ModuleType("Bio.PDB." + name)
        setattr(m2, cls, _Parser if cls == "PDBParser" else type(cls, (), {}))
        sys.modules["Bio.PDB." + name] = m2
        if name != "PDBParser":          # `from Bio import PDB; PDB.PDBParser(...)` (preprocessing.py:52) must stay the class
            setattr(pdb, name, m2)
    if REF not in sys.path:
        sys.path.insert(0, REF)


# ----------------------------------------------------------------------------------------------
def _read_tiles(gdir, prefix):
    """The reference's npz tiles of one directory in its own creation order (lexicographic i, j, k)."""
    recs = []
    for f in os.listdir(gdir):
        z = np.load(os.path.join(gdir, f))
        assert f == f"{prefix}_i{int(z['i'])}_j{int(z['j'])}_k{int(z['k'])}.npz", f
        recs.append(z)
    recs.sort(key=lambda z: (int(z["i"]), int(z["j"]), int(z["k"])))
    return recs


TILER_CASES = [((50, 70, 100), 51), ((96, 96, 96), 52), ((100, 100, 100), 53), ((7, 130, 48), 54)]
AXES = [(1, 2, 3), (3, 2, 1), (2, 1, 3), (2, 3, 1)]
TILINGS = [(48, 8), (32, 16)]


def tiler_goldens(tmp):
    from utils.create_grids import GridCreator
    gc = GridCreator(quiet=True)
    out = {"cases": []}
    for shape, seed in TILER_CASES:
        for axes in AXES:
            for grid, pad in TILINGS:
                if axes not in ((1, 2, 3), (3, 2, 1)) and shape != (50, 70, 100):
                    continue                                   # every axis order on the ragged shape; two on the others
                vol = synth_density(shape, seed)               # stored [section, row, column]
                starts = (5, -6, 7)                            # nxstart, nystart, nzstart
                path = os.path.join(tmp, "in.mrc")
                _mrc.write_mrc(path, vol, voxel_size=(1.0, 1.0, 1.0), origin=(1.5, -2.0, 3.25), mapc=axes[0], mapr=axes[1],
                               maps=axes[2], nxstart=starts[0], nystart=starts[1], nzstart=starts[2])
                gdir = os.path.join(tmp, "g")
                shutil.rmtree(gdir, ignore_errors=True)
                with redirect_stdout(io.StringIO()):
                    count, offset = gc.create_grids_from_mrc(path, gdir, grid_size=grid, padding=pad, file_prefix="pfx")
                recs = _read_tiles(gdir, "pfx")
                assert count == len(recs) > 0
                idx = [[int(z[k]) for k in ("i", "j", "k", "di", "dj", "dk")] for z in recs]
                tiles = np.stack([z["grid"] for z in recs])
                z0 = recs[0]
                meta = {"orig_shape": [int(v) for v in z0["orig_shape"]], "grid_size": int(z0["grid_size"]),
                        "padding": int(z0["padding"]), "voxel_size": [float(z0["voxel_size"][()][k]) for k in "xyz"],
                        "origin": [float(z0["origin"][()][k]) for k in "xyz"], "mapc": int(z0["mapc"]),
                        "mapr": int(z0["mapr"]), "maps": int(z0["maps"]), "grid_dtype": str(z0["grid"].dtype),
                        "keys": sorted(z0.files)}
                # the restatement must equal the reference, bit for bit
                tv, toff = vo.transpose_axes(vol, axes[0], axes[1], axes[2], [starts[2], starts[1], starts[0]])
                otiles, oidx = vo.tile_volume(tv, grid, pad)
                assert oidx.tolist() == idx and np.array_equal(otiles, tiles) and list(toff) == list(offset), (shape, axes)
                out["cases"].append({"shape": list(shape), "seed": seed, "axes": list(axes), "starts_xyz": list(starts),
                                     "grid": grid, "pad": pad, "count": int(count), "offset": [float(o) for o in offset],
                                     "idx": idx, "tile_sha256": [sha(t) for t in tiles], "tiles_sha256": sha(tiles),
                                     "meta": meta})
                print("tiler", shape, axes, (grid, pad), count, offset, flush=True)

    # wrappers: create_normalized_map_grids / create_AF3_encodings_grids result dicts and directory layout
    shape = (60, 40, 52)
    vol = synth_density(shape, 55)
    path = os.path.join(tmp, "resampled_normalized_map.mrc")
    _mrc.write_mrc(path, vol, nxstart=3, nystart=4, nzstart=5)
    gdir = os.path.join(tmp, "grids", "normalized_map_grids")
    with redirect_stdout(io.StringIO()):
        res = gc.create_normalized_map_grids(path, gdir)
    recs = _read_tiles(gdir, "normalized_map_grid")
    tv, _ = vo.transpose_axes(vol, 1, 2, 3, [5, 4, 3])
    otiles, oidx = vo.tile_volume(tv, 48, 8)
    assert np.array_equal(otiles, np.stack([z["grid"] for z in recs]))
    with redirect_stdout(io.StringIO()):
        miss = gc.create_normalized_map_grids(os.path.join(tmp, "nope.mrc"), gdir)
    out["normalized_map_grids"] = {"shape": list(shape), "seed": 55, "starts_xyz": [3, 4, 5],
                                   "result": {k: res[k] for k in ("success", "grid_count", "offset")},
                                   "result_keys": sorted(res), "missing_result_keys": sorted(miss),
                                   "missing_success": miss["success"], "files": sorted(os.listdir(gdir)),
                                   "tiles_sha256": sha(otiles)}
    edir = os.path.join(tmp, "AF3_encodings")
    os.makedirs(edir)
    chans = ["CA", "N", "ALA", "TYR"]
    enc = (synth_density((len(chans), *shape), 56) < 0.01).astype(np.float32)
    for c, name in enumerate(chans):
        _mrc.write_mrc(os.path.join(edir, f"{name}_encoding.mrc"), enc[c])
    adir = os.path.join(tmp, "grids", "AF3_encoding_grids")
    with redirect_stdout(io.StringIO()):
        res = gc.create_AF3_encodings_grids(edir, adir, parallel=False)
    layout = {}
    for c, name in enumerate(chans):
        recs = _read_tiles(os.path.join(adir, f"{name}_grids"), f"{name}_grid")
        tv, _ = vo.transpose_axes(enc[c], 1, 2, 3, [0, 0, 0])
        ot, _ = vo.tile_volume(tv, 48, 8)
        assert np.array_equal(ot, np.stack([z["grid"] for z in recs]))
        layout[name] = {"files": sorted(os.listdir(os.path.join(adir, f"{name}_grids"))), "tiles_sha256": sha(ot)}
    out["AF3_encoding_grids"] = {"shape": list(shape), "seed": 56, "channels": chans, "threshold": 0.01,
                                 "result": {k: res[k] for k in ("success", "successful_channels", "failed_channels",
                                                                "total_channels", "total_grids")},
                                 "result_keys": sorted(res), "dirs": sorted(os.listdir(adir)), "layout": layout}

    # training tiler (second witness, no transpose, skips tiles whose max < 0.01)
    sys.path.insert(0, os.path.join(REF, "scripts_for_training_data"))
    import create_grids_for_normalized_map as tt
    shape = (100, 50, 60)
    vol = synth_density(shape, 57)
    vol[:60, :, :] *= 0.009                                    # tiles made only of this slab are skipped
    path = os.path.join(tmp, "train.mrc")
    _mrc.write_mrc(path, vol, mapc=3, mapr=2, maps=1)          # axis order must be ignored by this tiler
    tr = {}
    for grid, pad in TILINGS:
        gdir = os.path.join(tmp, f"train_{grid}")
        n = tt.create_and_save_grids(path, gdir, grid_size=grid, padding=pad)
        recs = _read_tiles(gdir, "grid")
        assert n == len(recs)
        otiles, oidx = vo.tile_volume(vol, grid, pad)
        keep = [t for t in range(len(oidx)) if otiles[t].max() >= 0.01]
        assert len(keep) < len(oidx) and [oidx[t].tolist() for t in keep] == \
            [[int(z[k]) for k in ("i", "j", "k", "di", "dj", "dk")] for z in recs]
        assert np.array_equal(otiles[keep], np.stack([z["grid"] for z in recs]))
        tr[f"{grid}_{pad}"] = {"count": n, "all": len(oidx), "files": sorted(os.listdir(gdir)), "tiles_sha256": sha(otiles[keep])}
    out["training_tiler"] = {"shape": list(shape), "seed": 57, "slab": [60, 0.009], "axes": [3, 2, 1], "tilings": tr}
    json.dump(out, open(os.path.join(OUT, "tiler_ref.json"), "w"))
    print("tiler cases:", len(out["cases"]), flush=True)


# ----------------------------------------------------------------------------------------------
def _norm_inputs():
    """name -> (array as stored in the MRC, voxel size)."""
    c = {}
    c["f32_40"] = ((synth_density((40, 40, 40), 61) - 0.3) * 3.0, (1.0, 1.0, 1.0))
    c["f32_64"] = ((synth_density((64, 64, 64), 62) - 0.3) * 3.0, (1.0, 1.0, 1.0))
    c["f32_aniso_20x24x28"] = (synth_density((20, 24, 28), 63) - 0.3, (1.5, 1.25, 0.8))
    v = synth_density((31, 33, 35), 65) - 0.5                 # odd element count
    c["f32_odd_31x33x35"] = (v, (1.0, 1.0, 1.0))
    v = synth_density((16, 16, 16), 64) - 0.3
    v[1, 2, 3] = np.nan                                       # the spline prefilter spreads the NaN over the volume
    c["f32_nan_16"] = (v, (1.0, 1.0, 1.0))
    c["f32_allneg_12"] = (-synth_density((12, 12, 12), 66) - 1.0, (1.0, 1.0, 1.0))
    c["f32_const_12"] = (np.full((12, 12, 12), 0.25, np.float32), (1.0, 1.0, 1.0))
    u = synth_density((24, 20, 28), 67)
    c["i8_24x20x28"] = (np.round((u - 0.4) * 200).astype(np.int8), (1.0, 1.0, 1.0))
    c["i16_24x20x28"] = (np.round((u - 0.4) * 30000).astype(np.int16), (1.0, 1.0, 1.0))
    c["u16_24x20x28"] = (np.round(u * 60000).astype(np.uint16), (1.0, 1.0, 1.0))
    c["i16_aniso_18x20x22"] = (np.round((synth_density((18, 20, 22), 68) - 0.4) * 30000).astype(np.int16), (1.3, 0.9, 1.1))
    c["u16_aniso_18x20x22"] = (np.round(synth_density((18, 20, 22), 69) * 60000).astype(np.uint16), (1.3, 0.9, 1.1))
    c["i8_aniso_18x20x22"] = (np.round((synth_density((18, 20, 22), 70) - 0.4) * 200).astype(np.int8), (1.3, 0.9, 1.1))
    c["f16_16"] = ((synth_density((16, 16, 16), 71) - 0.3).astype(np.float16), (1.0, 1.0, 1.0))
    return c


def normaliser_goldens(tmp):
    from utils.preprocessing import DataPreprocessor
    out = {"cases": {}}
    for name, (vol, voxel) in _norm_inputs().items():
        d = os.path.join(tmp, "n_" + name)
        os.makedirs(os.path.join(d, "AF3_results"))
        src = os.path.join(d, "map.mrc")
        _mrc.write_mrc(src, vol, voxel_size=voxel, origin=(4.0, 5.0, 6.0), nxstart=1, nystart=2, nzstart=3)
        pp = DataPreprocessor(src, os.path.join(d, "AF3_results"), quiet=True)
        pp.logger.handlers.clear()
        with redirect_stdout(io.StringIO()):
            pp.resample_and_normalize_map()
        dst = os.path.join(d, "resampled_normalized_map.mrc")
        rec = {"dtype": str(vol.dtype), "shape": list(vol.shape), "voxel": list(voxel), "written": os.path.exists(dst)}
        # the restatement, on the array as mrcfile hands it over (the stored dtype)
        try:
            o, med, pct = vo.normalise_map(vol, voxel_size=voxel)
            oracle_ok = True
        except Exception as e:                               # the reference logs and writes nothing
            oracle_ok, rec["oracle_error"] = False, f"{type(e).__name__}: {e}"
        assert oracle_ok == rec["written"], (name, rec)
        if rec["written"]:
            got, hd = _mrc.read_mrc(dst)
            assert got.dtype == np.float32 and np.array_equal(got, o), name
            assert pp.normalized_map_path == dst
            rec.update({"out_shape": list(got.shape), "sha256": sha(got), "median": med, "percentile": pct,
                        "header": {"voxel": list(hd.voxel_size), "origin": list(hd.origin), "starts_xyz": [hd.nxstart, hd.nystart, hd.nzstart],
                                   "axes": [hd.mapc, hd.mapr, hd.maps]}})
            np.save(os.path.join(OUT, f"normaliser_ref_{name}.npy"), got[::3, ::3, ::3].copy())
        out["cases"][name] = rec
        print("normaliser", name, {k: rec[k] for k in rec if k not in ("header",)}, flush=True)
    import scipy
    out["versions"] = {"numpy": np.__version__, "scipy": scipy.__version__}
    json.dump(out, open(os.path.join(OUT, "normaliser_ref.json"), "w"), indent=1)


# ----------------------------------------------------------------------------------------------
def af3_atoms(seed, n_res, shape_zyx, origin, box=None):
    """A synthetic atom list around the map box (or inside the cube `box`), some atoms outside it (clipped by the
    reference).  Coordinates have three decimals, so a PDB text round trip is exact."""
    aas = ao.AMINO_ACIDS + ["UNK", "MSE"]
    u = synth_density((n_res, 8, 4), seed)
    res = []
    for r in range(n_res):
        name = aas[int(u[r, 0, 3] * len(aas)) % len(aas)]
        het = " " if u[r, 1, 3] < 0.9 else "H_" + name
        atoms = []
        for a, an in enumerate(["N", "CA", "C", "O", "CB", "CG", "OXT", "CD"][: 3 + int(u[r, 2, 3] * 6)]):
            lim = np.array([shape_zyx[2], shape_zyx[1], shape_zyx[0]], np.float64)      # x, y, z extents
            if box is not None:
                xyz = u[r, a, :3].astype(np.float64) * box + np.array(origin)
            else:
                xyz = (u[r, a, :3].astype(np.float64) * 1.3 - 0.15) * lim + np.array(origin)
            if a == 1 and r % 7 == 0:
                xyz = np.floor(xyz) + 0.5 + np.array(origin) - np.floor(np.array(origin))  # round-half-even ties
            atoms.append([an] + [float(np.float32(v)) for v in np.round(xyz, 3)])
        res.append(["AB"[r % 2], name, het, atoms])
    return res


def af3_goldens(tmp):
    from utils.preprocessing import DataPreprocessor
    out = {"cases": []}
    for shape, seed, origin, n_res, box in (((20, 24, 28), 81, (0.0, 0.0, 0.0), 100, None),
                                            ((16, 16, 16), 82, (-3.5, 2.25, 7.0), 200, None),
                                            ((12, 30, 18), 83, (10.0, -4.0, 0.5), 100, None),
                                            ((20, 24, 28), 84, (2.0, -1.5, 0.25), 200, 18.4),     # x clipped at nz-1 = 19 of nx = 28
                                            ((32, 32, 32), 85, (0.0, 0.0, 0.0), 400, None)):
        d = os.path.join(tmp, f"af_{seed}")
        os.makedirs(os.path.join(d, "AF3_results"))
        mp_ = os.path.join(d, "resampled_normalized_map.mrc")
        _mrc.write_mrc(mp_, synth_density(shape, seed), origin=origin, nxstart=1, nystart=2, nzstart=3)
        atoms = af3_atoms(seed, n_res, shape, origin, box)
        ap = os.path.join(d, "atoms.json")
        json.dump(atoms, open(ap, "w"))
        pp = DataPreprocessor(mp_, os.path.join(d, "AF3_results"), normalized_map_path=mp_, quiet=True)
        pp.logger.handlers.clear()
        with redirect_stdout(io.StringIO()) as buf:
            ok = pp.create_AF3_encodings(ap)
        rec = {"shape": list(shape), "seed": seed, "origin": list(origin), "n_res": n_res, "box": box, "success": bool(ok)}
        # restatement input: flat atom table as the product's PDB reader would deliver it
        flat = [(a[0], r[1], a[1], a[2], a[3]) for r in atoms if r[2] == " " for a in r[3]]
        try:
            want = ao.rasterise_atoms(np.array([f[2:] for f in flat], np.float32), [f[0] for f in flat],
                                      [f[1] for f in flat], origin, shape)
        except IndexError:
            want = None
        assert (want is not None) == bool(ok), (seed, buf.getvalue()[-300:])
        if ok:
            edir = os.path.join(d, "AF3_encodings")
            files = sorted(os.listdir(edir))
            assert files == sorted(f"{c}_encoding.mrc" for c in ao.CHANNEL_NAMES)
            got = np.stack([_mrc.read_mrc(os.path.join(edir, f"{c}_encoding.mrc"))[0] for c in ao.CHANNEL_NAMES])
            assert got.dtype == np.float32 and np.array_equal(got, want.astype(np.float32)), seed
            rec.update({"sha256": sha(got), "ones": int(got.sum()), "per_channel": [int(v) for v in got.sum(axis=(1, 2, 3))],
                        "n_atoms": len(flat)})
        rec["atoms"] = atoms
        out["cases"].append(rec)
        print("af3", shape, seed, {k: rec[k] for k in rec if k != "atoms"}, flush=True)
    json.dump(out, open(os.path.join(OUT, "af3_ref.json"), "w"))


def cluster_volumes(shape, seed):
    """CAProb / BBProb / AAProb / AAPred volumes made of integer-hash uniforms and exact arithmetic only (comparisons, products,
    one correctly rounded division): identical on every host.  Dense boxes of candidates (one of them on the x = 0 face, one with
    weak backbone density, one tiny) over a quiet background."""
    u = synth_density((4, *shape), seed)
    boxes = [((6, 8, 6), (18, 20, 22)), ((26, 4, 10), (38, 14, 24)), ((0, 24, 28), (7, 33, 40)), ((30, 24, 30), (36, 30, 36)), ((20, 30, 4), (22, 32, 6))]
    mask = np.zeros(shape, np.float32)
    strong = np.zeros(shape, np.float32)
    for n, (lo, hi) in enumerate(boxes):
        sl = tuple(slice(a, b) for a, b in zip(lo, hi))
        mask[sl] = 1.0
        strong[sl] = (1.0, 0.9, 0.8, 0.35, 1.0)[n]
    ca = (u[0] * (np.float32(0.25) + np.float32(0.75) * mask)).astype(np.float32)
    bb = (u[1] * (np.float32(0.1) + np.float32(0.9) * strong)).astype(np.float32)
    aa = synth_density((20, *shape), seed + 1) + np.float32(0.05)
    aa = (aa / aa.sum(axis=0, keepdims=True)).astype(np.float32)
    aapred = np.argmax(aa, axis=0).astype(np.float32)
    return ca, bb, aa, aapred


def cluster_goldens(_tmp):
    import logging
    import utils.modeler as M
    from oracle import cluster_oracle as co
    out = {"cases": []}
    for shape, seed, eps, minpts, thr, radius in (((40, 36, 44), 91, 3, 10, 0.3, 9), ((40, 36, 44), 92, 2, 6, 0.45, 5)):
        ca, bb, aa, aapred = cluster_volumes(shape, seed)
        me = types.SimpleNamespace(logger=logging.getLogger("mica_golden_cluster"), cluster_eps=eps, cluster_min_points=minpts,
                                   modeling_config=types.SimpleNamespace(CA_score_thrh=thr), CAProb=ca, AAPred=aapred, nms_radius=radius,
                                   neighbors2to6=[], neighbors0to6=[], neighbors0to7=[], neighbors2to7=[])
        M.NNPred.BBProb, M.NNPred.AAProb = bb, aa
        M.Solver.clustering(me)                             # the reference method, unmodified, on a duck-typed self
        # the same labels for the restatement: re-run the stand-in DBSCAN exactly as the method did
        pts = co.threshold_points(ca, thr)
        pc = sys.modules["open3d"].geometry.PointCloud()
        pc.points = pts
        labels = np.array(pc.cluster_dbscan(eps=eps, min_points=minpts))
        sums, avgs, val = co.cluster_scores(bb, pts, labels)
        pred = co.sorted_pred_list(ca, pts, val)
        cands = np.array(co.nms(pred.copy(), thr, radius))
        newc, newa, kept = co.refine_candidates(ca, aa, cands)
        dis, lists, mat = co.neighbour_matrix(newc, bb)
        assert len(kept) < len(cands), "a candidate on the volume face must be skipped"
        assert np.array_equal(newc, me.CA_cands) and np.array_equal(newa.T, me.CA_cands_AAProb)
        rc = np.round(newc).astype(int)
        assert np.array_equal(co.gather(aapred, rc), me.CA_cands_AA)
        assert np.array_equal(dis, me.cand_self_dis) and np.array_equal(mat, me.neigh_mat)
        for mine, ref in zip(lists, (me.neighbors2to6, me.neighbors0to6, me.neighbors0to7, me.neighbors2to7)):
            assert len(mine) == len(ref) and all(np.array_equal(a, b) for a, b in zip(mine, ref))
        rec = {"shape": list(shape), "seed": seed, "eps": eps, "min_points": minpts, "thr": thr, "nms_radius": radius,
               "n_points": int(len(pts)), "labels": labels.tolist(), "n_clusters": int(labels.max() + 1),
               "scores_sum": [float(v) for v in sums], "scores_avg": [float(v) for v in avgs], "n_valid": int(val.sum()),
               "nms_cands": cands.tolist(), "kept": kept.tolist(), "CA_cands": me.CA_cands.tolist(),
               "CA_cands_AAProb_sha256": sha(me.CA_cands_AAProb), "CA_cands_AA": [float(v) for v in me.CA_cands_AA],
               "cand_self_dis_sha256": sha(me.cand_self_dis), "neigh_mat_sha256": sha(me.neigh_mat),
               "neigh_mat_nonzero": int((me.neigh_mat != 0).sum()), "best_neigh": [[int(v) for v in b] for b in me.best_neigh],
               "numpy": np.__version__}
        out["cases"].append(rec)
        print("cluster", shape, seed, {k: rec[k] for k in ("n_points", "n_clusters", "n_valid", "neigh_mat_nonzero")}, "cands", len(cands), "kept", len(kept), flush=True)
    json.dump(out, open(os.path.join(OUT, "cluster_ref.json"), "w"))


def main(argv=None):
    from oracle._check import CheckRun
    install_adapters()
    tmp = tempfile.mkdtemp(prefix="mica_golden_r3_")
    try:
        with CheckRun(globals(), sys.argv[1:] if argv is None else argv, exact=True) as chk:
            which = chk.argv or ["tiler", "normaliser", "af3", "cluster"]
            if "tiler" in which:
                tiler_goldens(tmp)
            if "normaliser" in which:
                normaliser_goldens(tmp)
            if "af3" in which:
                af3_goldens(tmp)
            if "cluster" in which:
                cluster_goldens(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
