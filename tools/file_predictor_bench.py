"""The reference's getData + nnPred sequence (utils/modeler.py:673-738) through the three mirrors, end to end, on one map:

    GridCreator.create_normalized_map_grids + create_AF3_encodings_grids  ->  CryoEMPredictor(grids_path).run_prediction()

timed (a) with the in-process hand-off (mica_amd/handoff.py: the volumes stay on the GPU, the tile files are written in the
background - `write_files="background"`, what the Solver-flow shim mica_amd/solver_mirrors.py selects -, complete when each wrapper
returns - "sync", the plain mirrors' default = the reference's contract - or not at all), (b) with the predictor reading the 25 .npz
files per tile (what a predictor in another process does), next to (c) the disk-free VolumePredictor on the same map.  All must agree
bit for bit.  Then the WHOLE of getData + nnPred from the raw map and a docked model: plain mirrors (every file complete when its call
returns) against the shim (every file behind the calls, all present and complete when nnPred returns: the 25 MRC files are hashed and
a sample of tile files is compared member by member with the synchronous route's).
usage: python tools/file_predictor_bench.py [n=256]"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mica_amd import handoff, mrc
from mica_amd.af3_encoding import CHANNEL_NAMES
from mica_amd.create_grids import GridCreator
from mica_amd.engine import Engine
from mica_amd.pipeline import VolumePredictor
from mica_amd.predict import CryoEMPredictor
from mica_amd.weights import synth_state_dict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 256
tmp = tempfile.mkdtemp(prefix="mica_filebench_")
try:
    vol = np.random.default_rng(1).random((n, n, n), dtype=np.float32)
    af = np.empty((24, n, n, n), np.uint8)                 # binary encodings; float32 only in the files
    rng2 = np.random.default_rng(2)
    for c in range(24):
        af[c] = rng2.random((n, n, n), dtype=np.float32) < 1e-3
    w = synth_state_dict(2022)
    mp = os.path.join(tmp, "resampled_normalized_map.mrc")
    mrc.write_mrc(mp, vol)
    os.makedirs(os.path.join(tmp, "AF3_encodings"))
    for c, name in enumerate(CHANNEL_NAMES):
        mrc.write_mrc(os.path.join(tmp, "AF3_encodings", f"{name}_encoding.mrc"), af[c].astype(np.float32))
    ck = os.path.join(tmp, "ckpt.pth")
    torch.save({"model_state_dict": {k: torch.from_numpy(v.copy()) for k, v in w.items()}}, ck)
    grids = os.path.join(tmp, "grids")
    eng = Engine(0, max_batch=1, tile_size=64)           # the tiler's own context (gather kernel for the file writer)
    results = {}

    def chain(tag, write_files, resident=True, threads=4):
        """getData's two tiler calls + nnPred, as the reference's call sites make them; then what modeler.py:755 does"""
        shutil.rmtree(grids, ignore_errors=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gc = GridCreator(quiet=True, engine=eng, write_files=write_files)
        r1 = gc.create_normalized_map_grids(normalized_map_path=mp, output_dir=os.path.join(grids, "normalized_map_grids"))
        r2 = gc.create_AF3_encodings_grids(AF3_encodings_path=os.path.join(tmp, "AF3_encodings"), output_dir=os.path.join(grids, "AF3_encoding_grids"))
        assert r1["success"] and r2["success"]
        t1 = time.perf_counter()
        pred = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=os.path.join(tmp, "out"), save_output=False, device="cuda", quiet=True)
        pred.use_resident_volumes = resident
        pred.loader_threads = threads
        ok, vols = pred.run_prediction()
        t2 = time.perf_counter()
        assert ok and (pred.resident is not None) == resident
        T = r1["grid_count"]
        nfiles = sum(len([f for f in fs if f.endswith('.npz')]) for _, _, fs in os.walk(grids))
        shutil.rmtree(grids)
        t3 = time.perf_counter()
        print(f"{tag}: tiling {t1 - t0:.2f} s + prediction {t2 - t1:.2f} s (model load {pred.timing_stats['model_loading']:.2f}, inference "
              f"{pred.timing_stats['inference']:.2f}) = {t2 - t0:.2f} s -> {T / (t2 - t0):.1f} sub-grids/s end to end ({T} tiles of the {n}^3 map; "
              f"{nfiles} tile files on disk when nnPred returned; rmtree {t3 - t2:.2f} s)", flush=True)
        if "first" not in results:
            results["first"] = vols
        else:
            for k in vols:
                assert np.array_equal(vols[k], results["first"][k]), (tag, k)
        results.setdefault("tags", []).append(tag)
        return T

    if n <= 256:
        chain("warm-up (hand-off, no files)", False)
    T = chain("hand-off, tile files written in the background (Solver-flow shim)", "background")
    chain("hand-off, tile files complete when the wrappers return (plain mirrors' default, 'sync')", "sync")
    chain("hand-off, no tile files (write_files=False)", False)
    chain("predictor reads the tile files (cold path), 4 reader threads", "sync", resident=False)
    # ---- the whole of getData + nnPred (utils/modeler.py:673-738): DataPreprocessor in front, from the raw map and a docked model ----
    from mica_amd.preprocessing import DataPreprocessor
    raw = ((vol - np.float32(0.3)) * np.float32(3.0)).astype(np.float32)
    inp = os.path.join(tmp, "input", "9999")
    os.makedirs(os.path.join(inp, "AF3_results"))
    raw_path = os.path.join(inp, "emd_9999.mrc")
    mrc.write_mrc(raw_path, raw)
    rng = np.random.default_rng(5)
    aas = ['ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU', 'MET', 'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR']
    lines = []
    nres = max(200, n * n * n // 8000)                                     # about one residue per 20^3 voxels
    for r in range(nres):
        c0 = rng.random(3) * (n - 8) + 3
        for a in ("N", "CA", "C", "O", "CB"):
            x, y, z = c0 + rng.random(3) * 2.0
            nm = " " + a.ljust(3)
            lines.append("%-6s%5d %4s%1s%3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s\n" % ("ATOM", (len(lines) + 1) % 100000, nm, " ", aas[r % 20], "A", (r + 1) % 10000, x, y, z, 1.0, 20.0, a[0]))
    pdb = os.path.join(inp, "9999_af3_docked.pdb")
    open(pdb, "w").write("".join(lines) + "END\n")

    import hashlib
    from mica_amd import solver_mirrors
    from mica_amd import create_grids as plain_cg, preprocessing as plain_pp

    def sha_file(p):
        h = hashlib.sha256()
        with open(p, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    file_facts = {}

    def full_chain(tag, DP, GC, gc_kw=None, check_files=False):
        """DP / GC: the DataPreprocessor / GridCreator classes of the route (plain mirrors or the Solver-flow shim), constructed with
        the arguments the reference's call sites pass (utils/modeler.py:675-706)"""
        shutil.rmtree(grids, ignore_errors=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dp = DP(map_path=raw_path, AF3_results=os.path.join(inp, "AF3_results"), quiet=True, engine=eng)
        dp.resample_and_normalize_map()
        t1 = time.perf_counter()
        assert dp.create_AF3_encodings(pdb) is True
        t2 = time.perf_counter()
        gc = GC(quiet=True, engine=eng, **(gc_kw or {}))
        r1 = gc.create_normalized_map_grids(normalized_map_path=dp.normalized_map_path, output_dir=os.path.join(grids, "normalized_map_grids"))
        r2 = gc.create_AF3_encodings_grids(AF3_encodings_path=os.path.join(inp, "AF3_encodings"), output_dir=os.path.join(grids, "AF3_encoding_grids"))
        assert r1["success"] and r2["success"]
        t3 = time.perf_counter()
        pred = CryoEMPredictor(model_path=ck, grids_path=grids + "/", output_path=os.path.join(tmp, "out"), save_output=False, device="cuda", quiet=True)
        ok, vols = pred.run_prediction()
        t4 = time.perf_counter()
        assert ok and pred.resident is not None
        T = r1["grid_count"]
        if check_files:
            # what is on disk when nnPred has returned (outside the clock): every MRC hashed, every tile file counted, a sample compared
            mrcs = {os.path.basename(dp.normalized_map_path): sha_file(dp.normalized_map_path)}
            for name in CHANNEL_NAMES:
                mrcs[name] = sha_file(os.path.join(inp, "AF3_encodings", f"{name}_encoding.mrc"))
            names = sorted(os.path.relpath(os.path.join(d, f), grids) for d, _, fs in os.walk(grids) for f in fs)
            assert all(f.endswith(".npz") for f in names), [f for f in names if not f.endswith(".npz")][:3]      # no temporary name left behind
            sample = names[:: max(1, len(names) // 64)]
            members = {}
            for f in sample:
                z = np.load(os.path.join(grids, f))
                members[f] = {k: (str(z[k].dtype), z[k].shape, hashlib.sha256(np.ascontiguousarray(z[k]).tobytes()).hexdigest()) for k in z.files}
            file_facts[tag] = (mrcs, len(names), members)
        shutil.rmtree(grids)
        os.remove(dp.normalized_map_path)
        shutil.rmtree(os.path.join(inp, "AF3_encodings"))                 # what utils/modeler.py:755-757 does
        print(f"{tag}: resample+normalise {t1 - t0:.2f} s + AF3 encodings ({nres * 5} atoms) {t2 - t1:.2f} s + tiling {t3 - t2:.2f} s + prediction {t4 - t3:.2f} s "
              f"= {t4 - t0:.2f} s -> {T / (t4 - t0):.1f} sub-grids/s for ALL of getData + nnPred ({T} tiles)", flush=True)
        return vols

    plain, shim = (plain_pp.DataPreprocessor, plain_cg.GridCreator), (solver_mirrors.DataPreprocessor, solver_mirrors.GridCreator)
    if n <= 256:
        full_chain("getData + nnPred, warm-up", *plain)
    va = full_chain("getData + nnPred, plain mirrors: every file complete when the call it was asked of returns (the reference's contract)", *plain, check_files=True)
    for tag, cls, kw, chk in (("getData + nnPred, Solver-flow shim (mica_amd.solver_mirrors): every file written behind the calls, all complete when nnPred returns", shim, None, True),
                              ("getData + nnPred, shim, no tile files (GridCreator(write_files=False)), MRC files in the background", shim, dict(write_files=False), False)):
        vb = full_chain(tag, *cls, gc_kw=kw, check_files=chk)
        for k in va:
            assert np.array_equal(va[k], vb[k]), k
        del vb
    del va
    (m_a, n_a, mem_a), (m_b, n_b, mem_b) = list(file_facts.values())
    assert m_a == m_b and n_a == n_b == 25 * T and mem_a == mem_b
    print(f"the three full chains agree bit for bit; plain and shim routes left the same files when nnPred returned: 25 MRC files byte-identical "
          f"(sha-256), {n_b} tile files each, {len(mem_b)} sampled tile files equal member by member (keys, dtypes, shapes, bytes)")
    vp = VolumePredictor(Engine(0, max_batch=8, tile_size=64), 48, 8, 8)
    vp.e.load_state_dict(w)
    dv, da = torch.from_numpy(np.ascontiguousarray(vol.transpose(2, 1, 0))).cuda(), torch.from_numpy(np.ascontiguousarray(af.transpose(0, 3, 2, 1))).cuda()
    vp.predict_volume(dv, da); torch.cuda.synchronize()
    t0 = time.perf_counter(); mem = vp.predict_volume(dv, da); torch.cuda.synchronize(); dm = time.perf_counter() - t0
    print(f"disk-free VolumePredictor on the same map, volumes left on the GPU: {dm:.2f} s -> {T / dm:.1f} sub-grids/s")
    for k in results["first"]:
        assert np.array_equal(results["first"][k], mem[k].cpu().numpy()), k
    print("every route (%s) == disk-free volumes: bit-identical" % "; ".join(results["tags"]))
finally:
    handoff.clear()
    shutil.rmtree(tmp, ignore_errors=True)
