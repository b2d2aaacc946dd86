"""Headline benchmark: 64^3 sub-grids/s of the MICA hot path (tile gather -> network forward ->
softmax/argmax -> stitch) on a synthetic 512^3 map resident in HBM (BASELINE.json's metric config).
One JSON line on rank 0.

  python bench.py [--gpus 1] --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no launcher: this process touches no GPU and starts the N ranks itself)
  python bench.py --gpus 2 --backend gloo --single-device      (rehearse the N>1 path on one GPU)
  python bench.py --gpus 1 --backend nccl --force-exchange     (one rank, RCCL all-gather per step: the production branch on one GPU)
  python bench.py --gpus N --strong                            (strong scaling: ONE complete map sharded over N ranks, stitch + D2H timed)

A step = one batch of `--batch` tiles per GPU through the whole path.  The map is the synthetic 512^3 map at
every N (it fits one GPU: 0.54 GB map + 12.9 GB encodings + 12.3 GB output volumes + ~40 GB workspace);
"stride-32" tiling = grid 32 + 2*16 halo -> 4096 windows of 64^3.  N>1: tile batches are dealt round-robin to the
ranks (mica_amd/dist.py: the code `predict_volume_sharded` ships) and the cropped per-tile records are all-gathered over
RCCL/xGMI so that rank 0 stitches the volumes.  At N=1 the rate at the reference's default tiling (48, 8) -> 1331
windows of the same 64^3 size is measured too (`alt_tiling`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool (already exported there)

FLOPS_PER_TILE_AF = 7.3625e12       # BASELINE.md section 3 (AF path), measured on the reference model
PEAK_F16_MFMA_TF = 2500.0           # MI355X dense f16 MFMA (MI355X_MICROARCH.md)
PEAK_SPLIT_TF = PEAK_F16_MFMA_TF / 3.0   # this path spends three f16 MFMAs per f32-grade product
WINO_MFMA_PER_ALGORITHMIC = 3.0 * (14.0 / 13.5) / 1.5      # executed f16 MFMA flops per algorithmic flop of a 3^3 conv, F(2,3)-x kernel
WINO43_MFMA_PER_ALGORITHMIC = 3.0 * (14.0 / 13.5) / 2.0    # ... of the F(4,3)-x kernel (6 positions per 4 outputs: 13.5 MFMA-taps per output)


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    return max(1, min(n, 16))       # a one-GPU box's CPU share is 16


def cpu_baseline(weights, tiles_map, tiles_af, vol_host, keep=None):
    """The CPU oracle (torch CPU fp32 = the reference CPU path's arithmetic; validated bit-exact against the reference
    module in the build container) timed on this host per BASELINE.md section 4: one warm-up tile, then timed tiles of
    the same synthetic map, AF path, batch 1, at all usable cores and at 8 threads; plus the numpy tiler / stitch /
    normaliser on the same map.  `keep` (a dict) receives the oracle's logits of the timed tiles {tile index: (bb, ca, aa)}:
    the checker's outputs, compared with the GPU's records of the same tiles outside this function (`parity`)."""
    import numpy as np
    import torch
    from oracle import model_oracle as mo   # cpu_baseline leg only
    from oracle import volume_oracle as vo

    def run(threads, idxs, keep=None):
        torch.set_num_threads(threads)
        ts = []
        for i in idxs:
            t0 = time.perf_counter()
            o = mo.mica_forward(weights, tiles_map[i:i + 1], tiles_af[i:i + 1])
            ts.append(time.perf_counter() - t0)
            if keep is not None:
                keep[i] = tuple(t.numpy() for t in o)
        return ts

    k = host_threads()
    run(k, [0])                                          # warm-up (untimed)
    full = run(k, [1, 2, 3], keep)
    out = {"value": len(full) / sum(full), "unit": "sub-grids/s", "cores": k, "kind": "port",
           "sample": "3 timed 64^3 tiles (after 1 warm-up) of the same synthetic map, AF path, batch 1, torch CPU fp32, "
                     "%.1f s; a whole 4096-tile map is extrapolated linearly (tiles are independent and equal cost)" % sum(full),
           "seconds_per_tile": full}
    if k != 8:
        run(8, [0])
        t8 = run(8, [1, 2])
        out["threads_8"] = {"value": len(t8) / sum(t8), "cores": 8, "seconds_per_tile": t8}
    # the reference's numpy stages on the same map (single thread, as in the reference)
    n = vol_host.shape[0]
    m = min(n, 256)                                      # the spline resample costs ~8 s per 256^3 on one core: bounded sample
    t0 = time.perf_counter()
    vo.normalise_map(np.ascontiguousarray(vol_host[:m, :m, :m]) - np.float32(0.3))
    t_norm = time.perf_counter() - t0
    t0 = time.perf_counter()
    tiles, idx = vo.tile_volume(vol_host, 48, 8)
    t_tile = time.perf_counter() - t0
    t0 = time.perf_counter()
    vo.stitch_volume(tiles, idx, vol_host.shape, 8)
    t_stitch = time.perf_counter() - t0
    out["stages"] = {"map": f"{n}^3", "normalise_s": t_norm, "normalise_sample": f"{m}^3 corner of the map (scale by {(n / m) ** 3:.0f} for the whole map)", "tile_one_channel_s": t_tile, "stitch_one_channel_s": t_stitch,
                     "tiles": int(len(idx)),
                     "note": "numpy restatements of preprocessing.py:117-133 (zoom factor 1), create_grids.py:129-157 and "
                             "predict.py:459-501 on one channel, default tiling (48, 8); the reference tiles 25 channels and stitches 23"}
    return out


def _newest(pattern):
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return c[-1] if c else None


def load_traffic(src_hash, applicable):
    """Per-kernel PMC HBM bytes per launch of the newest profiles/rNN_pmc_traffic.json, if it was collected on these sources."""
    p = _newest("r[0-9][0-9]_pmc_traffic.json")
    if not applicable or p is None:
        return {}, "none (counter passes are taken at batch 8 on the AF path)" if p else "none"
    d = json.load(open(p))
    if d.get("library_source_hash") != src_hash:
        return {}, f"{os.path.basename(p)} REFUSED: collected on library sources {d.get('library_source_hash')}, running {src_hash}"
    return d.get("kernels", {}), os.path.basename(p)


def load_sq_summary(src_hash, applicable):
    """Per-kernel MFMA-busy share and held clock of the newest profiles/rNN_pmc_sq_summary.txt (tools/pmc_conv_summary.py), if it
    was collected on these sources: {kernel: {calls, avg_us, clock_ghz, mfma_busy}}."""
    p = _newest("r[0-9][0-9]_pmc_sq_summary.txt")
    if not applicable or p is None:
        return {}, "none"
    rows, stamp = {}, None
    for line in open(p):
        if line.startswith("library_source_hash:"):
            stamp = line.split(":", 1)[1].strip()
            continue
        f = line.split()
        if len(f) < 11 or "kernel" not in f[0]:
            continue
        try:
            nums = [float(v) for v in f[-10:]]
        except ValueError:
            continue
        rows[" ".join(f[:-10])] = {"calls": nums[0], "avg_us": nums[1], "clock_ghz": nums[2], "mfma_busy": nums[3] / 100.0}
    if stamp != src_hash:
        return {}, f"{os.path.basename(p)} REFUSED: collected on library sources {stamp}, running {src_hash}"
    return rows, os.path.basename(p)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this one has not touched the GPU),
    relay their output; rank 0 prints the JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--map", type=int, default=512, help="map edge (BASELINE metric: 512)")
    ap.add_argument("--grid", type=int, default=32)
    ap.add_argument("--pad", type=int, default=16)
    ap.add_argument("--no-af", action="store_true", help="zero-AF path (exp_downsizing branch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-tiling", action="store_true")
    ap.add_argument("--no-whole-map", action="store_true")
    ap.add_argument("--af-coverage", type=float, default=0.3,
                    help="N=1: also time the same map with atoms in a centred ball holding this fraction of the box only (a docked model covers "
                         "part of the map; tiles outside take the zero-AF branch, batches mix both) -> `mixed_af`; 0 = skip")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with --gpus 1: a process group of ONE rank whose record exchange still runs the collective "
                         "(--backend nccl executes the RCCL all_gather_into_tensor(async_op=True) branch on one GPU)")
    ap.add_argument("--af-f32", action="store_true", help="keep the AF3 encodings resident as float32 (default: uint8, a quarter of the bytes)")
    ap.add_argument("--gather-to-root", action="store_true",
                    help="N>1 / --force-exchange: bring the records to the stitching rank alone (dist.gather) instead of the all-gather "
                         "BASELINE.json's north_star names")
    ap.add_argument("--strong", action="store_true",
                    help="strong-scaling mode: ONE complete map, every window, sharded over the ranks (predict_volume_sharded), rank-0 "
                         "stitch and the final D2H of the four volumes inside the clock; --steps/--warmup are ignored")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])

    # stdout carries exactly ONE line, the JSON: libraries that greet on stdout (RCCL prints a version banner when its first
    # communicator comes up) are sent to stderr for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import numpy as np
    import torch
    import torch.distributed as dist

    from mica_amd.dist import RecordExchange
    from mica_amd.engine import Engine
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.weights import synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    grouped = world > 1 or args.force_exchange
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            sk.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    n = args.map
    B = args.batch
    dev = torch.device("cuda", local)
    seed = {256: 1001, 512: 1002}.get(n, 1000 + n)
    vol_host = np.random.default_rng(seed).random((n, n, n), dtype=np.float32)
    vol = torch.from_numpy(vol_host).to(dev)
    af = None
    if not args.no_af:
        # 24 binary channels with ~1e-3 occupancy, generated on the device channel by channel
        g = torch.Generator(device=dev).manual_seed(2001)
        # resident as uint8 (the encodings are binary, preprocessing.py:288-298): 3.2 GB instead of 12.9 GB at 512^3 on every rank;
        # mica_gather_tiles_u8 hands the network the same float32 tiles
        af = torch.empty((24, n, n, n), dtype=torch.float32 if args.af_f32 else torch.uint8, device=dev)
        for c in range(24):
            af[c] = (torch.rand((n, n, n), generator=g, device=dev) < 1e-3).to(af.dtype)
    weights = synth_state_dict(2022)
    S = args.grid + 2 * args.pad
    eng = Engine(local, max_batch=B, tile_size=S)
    eng.load_state_dict(weights)
    vp = VolumePredictor(eng, args.grid, args.pad, B)
    T = int(eng.lib.mica_tile_count(n, n, n, args.grid))
    g3, p = args.grid, args.pad
    out = torch.zeros((23, n, n, n), dtype=torch.float32, device=dev) if rank == 0 else None
    nb = max(T // B, 1)                         # whole batches per map; step k wraps around the map

    def first_of(k, r):
        return ((k * world + r) % nb) * B

    ex = None
    if grouped:
        # cropped records carry no halo: stitched with pad 0 on a grid-sized window
        ex = RecordExchange(B, (23, g3, g3, g3), dev, lambda rec, first: eng.stitch_tiles(rec, out, g3, 0, first), stitch_rank=0,
                            force_collective=args.force_exchange, gather_to_root=args.gather_to_root)

    def step(k, pred=vp, grid=g3, pad=p):
        rec = pred.run_batch(vol, af, first_of(k, rank), B)
        if ex is None:
            eng.stitch_tiles(rec, out, grid, pad, first_of(k, 0))
        else:
            ex.post(k, rec[:, :, p:p + g3, p:p + g3, p:p + g3], [(first_of(k, r), B) for r in range(world)])

    def sync():
        if ex is not None:
            ex.flush()
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    if args.strong:
        # ---- strong scaling: one complete map, fixed total work, everything the caller waits for inside the clock ----
        step(0)                                            # one untimed batch: kernels loaded, buffers touched
        sync()
        del out
        out = None
        sync()
        t0 = time.perf_counter()
        stats = {}
        # the four volumes reach the host slab by slab while later tiles compute (pipeline.py::SlabDownloader, pinned buffer
        # allocated beside the first slab's compute): what is left behind the last tile is one x slab, not 12.3 GB
        if grouped:
            vols = vp.predict_volume_sharded(vol, af, force_collective=args.force_exchange, stats=stats, to_host=True,
                                             gather_to_root=args.gather_to_root)
        else:
            vols = vp.predict_volume(vol, af, to_host=True)
        sync()
        dt = time.perf_counter() - t0
        if grouped:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if rank == 0:
            chk = float(abs(vols["amino_acid_probability"][:, ::16, ::16, ::16].sum(axis=0) - 1).max())
            rounds = (T + B * world - 1) // (B * world)
            emit({
                "metric": "64^3 sub-grids/sec", "value": T / dt, "unit": "sub-grids/s", "n_gpus": world, "steps": rounds, "warmup": 1,
                "ms_per_step": dt / rounds * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32 (3x f16 MFMA split products, f32 accumulate)", "data": "synthetic",
                "config": {"workload": f"ONE complete synthetic {n}^3 density map + 24-ch AF3 encodings, every one of its {T} windows (64 = grid "
                                       f"{args.grid} + 2x{args.pad} halo) sharded over {world} rank(s), gather+forward+softmax, "
                                       + (f"{'RCCL' if args.backend == 'nccl' else args.backend} {stats.get('collective', 'all_gather')} of cropped records, " if grouped else "")
                                       + "rank-0 stitch and the D2H of the four volumes (12.3 GB at 512^3; slab by slab behind the stitch) inside the clock",
                           "backend": args.backend if grouped else None, "tiles": T, "seconds_per_map": dt, "collectives": stats.get("collectives"),
                           "softmax_sum_check": chk, "af_path": not args.no_af}})
        if grouped:
            dist.barrier()
            dist.destroy_process_group()
        return

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    if grouped:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tiles = args.steps * B * world
    value = tiles / dt

    # the same map at the reference's default tiling (48, 8): 1331 windows of the same 64^3 size
    alt = None
    if rank == 0 and world == 1 and not args.no_alt_tiling and S == 64:
        vp48 = VolumePredictor(eng, 48, 8, B)
        T48 = int(eng.lib.mica_tile_count(n, n, n, 48))
        ks = max(args.steps // 2, 4)
        out.zero_()

        def step48(k):
            f = (k % max(T48 // B, 1)) * B
            eng.stitch_tiles(vp48.run_batch(vol, af, f, B), out, 48, 8, f)
        step48(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(ks):
            step48(1 + k)
        torch.cuda.synchronize()
        d48 = time.perf_counter() - t1
        alt = {"tiling": "grid 48 + 2x8 halo (reference default)", "tiles_per_map": T48, "value": ks * B / d48,
               "unit": "sub-grids/s", "steps": ks, "seconds_per_map": T48 / (ks * B / d48)}

    # one COMPLETE map, sustained: normalise (resample factor 1 + median/percentile select) -> gather -> forward -> softmax/argmax ->
    # stitch over all windows of the reference tiling, wall clock with a device sync on both sides.  Outside the timed steps; it shows
    # whether the rate of the short timed region holds over ~18 s of a power-bound kernel mix.
    whole = None
    if rank == 0 and world == 1 and not args.no_whole_map and S == 64:
        vpw = VolumePredictor(eng, 48, 8, B)
        Tw = int(eng.lib.mica_tile_count(n, n, n, 48))
        del out
        out = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        norm = eng.zoom_cubic(vol - 0.3, (1.0, 1.0, 1.0))           # preprocessing.py:117 runs zoom even at factor 1
        eng.normalise_map_(norm)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        vols = vpw.predict_volume(norm, af)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        chk = float(vols["amino_acid_probability"][:, ::16, ::16, ::16].sum(dim=0).sub(1).abs().max())
        whole = {"map": f"{n}^3", "tiling": "grid 48 + 2x8 halo (reference default)", "tiles": Tw, "seconds": t3 - t1,
                 "normalise_seconds": t2 - t1, "predict_seconds": t3 - t2, "value": Tw / (t3 - t2), "unit": "sub-grids/s",
                 "value_incl_normalise": Tw / (t3 - t1), "softmax_sum_check": chk,
                 "note": "one complete map, every window, wall clock; `value` = tiles / (gather+forward+softmax+stitch seconds), the "
                         "same scope as the headline value (compare alt_tiling, the same tiling over a short window)"}
        del vols, norm
        out = torch.zeros((23, n, n, n), dtype=torch.float32, device=dev)

    # the same map with atoms in part of the box only: a centred ball holding `--af-coverage` of the volume.  z is the fastest tile
    # index, so batches of consecutive tiles cross the ball's surface and mix the two input branches (models/model.py:56-74).  Since
    # round 6 only MultiScaleInput runs per branch; encoders, FPN and heads run once per batch (forward.hip::trunk).
    mixed = None
    if rank == 0 and world == 1 and af is not None and args.af_coverage > 0:
        rr = (args.af_coverage * 3.0 / (4.0 * np.pi)) ** (1.0 / 3.0) * n
        ax = (torch.arange(n, device=dev, dtype=torch.float32) - (n - 1) / 2.0) ** 2
        ball = (ax[:, None, None] + ax[None, :, None] + ax[None, None, :]) <= rr * rr
        afm = af * ball.to(af.dtype)
        cover = float(ball.float().mean())
        del ball
        ks = 40
        sel = [int(round(q * (nb - 1) / (ks - 1))) * B for q in range(ks)]           # batches spread evenly over the whole map
        runs, with_atoms = 0, 0

        def stepm(f, count=False):
            nonlocal runs, with_atoms
            eng.stitch_tiles(vp.run_batch(vol, afm, f, B), out, g3, p, f)
            if count:
                runs += eng.last_input_runs
                with_atoms += eng.last_af_tiles
        stepm(sel[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in sel:
            stepm(f)
        torch.cuda.synchronize()
        dm = time.perf_counter() - t1
        for f in sel:                       # untimed second pass: how the batches were cut
            stepm(f, True)
        torch.cuda.synchronize()
        mixed = {"af_coverage": cover, "value": ks * B / dm, "unit": "sub-grids/s", "steps": ks, "ratio_to_value": ks * B / dm / value,
                 "tiles_with_atoms": with_atoms / (ks * B), "input_runs_per_batch": runs / ks,
                 "note": "same map and tiling as `value`, atoms only inside a centred ball of that volume share; 40 batches spread evenly over "
                         "the map; input_runs_per_batch = runs of consecutive tiles with equal AF3 gate per batch (MultiScaleInput launches per "
                         "run; the rest of the network once per batch)"}
        del afm

    # rooflines: HIP events (on the launch stream, inside the library) around every dense-conv and every depthwise
    # conv3d launch of one extra, untimed batch.  PMC traffic comes from the committed rocprofv3 passes (profiles/).
    roof = hbm = None
    if rank == 0:
        eng.set_profiling(True)
        vp.run_batch(vol, af, 0, B)
        torch.cuda.synchronize()
        ms, launches, flops = eng.profile(0)
        dms, dl, dbytes = eng.profile(1)
        wms, wl, wflops = eng.profile(2)
        eng43 = eng.profile(5)
        e64ms, e64l, e64flops = eng.profile(6)
        eng.set_profiling(False)
        # counter figures come from the committed rocprofv3 passes of tools/profile.sh (profiles/rNN_pmc_traffic.json,
        # rNN_pmc_sq_summary.txt); each is stamped with the hash of the library sources it was collected on and is quoted only
        # if that equals the running tree's - a kernel change without a re-profile yields null, not stale bytes
        from mica_amd._cabi import source_hash
        here = source_hash()
        traffic, tnote = load_traffic(here, B == 8 and not args.no_af)
        sq, sqnote = load_sq_summary(here, B == 8 and not args.no_af)
        wach = wflops / (wms * 1e-3) / 1e12 if wms > 0 else 0.0          # algorithmic TF, F(2,3) launches
        fms, fl, fflops = eng43
        fach = fflops / (fms * 1e-3) / 1e12 if fms > 0 else 0.0          # algorithmic TF, F(4,3) launches
        ach_all = flops / (ms * 1e-3) / 1e12

        def mfma_obj(name, key, alg_tf, fac, tms, n, alg_flops, sqkeys):
            """Roofline of one 3^3 conv kernel against the dense f16 MFMA peak: achieved = f16 MFMA FLOPs the kernel ISSUES
            (algorithmic direct-conv FLOPs x executed-per-algorithmic) / HIP-event time."""
            o = {"kernel": name, "achieved": alg_tf * fac, "peak": PEAK_F16_MFMA_TF, "unit": "TFLOP/s", "frac": alg_tf * fac / PEAK_F16_MFMA_TF,
                 "traffic": traffic.get(key, {}).get("hbm_bytes"), "launches_per_batch": n, "avg_launch_ms": tms / max(n, 1),
                 "executed_gflop_per_launch_avg": alg_flops * fac / max(n, 1) / 1e9,
                 "algorithmic_gflop_per_launch_avg": alg_flops / max(n, 1) / 1e9,
                 "algorithmic_tflops": alg_tf, "executed_per_algorithmic": fac,
                 "algorithmic_frac_of_split_f16_peak": alg_tf / PEAK_SPLIT_TF}
            rows = [sq[k] for k in sqkeys if k in sq]
            if rows:
                t = sum(r["calls"] * r["avg_us"] for r in rows)
                busy = sum(r["calls"] * r["avg_us"] * r["mfma_busy"] for r in rows) / t
                clk = sum(r["calls"] * r["avg_us"] * r["clock_ghz"] for r in rows) / t
                o.update(mfma_busy=busy, held_clock_ghz=clk, mfma_busy_x_clock_over_2p4=busy * clk / 2.4,
                         profiled_avg_launch_ms=t / sum(r["calls"] for r in rows) / 1e3)
            else:
                o.update(mfma_busy=None, held_clock_ghz=None, mfma_busy_x_clock_over_2p4=None)
            return o

        w16 = mfma_obj("conv_wino16_kernel<128|64|32>: Winograd F(2,3)-x, the 3x3x3 convs outside encoder.2", "conv_wino16_kernel", wach,
                       WINO_MFMA_PER_ALGORITHMIC, wms, wl, wflops, ["conv_wino16_kernel<128>", "conv_wino16_kernel<64>", "conv_wino16_kernel<32>"])
        if fl > 0:
            roof = mfma_obj("conv_wino43_kernel<128>: the four 3x3x3 convs of encoder.2 (68 % of the network's FLOPs) via Winograd F(4,3)-x, split-f16 x3 "
                            "MFMA (v_mfma_f32_16x16x32_f16)", "conv_wino43_kernel<128>", fach, WINO43_MFMA_PER_ALGORITHMIC, fms, fl, fflops,
                            ["conv_wino43_kernel<128>", "conv_wino43_kernel"])
            roof["conv_wino16"] = w16
            if e64l > 0:
                roof["conv_wino43_64"] = mfma_obj("conv_wino43_kernel<64>: the tap-split 64-channel variant - FPN smooth convs, the heads' conv1 (operand-stream-"
                                                  "bound, not MFMA-bound: profiles/r05_f43_late_ab.txt)", "conv_wino43_kernel<64>", e64flops / (e64ms * 1e-3) / 1e12,
                                                  WINO43_MFMA_PER_ALGORITHMIC, e64ms, e64l, e64flops, ["conv_wino43_kernel<64>"])
        else:
            roof = w16
        ex_all = (wflops * WINO_MFMA_PER_ALGORITHMIC + (fflops + e64flops) * WINO43_MFMA_PER_ALGORITHMIC) / max((wms + fms + e64ms) * 1e-3, 1e-12) / 1e12
        roof = dict({"bound": "mfma"}, **roof)
        roof["all_3x3x3_convs"] = {"launches_per_batch": wl + fl + e64l, "ms_per_batch": wms + fms + e64ms, "achieved": ex_all, "peak": PEAK_F16_MFMA_TF,
                                   "frac": ex_all / PEAK_F16_MFMA_TF,
                                   "algorithmic_tflops": (wflops + fflops + e64flops) / max((wms + fms + e64ms) * 1e-3, 1e-12) / 1e12}
        roof["all_dense_convs"] = {"algorithmic_tflops": ach_all, "launches_per_batch": launches, "ms_per_batch": ms,
                                   "note": "3x3x3 and 1x1x1 launches together (the 1x1 kernel is HBM-bound and also does the operand passes)"}
        roof["profile_source"] = {"library_source_hash": here, "traffic": tnote, "counters": sqnote}
        per_layer = traffic.get("conv_wino43_kernel<128>", {}).get("per_layer")
        if per_layer:
            roof["traffic_by_layer_read_bytes"] = {k: v["hbm_read_bytes"] for k, v in per_layer.items()}
            roof["traffic_note"] = ("FETCH_SIZE / WRITE_SIZE count requests at the L2's fabric side, Infinity-Cache hits included; by layer the reads are "
                                    "'packed weights once per XCD round of items + 1.2 x the operand' to 1 % - two thirds are re-reads of <= 31.5 MB of "
                                    "weights, a table that stays in the 256-MB Infinity Cache (DESIGN.md section 4, 'Round 6' (c)): the HBM side is "
                                    "about 1.5 x the algorithmic 5.1 GB, not 4 x")
        roof["note"] = ("achieved = f16 MFMA FLOPs the dominant kernel issues per launch (algorithmic direct-conv FLOPs 2*27*Cin*Cout*V, unpadded, x "
                        "executed_per_algorithmic: / 2 for Winograd F(4,3) [/ 1.5 for F(2,3)] x 3 split products x 14/13.5 tap pairing) / average launch "
                        "time from HIP events on the launch stream; peak = 2500 TF dense f16 MFMA; frac is a hardware fraction (<= 1) and should equal "
                        "mfma_busy x held_clock / 2.4 GHz from the SQ counters of the same tree; the algorithmic rate (which credits Winograd's saving "
                        "and the 3-product split: 833 TF ceiling) is in algorithmic_tflops; traffic = PMC HBM bytes per launch at batch 8 (rocprofv3 "
                        "--pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections), null when the committed profile was taken on other library sources")
        dach = dbytes / (dms * 1e-3) / 1e9
        # what a plain streaming kernel (y = 2 x, one read + one write per element, 2.7 GB) reaches on THIS box: the practical ceiling the
        # depthwise kernel's traffic shape could approach (MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 TB/s peak) - context, not product
        sx = torch.empty(336 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
        sy = torch.empty_like(sx)
        torch.mul(sx, 2.0, out=sy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            torch.mul(sx, 2.0, out=sy)
        e1.record()
        torch.cuda.synchronize()
        plain = 5 * 2 * sx.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del sx, sy
        hbm = {"bound": "hbm", "kernel": "depthwise_kernel (Conv3d groups=C, 3x3x3, IN+ReLU+SE gate fused on load, IN stats fused)",
               "achieved": dach, "peak": 8000.0, "unit": "GB/s", "frac": dach / 8000.0,
               "traffic": traffic.get("depthwise_kernel", {}).get("hbm_bytes"), "launches_per_batch": dl,
               "avg_launch_ms": dms / max(dl, 1), "algorithmic_bytes_per_launch": dbytes / max(dl, 1),
               "plain_stream_gbs_on_this_box": plain, "frac_of_plain_stream": dach / plain,
               "note": "BASELINE metric's 'HBM GB/s on conv3d': algorithmic 8 B per voxel and channel / HIP-event time"}
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N=1 only
        first = (T // 2 // 4) * 4
        tm = eng.gather_tiles(vol, g3, p, first, 4).cpu().view(4, 1, S, S, S)
        ta = eng.gather_tiles(af, g3, p, first, 4).cpu() if af is not None else torch.zeros((4, 24, S, S, S))
        # the GPU's logits of the tiles the CPU leg is about to compute, EVERY voxel (the checker's outputs used as a checker)
        from mica_amd.engine import AF_PER_TILE
        gl = [t.cpu().numpy() for t in eng.forward_logits(tm.to(dev), ta.to(dev), AF_PER_TILE)]
        del out
        kept = {}
        cpu = cpu_baseline(weights, tm, ta, vol_host, kept)

        def scaled(a, b):
            a, b = a.astype(np.float64), b.astype(np.float64)
            e = np.abs(a - b) / np.maximum(np.abs(b), np.sqrt(np.mean(b * b)))
            return float(e.max()), float(np.sqrt(np.mean(e * e)))
        per = [[scaled(gl[h][i:i + 1], kept[i][h]) for h in range(3)] for i in sorted(kept)]
        parity = {"tiles": len(per), "voxels_per_tile": int(S ** 3), "logits_compared": int(len(per) * 29 * S ** 3),
                  "max_scaled": [max(t[h][0] for t in per) for h in range(3)], "rms_scaled": [float(np.sqrt(np.mean([t[h][1] ** 2 for t in per]))) for h in range(3)],
                  "heads": ["backbone", "carbon_alpha", "amino_acid"],
                  "against": "oracle/model_oracle.py (torch CPU float32 = the reference CPU path's arithmetic, bit-equal to the reference module) on "
                             "every logit of the 64^3 tiles the cpu_baseline leg times; metric max / rms of |gpu - ref| / max(|ref|, rms(ref)); "
                             "the reference's own float32 runs differ from each other by up to 1.06e-4 in this metric (tests/golden/manifest.json S64)"}
    if rank == 0:
        emit({
            "metric": "64^3 sub-grids/sec", "value": value, "unit": "sub-grids/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (3x f16 MFMA split products, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"synthetic {n}^3 density map + 24-ch AF3 encodings, window 64 = grid {args.grid} + 2x{args.pad} halo, "
                                   f"{T} tiles per map, {B} tiles per step per GPU, gather+forward+softmax+stitch"
                                   + ("" if not grouped else f", {'RCCL' if args.backend == 'nccl' else args.backend + ' (rehearsal, host-staged)'} "
                                      + ("gather of cropped records into rank 0, which stitches" if args.gather_to_root else "all-gather of cropped records to every rank, rank 0 stitches")
                                      + (f" (collective forced in a group of one rank; {ex.collectives} all-gathers issued)" if world == 1 else ""))
                                   + (", every rank on cuda:0" if args.single_device else ""),
                       "backend": args.backend if grouped else None,
                       "tiles_per_step": B * world, "af_path": not args.no_af, "encodings_resident_as": None if af is None else str(af.dtype).replace("torch.", ""),
                       "flops_per_tile": FLOPS_PER_TILE_AF,
                       "seconds_per_map": T / value},
            "sustained": None if whole is None else {
                "value": whole["value"], "unit": "sub-grids/s", "seconds": whole["predict_seconds"], "tiles": whole["tiles"],
                "note": "ONE complete map, every window of the reference tiling (48, 8), wall clock: what the power-bound kernel mix holds "
                        "over a whole map; `value` above is the contract's K timed steps (a ~2-s window, typically 2-3 % higher)"},
            "alt_tiling": alt, "whole_map": whole, "mixed_af": mixed, "parity": parity, "roofline": roof, "hbm_conv3d": hbm, "cpu_baseline": cpu})
    if grouped:
        dist.barrier()                          # rank 0 ran the extra profiled batch: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
