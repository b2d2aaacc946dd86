"""Headline benchmark: 64^3 sub-grids/s of the MICA hot path (tile gather -> network forward ->
softmax/argmax -> stitch) on synthetic maps resident in HBM.  One JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one batch of `--batch` tiles through the whole path.  N=1: BASELINE.json configs[1]
(synthetic 256^3 map, stride-32 tiling = grid 32 + 2*16 halo -> 512 tiles of 64^3); N>1: configs[2]
(synthetic 512^3 map, 4096 tiles) with tile batches dealt round-robin to the ranks and the cropped
per-tile records all-gathered over RCCL/xGMI so that rank 0 stitches the volumes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_TILE_AF = 7.3625e12       # BASELINE.md section 3 (AF path), measured on the reference model
PEAK_F16_MFMA_TF = 2500.0           # MI355X dense f16 MFMA (MI355X_MICROARCH.md)
PEAK_SPLIT_TF = PEAK_F16_MFMA_TF / 3.0   # this path spends three f16 MFMAs per f32-grade product


def cpu_baseline(weights, tile_map, tile_af, threads):
    """The CPU oracle (validated bit-exact against the reference module) timed on this host:
    one 64^3 tile of the same workload, AF path, batch 1."""
    import torch
    from oracle import model_oracle as mo   # cpu_baseline leg only
    torch.set_num_threads(threads)
    t0 = time.time()
    mo.mica_forward(weights, tile_map, tile_af)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "sub-grids/s", "cores": threads, "kind": "port",
            "sample": "1 tile of 64^3 (AF path, batch 1, torch CPU fp32) of the same synthetic map, %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--map", type=int, default=0, help="map edge (default 256 at N=1, 512 at N>1)")
    ap.add_argument("--grid", type=int, default=32)
    ap.add_argument("--pad", type=int, default=16)
    ap.add_argument("--no-af", action="store_true", help="zero-AF path (exp_downsizing branch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from mica_amd.engine import Engine
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.weights import synth_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    n = args.map or (256 if world == 1 else 512)
    B = args.batch
    dev = torch.device("cuda", local)
    rng = np.random.default_rng(1001 if n == 256 else 1002)
    vol = torch.from_numpy(rng.random((n, n, n), dtype=np.float32)).to(dev)
    af = None
    if not args.no_af:
        # 24 binary channels with ~1e-3 occupancy, generated on the device
        g = torch.Generator(device=dev).manual_seed(2001)
        af = (torch.rand((24, n, n, n), generator=g, device=dev) < 1e-3).float()
    weights = synth_state_dict(2022)
    eng = Engine(local, max_batch=B, tile_size=args.grid + 2 * args.pad)
    eng.load_state_dict(weights)
    vp = VolumePredictor(eng, args.grid, args.pad, B)
    T = int(eng.lib.mica_tile_count(n, n, n, args.grid))
    g3 = args.grid
    p = args.pad
    out = torch.zeros((23, n, n, n), dtype=torch.float32, device=dev) if rank == 0 else None
    # N > 1: two slots, so that the all-gather of step k (async, on RCCL's stream) overlaps the compute of step k+1
    gathered = [torch.empty((world * B, 23, g3, g3, g3), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    crops = [torch.empty((B, 23, g3, g3, g3), dtype=torch.float32, device=dev) for _ in range(2)] if world > 1 else None
    pending = []

    def finish():
        work, k, slot = pending.pop()
        work.wait()
        if rank == 0:
            # cropped records carry no halo: stitch them with pad 0 on a grid-sized window
            for r in range(world):
                f = ((k * world + r) * B) % max(T - B + 1, 1)
                eng.stitch_tiles(gathered[slot][r * B:(r + 1) * B], out, g3, 0, f)

    def step(k):
        first = ((k * world + rank) * B) % max(T - B + 1, 1)
        rec = vp.run_batch(vol, af, first, B)
        if world == 1:
            eng.stitch_tiles(rec, out, args.grid, p, first)
            return
        slot = k & 1
        crops[slot].copy_(rec[:, :, p:p + g3, p:p + g3, p:p + g3])
        if args.backend == "nccl":
            work = dist.all_gather_into_tensor(gathered[slot], crops[slot], async_op=True)
        else:                                   # gloo rehearsal: stage through the host
            class _Done:
                def wait(self):
                    pass
            parts = [torch.empty(crops[slot].shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, crops[slot].cpu())
            gathered[slot].copy_(torch.cat(parts).to(dev))
            work = _Done()
        if pending:
            finish()                            # step k-1: its gather ran beside this step's kernels
        pending.append((work, k, slot))

    def sync():
        if pending:
            finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tiles = args.steps * B * world
    value = tiles / dt

    # rooflines: HIP events (on the launch stream, inside the library) around every dense-conv and every depthwise
    # conv3d launch of one extra, untimed batch.  PMC traffic comes from the committed rocprofv3 passes (profiles/).
    roof = hbm = None
    if rank == 0:
        eng.set_profiling(True)
        vp.run_batch(vol, af, 0, B)
        torch.cuda.synchronize()
        ms, launches, flops = eng.profile(0)
        dms, dl, dbytes = eng.profile(1)
        eng.set_profiling(False)
        traffic = {}
        tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")     # re-collected whenever the conv kernels change
        if os.path.exists(tp) and B == 8 and n == 256 and not args.no_af:      # measured for exactly this workload
            traffic = json.load(open(tp)).get("kernels", {})
        ach = flops / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_wino16_kernel (+ conv2_kernel for 1x1x1): dense 3x3x3 via Winograd F(2,3)-x, split-f16 x3 MFMA",
                "achieved": ach, "peak": PEAK_SPLIT_TF, "unit": "TFLOP/s", "frac": ach / PEAK_SPLIT_TF,
                "traffic": traffic.get("conv_wino16_kernel", traffic.get("conv_wino_kernel", {})).get("hbm_bytes"),
                "launches_per_batch": launches, "avg_launch_ms": ms / max(launches, 1),
                "algorithmic_gflop_per_launch_avg": flops / max(launches, 1) / 1e9,
                "note": "achieved = algorithmic direct-conv FLOPs (2*k^3*Cin*Cout*V, unpadded) / HIP-event time of the conv launches; "
                        "peak = f16 dense MFMA 2500 TF / 3 MFMAs per f32-grade product; Winograd executes 1.5x fewer MFMAs than the "
                        "algorithmic count; traffic = PMC HBM bytes per conv_wino16 launch of `python bench.py` defaults (batch 8; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections, profiles/r01_pmc_traffic.json); null for other batch sizes"}
        dach = dbytes / (dms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "kernel": "depthwise_kernel (Conv3d groups=C, 3x3x3, IN+ReLU+SE gate fused on load, IN stats fused)",
               "achieved": dach, "peak": 8000.0, "unit": "GB/s", "frac": dach / 8000.0,
               "traffic": traffic.get("depthwise_kernel", {}).get("hbm_bytes"), "launches_per_batch": dl,
               "avg_launch_ms": dms / max(dl, 1),
               "note": "BASELINE metric's 'HBM GB/s on conv3d': algorithmic 8 B per voxel and channel / HIP-event time"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N=1 only
        S = args.grid + 2 * args.pad
        tm = eng.gather_tiles(vol, args.grid, p, T // 2, 1).cpu()
        ta = eng.gather_tiles(af, args.grid, p, T // 2, 1).cpu() if af is not None else None
        cpu = cpu_baseline(weights, tm.view(1, 1, S, S, S), ta, threads=min(os.cpu_count() or 8, 16))
    if rank == 0:
        print(json.dumps({
            "metric": "64^3 sub-grids/sec", "value": value, "unit": "sub-grids/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (3x f16 MFMA split products, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"synthetic {n}^3 density map + 24-ch AF3 encodings, window 64 = grid {args.grid} + 2x{args.pad} halo, "
                                   f"{T} tiles per map, {B} tiles per step per GPU, gather+forward+softmax+stitch"
                                   + ("" if world == 1 else ", RCCL all-gather of cropped records to every rank, rank 0 stitches"),
                       "tiles_per_step": B * world, "af_path": not args.no_af, "flops_per_tile": FLOPS_PER_TILE_AF},
            "roofline": roof, "hbm_conv3d": hbm, "cpu_baseline": cpu}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
