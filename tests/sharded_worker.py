"""One rank of the rehearsals of BASELINE configs[2] on a single GPU (started by tests/test_gpu_configs.py as a fresh child
process, rendezvous on 127.0.0.1, every rank on cuda:0): runs VolumePredictor.predict_volume_sharded with the real engine;
rank 0 writes the four volumes.  MICA_TEST_BACKEND=gloo (default; two ranks, records staged through the host) or nccl
(= RCCL; ONE rank with force_collective, which executes the production branch of RecordExchange: device tensors,
all_gather_into_tensor(async_op=True), work.wait() ordering, double-buffered slots).  MICA_TEST_MODE=root: the round-5 form - encodings
resident as uint8, records gathered into rank 0 alone (dist.gather), volumes downloaded slab by slab behind the stitch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, shape, batch = sys.argv[1], tuple(int(v) for v in sys.argv[2].split("x")), int(sys.argv[3])
    grid, pad, seed = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (48, 8, 91)
    zero_x = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    import numpy as np
    import torch
    import torch.distributed as dist

    from mica_amd.engine import Engine
    from mica_amd.pipeline import VolumePredictor
    from mica_amd.synth import synth_af, synth_density
    from mica_amd.weights import synth_state_dict

    backend = os.environ.get("MICA_TEST_BACKEND", "gloo")
    force = backend == "nccl"
    if backend == "nccl":       # the group comes first: nothing else has touched the GPU in this process
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    eng = Engine(0, max_batch=batch, tile_size=64)
    eng.load_state_dict(synth_state_dict(2022))
    vol = torch.from_numpy(synth_density(shape, seed)).cuda()
    af = torch.from_numpy(synth_af(shape, seed, 2e-3)).cuda()
    if zero_x >= 0:
        af[:, :zero_x] = 0                                 # windows that end before x = zero_x see no atoms
    else:
        af[:, :, :, : shape[2] // 2] = 0                   # some tiles see no atoms: per-tile gating on every rank
    stats = {}
    root_mode = os.environ.get("MICA_TEST_MODE", "") == "root"
    if root_mode:
        af = af.to(torch.uint8)
    out = VolumePredictor(eng, grid, pad, batch).predict_volume_sharded(vol, af, force_collective=force, stats=stats, to_host=root_mode,
                                                                        gather_to_root=root_mode)
    # coverage of the sharded stitch: all-one records through the same exchange fill a counter volume (no hole), and
    # sharded_records itself raises unless every batch arrived exactly once
    from mica_amd.dist import sharded_records
    T = int(eng.lib.mica_tile_count(*shape, grid))
    cover = torch.zeros((1, *shape), device="cuda")
    ones = torch.ones((batch, 1, grid, grid, grid), device="cuda")
    sharded_records(lambda first, count: ones[:count], lambda rec, first: eng.stitch_tiles(rec.contiguous(), cover, grid, 0, first),
                    T, batch, (1, grid, grid, grid), torch.device("cuda"), stitch_rank=0, force_collective=force, gather_to_root=root_mode)
    if dist.get_rank() == 0:
        np.savez(out_path, coverage=cover[0].cpu().numpy(), collectives=stats["collectives"], backend=str(stats["backend"]),
                 collective=str(stats["collective"]), **{k: (v if root_mode else v.cpu().numpy()) for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
