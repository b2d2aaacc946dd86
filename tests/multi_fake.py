"""Stand-in per-rank compute object for the CPU tests of mica_amd/multi.py::RankPool (TEST INFRASTRUCTURE): the process logic of the
pool - spawn, rendezvous, command, broadcast, sharded exchange, status, failure, shutdown - runs for real over gloo with CPU tensors;
only the network is replaced by "record = the tile's cropped window times the number in the 'checkpoint' file".

Failure injection through MICA_FAKE_FAIL = "<where>:<rank>" with where in init | load | predict."""
import os
import sys

import numpy as np
import torch


def _fail(where: str, rank: int) -> bool:
    return os.environ.get("MICA_FAKE_FAIL", "") == f"{where}:{rank}"


def _argv_rank() -> int:
    return int(sys.argv[sys.argv.index("--rank") + 1]) if "--rank" in sys.argv else 0


class FakeRunner:
    def __init__(self, device_index, tile, batch, conv_variant=None):
        if _fail("init", _argv_rank()):
            raise RuntimeError("injected: no such device")
        self.device = torch.device("cpu")
        self.tile, self.batch = tile, batch
        self.scale = None
        self.loads = 0

    def load_model(self, model_path):
        import torch.distributed as dist
        if _fail("load", dist.get_rank()):
            raise FileNotFoundError(f"injected: {model_path}")
        self.scale = float(open(model_path).read())
        self.loads += 1

    def empty(self, shape, dtype):
        return torch.empty(tuple(shape), dtype=dtype)

    def predict(self, vol, af, grid, pad, force_collective, gather_to_root, to_host, stats):
        import torch.distributed as dist
        from mica_amd.dist import sharded_records
        if _fail("predict", dist.get_rank()):
            raise RuntimeError("injected: kernel fault")
        n0, n1, n2 = vol.shape
        nt = [-(-n // grid) for n in (n0, n1, n2)]
        T = nt[0] * nt[1] * nt[2]
        padded = torch.zeros((n0 + grid, n1 + grid, n2 + grid), dtype=torch.float32)
        padded[:n0, :n1, :n2] = vol
        extra = 0.0 if af is None else float(af.sum())
        out = torch.zeros((1, n0, n1, n2)) if dist.get_rank() == 0 else None

        def origin(t):
            return (t // (nt[1] * nt[2])) * grid, ((t // nt[2]) % nt[1]) * grid, (t % nt[2]) * grid

        def run(first, count):
            rec = torch.empty((count, 1, grid, grid, grid))
            for q in range(count):
                i, j, k = origin(first + q)
                rec[q, 0] = padded[i:i + grid, j:j + grid, k:k + grid] * self.scale + extra
            return rec

        def stitch(rec, first):
            for q in range(rec.shape[0]):
                i, j, k = origin(first + q)
                di, dj, dk = min(grid, n0 - i), min(grid, n1 - j), min(grid, n2 - k)
                out[0, i:i + di, j:j + dj, k:k + dk] = rec[q, 0, :di, :dj, :dk]
        sharded_records(run, stitch, T, self.batch, (1, grid, grid, grid), torch.device("cpu"), stitch_rank=0, stats=stats,
                        gather_to_root=gather_to_root)
        if dist.get_rank() != 0:
            return None
        return {"volume": out[0].numpy() if to_host else out[0]}

    def close(self):
        pass
