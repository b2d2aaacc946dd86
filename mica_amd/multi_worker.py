"""Entry of a worker rank of mica_amd/multi.py::RankPool (`python -m mica_amd.multi_worker ...`): a FRESH process per GPU, started by
rank 0 (the process that called `CryoEMPredictor(..., gpus=N).run_prediction()`), which stays for every map of that pool and exits
when told to - or when its parent is gone."""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time


def _leave_with_parent(ppid: int):
    """A rank-0 process that was killed cannot say "exit": a worker whose parent has changed leaves by itself (it holds a 37-GB
    workspace on its GPU)."""
    def watch():
        while True:
            time.sleep(1.0)
            if os.getppid() != ppid:
                os._exit(4)
    threading.Thread(target=watch, name="mica-parent-watch", daemon=True).start()


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m mica_amd.multi_worker")
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--runner", default="mica_amd.multi:EngineRunner")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--conv-variant", type=int, default=None)
    ap.add_argument("--timeout", type=float, default=600.0)
    ap.add_argument("--spawned", type=float, default=None)
    a = ap.parse_args(argv)
    _leave_with_parent(os.getppid())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from mica_amd import multi
    multi.serve(a.rank, a.world, a.port, a.backend, a.device, a.runner, a.tile, a.batch, a.conv_variant, a.timeout,
                a.spawned if a.spawned is not None else time.time())
    return 0


if __name__ == "__main__":
    sys.exit(main())
