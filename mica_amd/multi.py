"""The GPUs of one node behind the reference's single-process call site (BASELINE.json configs[2]).

`Solver.nnPred` is ONE Python process that builds `CryoEMPredictor(...)` and calls `run_prediction()` (reference
utils/modeler.py:722-738, utils/predict.py:48, 589-634).  Tiles are independent, so the map shards over the GPUs of the node - but
somebody has to start the other ranks, hand them the map and collect their records.  That is this module:

  * rank 0 IS the calling process (its engine, its volumes, its stitch and its download - everything `run_prediction` does on one GPU);
  * ranks 1 .. N-1 are FRESH child processes (`python -m mica_amd.multi_worker`; never a re-exec of a process that touched a GPU),
    one per GPU, started once and kept for the next map: each imports torch, builds its Engine (the 37-GB workspace) and then joins
    the rendezvous on 127.0.0.1, so that start-up overlaps whatever rank 0 is still doing (`prestart`: the Solver-flow shim starts
    them when `DataPreprocessor` is constructed, i.e. beside the whole of getData);
  * per map: one command over a gloo control group (CPU sockets: an idle worker sleeps in `recv`, it does not spin a kernel), the
    normalised map (f32) and the encodings (u8) with ONE `broadcast` each over RCCL / xGMI (0.54 + 3.2 GB at 512^3), then
    `VolumePredictor.predict_volume_sharded` on every rank - round-robin batches, one all-gather (or gather to rank 0) of cropped
    records per round, rank 0 stitches and downloads slab by slab - and one status gather (errors, timings);
  * the workers exit with the pool: `close()` (or interpreter exit, or a failed map) sends "exit", joins every child with a timeout and
    kills by exact PID what does not leave; a worker that dies takes the map down loudly, never silently.

Nothing here computes: the ranks run the same HIP path as one GPU does.  The process logic is device-agnostic (a `runner` factory
builds the per-rank compute object), so that start-up, command, failure and shutdown are covered by CPU tests with a stand-in
runner (tests/multi_fake.py) beside the GPU test that drives two real ranks on one card over gloo.
"""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

from ._cabi import MicaHipError

DEFAULT_RUNNER = "mica_amd.multi:EngineRunner"
_POOLS: dict = {}
_LOCK = threading.RLock()


def configured_gpus(gpus=None) -> int:
    """`gpus` argument, else MICA_GPUS, else 1."""
    if gpus is None:
        gpus = os.environ.get("MICA_GPUS", "1")
    n = int(gpus)
    if n < 1 or n > 64:
        raise ValueError(f"gpus must be in [1, 64], got {gpus!r}")
    return n


def _resolve(spec: str):
    mod, _, name = spec.partition(":")
    import importlib
    return getattr(importlib.import_module(mod), name)


# ---- the per-rank compute object of the product --------------------------------------------------------------------------------------
class EngineRunner:
    """One rank's HIP path: an Engine on its GPU, weights from a checkpoint, `predict_volume_sharded` over the default group."""

    def __init__(self, device_index: int, tile: int, batch: int, conv_variant=None, engine=None, loaded_model: str | None = None):
        """engine: rank 0 passes the engine it already has (with the weights of `loaded_model` in it); a worker builds its own."""
        from .engine import Engine
        self.own = engine is None
        self.engine = engine if engine is not None else Engine(device_index, max_batch=batch, tile_size=tile, conv_variant=conv_variant)
        self.device = self.engine.device
        self.batch = batch
        self.model_key = None
        if loaded_model is not None and self.engine.weights_loaded:
            try:
                self.model_key = self._key(loaded_model)
            except OSError:
                pass                                # load_model will say so, where every rank reports

    @staticmethod
    def _key(model_path: str):
        st = os.stat(model_path)
        return (os.path.realpath(model_path), st.st_size, st.st_mtime_ns)

    def load_model(self, model_path: str):
        from .weights import load_checkpoint_state_dict
        key = self._key(model_path)
        if key != self.model_key:
            if self.engine.weights_loaded:          # a context packs its weights once: another checkpoint needs a fresh one
                from .engine import Engine
                old = self.engine
                self.engine = Engine(self.device, max_batch=old.max_batch, tile_size=old.tile_size, conv_variant=old.conv_variant)
                if self.own:
                    old.close()
                self.own = True
            self.engine.load_state_dict(load_checkpoint_state_dict(model_path))
            self.model_key = key

    def empty(self, shape, dtype):
        return torch.empty(tuple(shape), dtype=dtype, device=self.device)

    def predict(self, vol, af, grid: int, pad: int, force_collective: bool, gather_to_root: bool, to_host: bool, stats: dict):
        from .pipeline import VolumePredictor
        vp = VolumePredictor(self.engine, grid, pad, self.batch)
        return vp.predict_volume_sharded(vol, af, force_collective=force_collective, stats=stats, to_host=to_host, gather_to_root=gather_to_root)

    def close(self):
        if self.own and self.engine is not None:
            self.engine.close()
        self.engine = None


# ---- collectives shared by rank 0 and the workers ------------------------------------------------------------------------------------
def _bcast(t: torch.Tensor, backend: str):
    """Rank 0's tensor to every rank.  RCCL moves device tensors; with gloo (rehearsal of N > 1 on one card) they go through the host."""
    import torch.distributed as dist
    if backend == "gloo" and t.device.type == "cuda":
        h = t.cpu() if dist.get_rank() == 0 else torch.empty(t.shape, dtype=t.dtype)
        dist.broadcast(h, src=0)
        if dist.get_rank() != 0:
            t.copy_(h)
    else:
        dist.broadcast(t, src=0)


def _store(port: int, world: int, master: bool, timeout_s: float):
    import torch.distributed as dist
    return dist.TCPStore("127.0.0.1", port, world, is_master=master, timeout=datetime.timedelta(seconds=timeout_s), wait_for_workers=False)


def _init_group(backend: str, rank: int, world: int, store, device, timeout_s: float):
    import torch.distributed as dist
    kw = dict(backend=backend, rank=rank, world_size=world, store=dist.PrefixStore("pg", store), timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl":
        kw["device_id"] = torch.device(device)
    dist.init_process_group(**kw)
    # commands and status travel over CPU sockets: a worker between two maps sleeps in a socket read instead of spinning a GPU kernel
    return dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s)) if backend != "gloo" else None


def serve(rank: int, world: int, port: int, backend: str, device_index: int, runner_spec: str, tile: int, batch: int, conv_variant, timeout_s: float,
          t_spawn: float):
    """Body of a worker process (mica_amd/multi_worker.py): build the runner, join the group, obey commands until "exit"."""
    import torch.distributed as dist
    marks = {"spawned": t_spawn, "imported": time.time()}
    runner = _resolve(runner_spec)(device_index, tile, batch, conv_variant)
    marks["engine"] = time.time()
    if getattr(getattr(runner, "device", None), "type", "cpu") == "cuda":
        torch.cuda.set_device(runner.device)        # this rank's GPU is the current device for everything the group creates
    # "ready" before the group: rank 0 enters the (uninterruptible) rendezvous only when every worker stands here, so a worker that
    # died on its way up is seen by a poll of its process, not by a collective's time-out
    store = _store(port, world, False, timeout_s)
    store.set(f"ready/{rank}", "1")
    store.wait(["go"])
    ctl = _init_group(backend, rank, world, store, getattr(runner, "device", "cpu"), timeout_s)
    marks["joined"] = time.time()
    first = True
    try:
        while True:
            box = [None]
            dist.broadcast_object_list(box, src=0, group=ctl)
            cmd = box[0]
            if cmd["op"] == "exit":
                break
            status = {"rank": rank, "ok": True}
            vol = af = None
            try:                                    # phase A: everything that can fail before a collective is entered
                t0 = time.time()
                runner.load_model(cmd["model_path"])
                status["load_model_s"] = time.time() - t0
                vol = runner.empty(cmd["shape"], torch.float32)
                if cmd["af_dtype"] is not None:
                    af = runner.empty((24, *cmd["shape"]), getattr(torch, cmd["af_dtype"]))
            except Exception as ex:
                status.update(ok=False, error=f"{type(ex).__name__}: {ex}")
            got = [None] * world
            dist.all_gather_object(got, status, group=ctl)
            if not all(s["ok"] for s in got):
                del vol, af
                continue                            # rank 0 reports and closes the pool ("exit" follows, or the kill)
            status = {"rank": rank, "ok": True}
            try:                                    # phase B: the volumes and the map - every rank is inside collectives from here on
                t0 = time.time()
                _bcast(vol, backend)
                if af is not None:
                    _bcast(af, backend)
                status["broadcast_s"] = time.time() - t0
                t0 = time.time()
                stats = {}
                runner.predict(vol, af, cmd["grid"], cmd["pad"], False, cmd["gather_to_root"], False, stats)
                status["predict_s"] = time.time() - t0
                if first:
                    marks["first_map_done"] = time.time()
                    status["startup"] = dict(marks)
                    first = False
                del vol, af
            except Exception as ex:
                # the other ranks stand in a collective this rank will never enter: leaving the process is what ends their wait (gloo:
                # at once, connection reset; RCCL: when the group's time-out fires) - a status message could not reach them
                print(f"mica_amd.multi worker rank {rank}: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)
                os._exit(3)
            dist.all_gather_object(got, status, group=ctl)
    finally:
        try:
            runner.close()
        finally:
            if dist.is_initialized():
                dist.destroy_process_group()


# ---- rank 0: the pool ------------------------------------------------------------------------------------------------------------------
class RankPool:
    """N ranks for `predict(vol, af)`: this process + N-1 persistent children.  Not thread-safe; one pool per configuration."""

    def __init__(self, gpus: int, tile: int = 64, batch: int = 8, backend: str | None = None, devices=None, conv_variant=None,
                 runner: str = DEFAULT_RUNNER, timeout_s: float | None = None, force_collective: bool = False):
        self.world = int(gpus)
        self.tile, self.batch, self.conv_variant = int(tile), int(batch), conv_variant
        self.backend = backend or os.environ.get("MICA_RANK_BACKEND", "nccl")
        if devices is None:
            env = os.environ.get("MICA_RANK_DEVICES")
            devices = [int(v) for v in env.split(",")] if env else list(range(self.world))
        if len(devices) != self.world:
            raise ValueError(f"{self.world} ranks need {self.world} device indices, got {devices}")
        self.devices = list(devices)
        self.runner_spec = runner
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("MICA_RANK_TIMEOUT", "600"))
        self.force_collective = bool(force_collective)
        self.procs: list = []
        self.port = None
        self.store = None
        self.ctl = None
        self.joined = False
        self.closed = False
        self.t_spawn = None
        self.t_joined = None
        self.startup: list = []          # per worker: wall-clock marks of its start-up (first map only)
        self.last_status: list = []
        self.maps = 0

    # -- life cycle ---------------------------------------------------------------------------------------------------------------
    def spawn(self):
        """Start the N-1 worker processes (idempotent).  They import torch, build their engines and wait at the rendezvous; this
        process joins it at the first `predict` - whatever it does until then runs beside their start-up."""
        if self.procs or self.world == 1:
            return self
        self._open_store()
        self.t_spawn = time.time()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for r in range(1, self.world):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            env.pop("MICA_GPUS", None)       # a worker is one rank: it must not start a pool of its own
            argv = [sys.executable, "-m", "mica_amd.multi_worker", "--rank", str(r), "--world", str(self.world), "--port", str(self.port),
                    "--backend", self.backend, "--device", str(self.devices[r]), "--runner", self.runner_spec, "--tile", str(self.tile),
                    "--batch", str(self.batch), "--timeout", str(self.timeout_s), "--spawned", repr(self.t_spawn)]
            if self.conv_variant is not None:
                argv += ["--conv-variant", str(self.conv_variant)]
            self.procs.append(subprocess.Popen(argv, env=env, cwd=root))
        return self

    def _open_store(self):
        if self.store is None:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            self.port = s.getsockname()[1]
            s.close()
            self.store = _store(self.port, self.world, True, self.timeout_s)

    def _dead(self):
        return [(r + 1, p.returncode) for r, p in enumerate(self.procs) if p.poll() is not None]

    def _join(self, device):
        import torch.distributed as dist
        if self.joined:
            return
        if dist.is_initialized():
            raise MicaHipError("torch.distributed is already initialised in this process: its ranks belong to the caller - shard with "
                               "VolumePredictor.predict_volume_sharded on those ranks instead of gpus=N")
        self.spawn()
        self._open_store()
        # every worker says "ready" (engine built, standing in front of the rendezvous) before anybody enters it: the rendezvous itself
        # cannot be interrupted, so a worker that died on the way (import error, no such device, out of memory) is noticed HERE, by a
        # poll of its process, and not by a time-out ten minutes later
        t_end = time.time() + self.timeout_s
        while not all(self.store.check([f"ready/{r}"]) for r in range(1, self.world)):
            dead = self._dead()
            if dead or time.time() > t_end:
                self._kill()
                raise MicaHipError(f"worker rank(s) {dead} exited during start-up (rank, exit code): see their stderr above" if dead else
                                   f"workers not ready after {self.timeout_s:.0f} s")
            time.sleep(0.05)
        self.store.set("go", "1")
        try:
            self.ctl = _init_group(self.backend, 0, self.world, self.store, device, self.timeout_s)
        except Exception as ex:
            self._kill()
            raise MicaHipError(f"rendezvous of {self.world} ranks failed: {ex}")
        self.joined = True
        self.t_joined = time.time()

    def _command(self, cmd):
        import torch.distributed as dist
        dist.broadcast_object_list([cmd], src=0, group=self.ctl)

    def _statuses(self, mine):
        import torch.distributed as dist
        got = [None] * self.world
        dist.all_gather_object(got, mine, group=self.ctl)
        return got

    def predict(self, runner, model_path: str, vol: torch.Tensor, af: torch.Tensor | None, grid: int, pad: int, gather_to_root: bool = False,
                to_host: bool = True):
        """One map on all ranks.  `runner`: rank 0's compute object (its engine has the weights of `model_path` loaded, or loads them
        here); vol f32 [N0,N1,N2] / af u8 or f32 [24,N0,N1,N2] on rank 0's device.  -> the dict of four volumes (numpy with to_host)."""
        if self.closed:
            raise MicaHipError("RankPool is closed")
        dead = self._dead()
        if dead:
            self.close()
            raise MicaHipError(f"worker rank(s) {dead} have exited (rank, exit code): the pool is closed")
        try:
            self._join(runner.device)
            self._command({"op": "predict", "model_path": os.path.abspath(model_path), "shape": tuple(int(v) for v in vol.shape),
                           "af_dtype": None if af is None else str(af.dtype).replace("torch.", ""), "grid": int(grid), "pad": int(pad),
                           "gather_to_root": bool(gather_to_root)})
            mine = {"rank": 0, "ok": True}
            try:
                t0 = time.time()
                runner.load_model(model_path)
                mine["load_model_s"] = time.time() - t0
            except Exception as ex:
                mine.update(ok=False, error=f"{type(ex).__name__}: {ex}")
            got = self._statuses(mine)
            bad = [s for s in got if not s["ok"]]
            if bad:
                raise MicaHipError("; ".join(f"rank {s['rank']}: {s['error']}" for s in bad))
            t0 = time.time()
            _bcast(vol, self.backend)
            if af is not None:
                _bcast(af, self.backend)
            mine = {"rank": 0, "ok": True, "broadcast_s": time.time() - t0}
            stats = {}
            out = None
            try:
                t0 = time.time()
                out = runner.predict(vol, af, grid, pad, self.force_collective, gather_to_root, to_host, stats)
                mine["predict_s"] = time.time() - t0
                mine["stats"] = {k: (str(v) if not isinstance(v, (int, float, str, type(None))) else v) for k, v in stats.items()}
            except Exception as ex:
                mine.update(ok=False, error=f"{type(ex).__name__}: {ex}")
            got = self._statuses(mine)
            self.last_status = got
            for s in got:
                if "startup" in s:
                    self.startup.append(dict(s["startup"], rank=s["rank"]))
            bad = [s for s in got if not s["ok"]]
            if bad:
                raise MicaHipError("; ".join(f"rank {s['rank']}: {s['error']}" for s in bad))
            self.maps += 1
            return out
        except Exception:
            # after a failure nobody knows which collective a worker is standing in: the pool goes, the next map starts a fresh one
            self.close(graceful=False)
            raise

    def _kill(self):
        for p in self.procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in self.procs:
            try:
                p.wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()                      # exactly the PIDs this pool started
                p.wait()
        self.procs = []

    def close(self, graceful: bool = True):
        """Send "exit", join every worker (time-out, then terminate / kill by PID), leave the process group.  Idempotent."""
        if self.closed:
            return
        self.closed = True
        import torch.distributed as dist
        try:
            if graceful and self.joined and not self._dead():
                self._command({"op": "exit"})
                t_end = time.time() + 30
                for p in self.procs:
                    try:
                        p.wait(max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        pass
        except Exception:
            pass
        finally:
            self._kill()
            if self.joined and dist.is_initialized():
                # leaving the group can block when a collective of this rank is still pending on peers that are gone (RCCL): it runs
                # on a helper thread and the caller gets its exception / its volumes after at most 15 s either way
                def leave():
                    try:
                        dist.destroy_process_group()
                    except Exception:
                        pass
                th = threading.Thread(target=leave, name="mica-rank-leave", daemon=True)
                th.start()
                th.join(15.0)
            self.joined = False
            self.store = None
            with _LOCK:
                for k, v in list(_POOLS.items()):
                    if v is self:
                        del _POOLS[k]

    def startup_report(self) -> str:
        """What a worker's start costs, from the first map's marks (profiles/rNN_rank_startup.txt)."""
        lines = [f"ranks {self.world} backend {self.backend} devices {self.devices}"]
        for s in sorted(self.startup, key=lambda s: s["rank"]):
            t0 = s["spawned"]
            lines.append(f"rank {s['rank']}: python + torch import {s['imported'] - t0:.2f} s, engine (workspace) +{s['engine'] - s['imported']:.2f} s, "
                         f"waited at the rendezvous +{s['joined'] - s['engine']:.2f} s, first map done {s['first_map_done'] - t0:.2f} s after spawn")
        if self.t_joined is not None and self.t_spawn is not None:
            lines.append(f"rank 0 joined the rendezvous {self.t_joined - self.t_spawn:.2f} s after it spawned the workers")
        for s in self.last_status:
            lines.append(f"rank {s['rank']} last map: " + ", ".join(f"{k} {v:.3f}" for k, v in s.items() if k.endswith("_s")))
        return "\n".join(lines)


def get_pool(gpus: int, tile: int = 64, batch: int = 8, backend: str | None = None, devices=None, conv_variant=None,
             runner: str | None = None) -> RankPool:
    """The process-wide pool for this configuration (workers persist from map to map; another configuration replaces it).
    runner: "module:factory" of the per-rank compute object (default: the HIP path, `EngineRunner`; MICA_RANK_RUNNER overrides it -
    the CPU tests put a stand-in there)."""
    backend = backend or os.environ.get("MICA_RANK_BACKEND", "nccl")
    runner = runner or os.environ.get("MICA_RANK_RUNNER", DEFAULT_RUNNER)
    key = (int(gpus), int(tile), int(batch), backend, None if devices is None else tuple(devices), conv_variant, runner)
    with _LOCK:
        p = _POOLS.get(key)
        if p is not None and not p.closed and not p._dead():
            return p
        for old in list(_POOLS.values()):     # one set of workers at a time: their engines hold 37 GB each
            old.close()
        p = RankPool(gpus, tile, batch, backend, devices, conv_variant, runner)
        _POOLS[key] = p
        return p


def prestart(gpus=None, **kw):
    """Start the workers now (no-op for one GPU): called by the Solver-flow shim when getData begins, so that python + torch import and
    the engines' 37-GB workspaces come up beside the normaliser and the tilers instead of in front of the first tile."""
    n = configured_gpus(gpus)
    if n > 1:
        get_pool(n, **kw).spawn()


def shutdown():
    with _LOCK:
        pools = list(_POOLS.values())
    for p in pools:
        p.close()


try:                                               # before threading's own shutdown, like the file writers (handoff.py)
    threading._register_atexit(shutdown)
except Exception:
    import atexit
    atexit.register(shutdown)
