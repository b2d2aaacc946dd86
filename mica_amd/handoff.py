"""In-process hand-off between the three mirrors of the reference's call sites (reference utils/modeler.py:673-738).

`Solver.getData` / `Solver.nnPred` chain DataPreprocessor -> GridCreator -> CryoEMPredictor through files: one normalised MRC, 24
encoding MRCs, 25 `.npz` per tile (utils/create_grids.py:159-174, dataset/dataset.py:196-216).  With INTEGRATION.md's three import
lines those call sites run unchanged, so the files are still what they ask for - but the mirrors live in one process, and what a
stage produced is still on the GPU when the next one asks for the file.  This module is the registry that lets the next stage find
it:

  * `register_file(path, tensor, header)`    DataPreprocessor: the normalised map / an encoding channel it is writing to `path`
  * `register_grids(output_dir, entry)`      GridCreator: the transposed volume it was asked to cut into `output_dir`
  * `lookup_grids(grids_path)`               CryoEMPredictor(grids_path=...): the resident volumes behind that directory, if any

plus `TileFileWriter`, which produces the reference's `.npz` tile files from the resident volume OFF the caller's critical path
(GPU gather on a side stream -> pinned buffer -> a pool of writer threads that lay the ZIP container out by hand: `np.savez`
spends most of its time in Python's zipfile machinery under the GIL).  The files are what a foreign consumer - another process,
the reference's own predictor - reads; the predictor of THIS process does not wait for them.

Nothing here computes: tensors are produced by the HIP kernels of the stages that register them.
"""
from __future__ import annotations

import io
import os
import struct
import threading
import time
import zlib
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np
import torch

_LOCK = threading.RLock()
_FILES: "OrderedDict[str, FileEntry]" = OrderedDict()
_GRIDS: "OrderedDict[str, GridEntry]" = OrderedDict()
MAX_FILE_ENTRIES = 32        # one normalised map + 24 encoding channels, with room for a second map
MAX_GRID_ENTRIES = 4         # (map, encodings) of at most two maps


def _key(path: str) -> str:
    return os.path.realpath(os.path.normpath(path))


def _stamp(path: str):
    try:
        st = os.stat(path)
        return (st.st_size, st.st_mtime_ns)
    except OSError:
        return None


# ---- files a stage is writing whose content is still on the GPU ---------------------------------------------------------------
@dataclass
class FileEntry:
    tensor: torch.Tensor                 # device tensor in the FILE's layout [nz, ny, nx] (float32, or uint8 for binary encodings)
    header: object                       # mrc.MrcHeader of the file
    done: threading.Event = field(default_factory=threading.Event)   # the background write has finished
    stamp: tuple | None = None           # (size, mtime_ns) of the file when the write finished
    error: Exception | None = None


def register_file(path: str, tensor: torch.Tensor, header, writer=None) -> FileEntry:
    """`writer()` (optional) writes the file; it runs on a background thread, and whoever needs the FILE (`wait_file`, process exit)
    joins it.  Without a writer the file is taken to exist already.  A caller that removes an older file of the same name first
    calls `drop_file(path)` before it does (that joins the older writer); an entry still registered for the path is joined here."""
    e = FileEntry(tensor, header)
    evicted = []
    with _LOCK:
        old = _FILES.pop(_key(path), None)   # a second write of the same path (a re-run, a second map in the same directory)
        if old is not None:
            evicted.append(old)
        _FILES[_key(path)] = e
        while len(_FILES) > MAX_FILE_ENTRIES:
            evicted.append(_FILES.popitem(last=False)[1])
    for old in evicted:                      # outside the lock: their writers may still be on their way to the disk - the new writer
        old.done.wait()                      # of the same path must not start before the old one has closed its file
    if writer is None:
        e.stamp = _stamp(path)
        e.done.set()
    else:
        def run():
            try:
                writer()
                e.stamp = _stamp(path)
            except Exception as ex:          # reported to whoever waits for the file
                e.error = ex
            finally:
                e.done.set()
        _file_pool().submit(run)
    return e


_FILE_POOL = None


def _file_pool():
    """A few threads for whole-volume files (an MRC writer holds a float32 copy of its volume, and a float64 one for the header
    statistics when the caller has none).  MICA_FILE_WRITERS overrides the count."""
    global _FILE_POOL
    with _LOCK:
        if _FILE_POOL is None:
            _FILE_POOL = ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("MICA_FILE_WRITERS", "3"))), thread_name_prefix="mica-file-writer")
        return _FILE_POOL


def files_under(directory: str, suffix: str = ""):
    """Paths registered by this process inside `directory` (their files may still be on their way to the disk)."""
    d = _key(directory)
    with _LOCK:
        keys = [k for k in _FILES if os.path.dirname(k) == d and k.endswith(suffix)]
    return [k for k in keys if lookup_file(k) is not None]      # not the ones whose file the caller has deleted or rewritten since


def lookup_file(path: str) -> FileEntry | None:
    """The resident content of `path`, or None if this process did not produce the file - or somebody rewrote it since."""
    with _LOCK:
        e = _FILES.get(_key(path))
    if e is None:
        return None
    if e.done.is_set() and (e.error is not None or e.stamp != _stamp(path)):
        drop_file(path)
        return None
    return e


def wait_file(path: str):
    """Before reading `path` from disk: join its background writer, if this process has one."""
    with _LOCK:
        e = _FILES.get(_key(path))
    if e is not None:
        e.done.wait()
        if e.error is not None:
            raise e.error


def drop_file(path: str):
    with _LOCK:
        e = _FILES.pop(_key(path), None)
    if e is not None:
        e.done.wait()


# ---- tile directories whose volume is still on the GPU --------------------------------------------------------------------------
@dataclass
class GridEntry:
    kind: str                            # "map" | "af3"
    volume: torch.Tensor                 # map: f32 [N0,N1,N2]; af3: u8 or f32 [24,N0,N1,N2]; indexed (x, y, z) like the tiles
    grid_size: int
    padding: int
    offset: list | None = None
    channels: tuple = ()                 # af3: names of the channels present (all 24 -> usable)
    writer: "TileFileWriter | None" = None
    marker: str | None = None            # a hidden file beside the directory, see register_grids
    inode: int | None = None             # of the directory when it was registered
    sources: tuple = ()                  # the MRC files the volume was taken from (their FileEntry copies are dropped with the entry)
    dtype: object = None                 # of the volume, kept when `release()` has given the tensor back
    _shape: tuple = ()

    @property
    def shape(self):
        return tuple(self.volume.shape[-3:]) if self.volume is not None else self._shape

    def release(self):
        """The consumer is done with the volume: drop the tensor (HBM), keep what describes the route that was taken."""
        if self.volume is not None:
            self.dtype, self._shape = self.volume.dtype, tuple(self.volume.shape[-3:])
            self.volume = None


def _unmark(e):
    try:
        if e.marker:
            os.remove(e.marker)
    except OSError:
        pass


def join_writer(writer):
    """Join a tile-file writer whose files nobody is waiting for any more; its error, if any, has nowhere to go but the log."""
    if writer is None:
        return
    try:
        writer.wait()
    except Exception as ex:
        import logging
        logging.getLogger(__name__).error(f"tile-file writer failed: {ex}")


def register_grids(output_dir: str, entry: GridEntry):
    """The entry is valid for as long as `output_dir` is the directory it was registered for AND the hidden marker file written
    beside it (in its parent, so that the directory itself lists exactly the reference's files) exists: a caller that deletes
    grids_path (utils/modeler.py:755 does after every map) or replaces the directory invalidates the hand-off, and the next
    predictor reads whatever files are there."""
    os.makedirs(output_dir, exist_ok=True)
    k = _key(output_dir)
    entry.inode = os.stat(k).st_ino
    entry.marker = os.path.join(os.path.dirname(k), f".mica_resident_{os.path.basename(k)}_{os.getpid()}_{id(entry):x}")
    with open(entry.marker, "w") as f:
        f.write("the volume behind %s is resident in process %d (mica_amd/handoff.py)\n" % (k, os.getpid()))
    with _LOCK:
        old = _GRIDS.pop(_key(output_dir), None)
        _GRIDS[_key(output_dir)] = entry
        evicted = []
        while len(_GRIDS) > MAX_GRID_ENTRIES:
            evicted.append(_GRIDS.popitem(last=False)[1])
    for e in ([old] if old is not None else []) + evicted:
        join_writer(e.writer)
        _unmark(e)


def lookup_grids(output_dir: str) -> GridEntry | None:
    with _LOCK:
        e = _GRIDS.get(_key(output_dir))
    if e is not None:
        try:
            same = os.path.exists(e.marker) and os.stat(_key(output_dir)).st_ino == e.inode
        except OSError:
            same = False
        if not same:
            drop_grids(output_dir, cancel_files=True)       # the directory was deleted or replaced under the entry
            return None
    return e


def drop_grids(output_dir: str, cancel_files: bool = False):
    """Forget the resident volume behind `output_dir` (frees its HBM); its file writer is joined - or cancelled first."""
    with _LOCK:
        e = _GRIDS.pop(_key(output_dir), None)
    if e is not None:
        if e.writer is not None:
            if cancel_files:
                e.writer.cancel()
            join_writer(e.writer)
        _unmark(e)


def flush():
    """Join every background writer (tile files and MRC files).  Registered with atexit: files a caller asked for exist when the
    process ends."""
    with _LOCK:
        fs, gs = list(_FILES.values()), list(_GRIDS.values())
    for e in fs:
        e.done.wait()
    for g in gs:
        join_writer(g.writer)


def clear():
    """Drop every entry (tests)."""
    flush()
    with _LOCK:
        for e in _GRIDS.values():
            _unmark(e)
        _FILES.clear()
        _GRIDS.clear()


def _at_exit():
    flush()
    with _LOCK:
        for e in _GRIDS.values():
            _unmark(e)


# At interpreter exit the files a caller asked for must be complete.  An ordinary atexit handler runs too late: threading's own
# shutdown (which stops concurrent.futures' pools - the writers could no longer submit - and joins non-daemon threads) comes first;
# threading._register_atexit callbacks run at the start of that shutdown, last registered first, i.e. before the pools are stopped.
try:
    threading._register_atexit(_at_exit)
except Exception:                                # an interpreter without that hook
    import atexit
    atexit.register(_at_exit)


# ---- the reference's tile files, written off the critical path -----------------------------------------------------------------------
def _npy_bytes(value) -> bytes:
    """One member of an .npz exactly as np.savez stores it (numpy.lib.format, version chosen by numpy)."""
    b = io.BytesIO()
    np.lib.format.write_array(b, np.asanyarray(value), allow_pickle=False)
    return b.getvalue()


def _npy_header(dtype, shape) -> bytes:
    b = io.BytesIO()
    np.lib.format.write_array_header_1_0(b, {"descr": np.lib.format.dtype_to_descr(np.dtype(dtype)), "fortran_order": False, "shape": tuple(shape)})
    return b.getvalue()


def _dos_time(t=None):
    tm = time.localtime(t)
    return (tm.tm_hour << 11) | (tm.tm_min << 5) | (tm.tm_sec // 2), ((max(tm.tm_year, 1980) - 1980) << 9) | (tm.tm_mon << 5) | tm.tm_mday


class NpzLayout:
    """A ZIP container with STORED members `name.npy` in the order np.savez writes them (reference utils/create_grids.py:163-174:
    grid first, then the scalars), laid out by hand: per file one CRC of the grid and one writev.  `np.load` reads it like any
    .npz (keys, dtypes and shapes are the ones np.savez produces: checked against the reference's own files in the tests); the
    fast reader of mica_amd/dataset.py finds the grid behind the first local header."""

    def __init__(self, grid_dtype, grid_shape, constant_members: dict):
        self.grid_head = _npy_header(grid_dtype, grid_shape)
        self.grid_nbytes = int(np.prod(grid_shape)) * np.dtype(grid_dtype).itemsize
        self.const = [(k, _npy_bytes(v)) for k, v in constant_members.items()]
        self._small = {}
        self.time, self.date = _dos_time()

    def _member(self, name: bytes, crc: int, size: int, offset: int):
        local = struct.pack("<IHHHHHIIIHH", 0x04034B50, 20, 0, 0, self.time, self.date, crc, size, size, len(name), 0) + name
        central = struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 20, 20, 0, 0, self.time, self.date, crc, size, size, len(name), 0, 0, 0, 0,
                              0o600 << 16, offset) + name
        return local, central

    def _small_member(self, key: str, value):
        k = (key, int(value))
        if k not in self._small:
            body = _npy_bytes(int(value))
            self._small[k] = (body, zlib.crc32(body))
        return self._small[k]

    def pieces(self, grid: np.ndarray, scalars: dict, order):
        """-> (head bytes, grid buffer, tail bytes) of the file.  `order`: member names in np.savez's keyword order; members not in
        `scalars` come from the constants."""
        gbuf = memoryview(grid).cast("B")
        assert gbuf.nbytes == self.grid_nbytes
        crc = zlib.crc32(gbuf, zlib.crc32(self.grid_head))
        const = dict(self.const)
        off = 0
        centrals, tail = [], []
        local, central = self._member(b"grid.npy", crc, len(self.grid_head) + self.grid_nbytes, off)
        head = local + self.grid_head
        centrals.append(central)
        off += len(head) + self.grid_nbytes
        for name in order:
            if name in scalars:
                body, c = self._small_member(name, scalars[name])
            else:
                body = const[name]
                c = zlib.crc32(body)
            local, central = self._member(name.encode() + b".npy", c, len(body), off)
            tail.append(local + body)
            centrals.append(central)
            off += len(local) + len(body)
        cd = b"".join(centrals)
        end = struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(centrals), len(centrals), len(cd), off, 0)
        return head, gbuf, b"".join(tail) + cd + end


SCALAR_ORDER = ("i", "j", "k", "di", "dj", "dk", "orig_shape", "grid_size", "padding", "voxel_size", "origin", "mapc", "mapr", "maps")


class TileFileWriter:
    """Writes `<dir>/<prefix>_i{i}_j{j}_k{k}.npz` for every tile of a resident volume (several channels = several directories), in
    the background.  A feeder thread gathers `chunk` tiles at a time with the HIP gather kernel on its own stream, copies them into
    one of two pinned buffers and hands one job per file to the pool; `wait()` joins, `cancel()` drops what has not been started."""

    def __init__(self, engine, volume: torch.Tensor, table: np.ndarray, targets, grid_size: int, padding: int, constant_members: dict,
                 file_dtype=np.float32, min_grid_max=None, chunk_bytes: int = 64 << 20):
        """volume: f32 / u8 [C,N0,N1,N2] on the engine's device; targets: [(directory, file prefix)] per channel."""
        self.engine, self.volume, self.table, self.targets = engine, volume, table, list(targets)
        self.grid, self.pad, self.W = grid_size, padding, grid_size + 2 * padding
        self.file_dtype = np.dtype(file_dtype)
        self.min_grid_max = min_grid_max
        self.layout = NpzLayout(self.file_dtype, (self.W,) * 3, constant_members)
        C = volume.shape[0]
        assert C == len(self.targets)
        self.chunk = max(1, min(len(table), chunk_bytes // (C * self.W ** 3 * 4)))
        self.written = 0
        self.skipped = 0
        self.error: Exception | None = None
        self._cancel = threading.Event()
        self._go = threading.Event()         # set by whoever wants the files now (release / wait / cancel), else after START_DELAY
        self._count_lock = threading.Lock()
        self._thread = threading.Thread(target=self._feed, name="mica-tile-writer", daemon=False)
        self._started = False
        self._ready = None                   # event on the caller's stream behind the kernels that produced `volume`

    def start(self):
        """Called on the thread that produced `volume`: whatever that thread has queued on its current stream (the transpose / stack
        kernels of GridCreator._device_volume run asynchronously) is ordered in front of the writer's gathers, which run on a
        private stream of the feeder thread (advisor, round 5: with `wait()` right behind `start()` the first chunks could be cut
        from a half-written volume)."""
        for d, _ in self.targets:
            os.makedirs(d, exist_ok=True)
        self._ready = torch.cuda.Event()
        self._ready.record(torch.cuda.current_stream(self.engine.device))
        self._started = True
        self._thread.start()
        return self

    # The writer yields to the caller's critical path: it holds back until the predictor mirror has its volumes and weights and enters its
    # forward loop (`release`, from CryoEMPredictor.run_inference_resident), until somebody asks for the files (`wait`), or for START_DELAY
    # seconds - a foreign consumer calls neither.  Started at once, its page-locking, its gathers and its first thousand CRCs ran beside
    # the tiler's own uploads and the predictor's weight load: 0.2 s of a 2.9-s chain on a 256^3 map.
    START_DELAY = 0.75

    def release(self):
        self._go.set()

    def cancel(self):
        self._cancel.set()
        self._go.set()

    def wait(self):
        """Join; raises the first error a writer thread met (disk full, directory removed under it ...)."""
        self._go.set()
        if self._started:
            self._thread.join()
        if self.error is not None:
            raise self.error
        return self.written

    def done(self) -> bool:
        return not self._started or not self._thread.is_alive()

    # -- one file ---------------------------------------------------------------------------------------------------------------
    def _write_one(self, host: np.ndarray, t: int, c: int, row):
        if self._cancel.is_set():
            return
        g = host[t, c]
        if self.min_grid_max is not None and g.max() < self.min_grid_max:      # the training tiler's skip (create_grids_for_normalized_map.py:78)
            with self._count_lock:
                self.skipped += 1
            return
        if self.file_dtype != np.float32:
            g = g.astype(self.file_dtype)
        i, j, k, di, dj, dk = (int(v) for v in row)
        head, body, tail = self.layout.pieces(g, {"i": i, "j": j, "k": k, "di": di, "dj": dj, "dk": dk}, SCALAR_ORDER)
        d, prefix = self.targets[c]
        name = f"{prefix}_i{i}_j{j}_k{k}.npz"
        # under a hidden name that no `*.npz` glob matches, renamed into place when complete: a reader outside this package (the
        # reference's own dataset, another process) sees a whole tile file or none
        part = os.path.join(d, f".{name}.{os.getpid()}.part")
        fd = os.open(part, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            todo = [head, body, tail]
            n = os.writev(fd, todo)
            total = len(head) + body.nbytes + len(tail)
            if n != total:                       # a short write: finish it piece by piece
                flat = head + bytes(body) + tail
                while n < total:
                    n += os.write(fd, flat[n:])
        except BaseException:
            os.close(fd)
            try:
                os.remove(part)
            except OSError:
                pass
            raise
        os.close(fd)
        os.replace(part, os.path.join(d, name))
        with self._count_lock:
            self.written += 1

    # -- feeder -----------------------------------------------------------------------------------------------------------------
    def _feed(self):
        e = self.engine
        try:
            self._go.wait(self.START_DELAY)
            torch.cuda.set_device(e.device)
            stream = torch.cuda.Stream(device=e.device)
            stream.wait_event(self._ready)       # the volume's producers, queued by the thread that called start()
            C, W, T = self.volume.shape[0], self.W, len(self.table)
            pins = [_take_pinned(self.chunk * C * W ** 3).view(self.chunk, C, W, W, W) for _ in range(2)]
            with torch.cuda.stream(stream):      # allocated on the stream that uses them: the caching allocator orders a later reuse
                devs = [torch.empty((self.chunk, C, W, W, W), dtype=torch.float32, device=e.device) for _ in range(2)]
            inflight = [[], []]
            pool = _npz_pool()
            try:
                for n, first in enumerate(range(0, T, self.chunk)):
                    if self._cancel.is_set():
                        break
                    b = n & 1
                    for f in inflight[b]:
                        f.result()               # the jobs that read this pinned buffer two chunks ago
                    count = min(self.chunk, T - first)
                    # "a ctx is not thread-safe" (include/mica_hip.h): every Engine method runs under the engine's own lock (engine.py)
                    with torch.cuda.stream(stream):
                        e.gather_tiles(self.volume, self.grid, self.pad, first, count, out=devs[b][:count])
                        pins[b][:count].copy_(devs[b][:count], non_blocking=True)
                    stream.synchronize()
                    host = pins[b].numpy()
                    inflight[b] = [pool.submit(self._write_one, host, t, c, self.table[first + t]) for t in range(count) for c in range(C)]
                for fs in inflight:
                    for f in fs:
                        f.result()
            finally:
                for fs in inflight:                  # after an error: nothing may still read the staging buffers when they go back
                    for f in fs:
                        f.exception()
                for pbuf in pins:
                    _give_pinned(pbuf)
                stream.synchronize()             # nothing of this writer is running when its staging tensors are released
        except Exception as ex:
            self.error = ex
            self._cancel.set()
        finally:
            self.volume = None                   # a finished writer does not keep a 0.5 + 3.2-GB volume alive (the GridEntry owns it)


# One pool of writer threads and a small cache of pinned staging buffers for every TileFileWriter of the process: two writers (map and
# encodings) run at the same time behind every getData, and neither a thread pool per writer nor page-locking fresh staging memory
# per map is free for the caller they run beside (hipHostMalloc serialises with the caller's own uploads).
_NPZ_POOL = None
_PINNED: list = []


def _npz_pool():
    global _NPZ_POOL
    with _LOCK:
        if _NPZ_POOL is None:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
            _NPZ_POOL = ThreadPoolExecutor(max_workers=max(2, min(8, cores - 4)), thread_name_prefix="mica-npz")
        return _NPZ_POOL


def _take_pinned(n_floats: int) -> torch.Tensor:
    with _LOCK:
        for i, t in enumerate(_PINNED):
            if t.numel() >= n_floats:
                return _PINNED.pop(i)[:n_floats]
    return torch.empty((n_floats,), dtype=torch.float32).pin_memory()


def _give_pinned(t: torch.Tensor):
    base = t._base if t._base is not None else t
    while base._base is not None:
        base = base._base
    with _LOCK:
        if len(_PINNED) < 4:
            _PINNED.append(base.view(-1))
