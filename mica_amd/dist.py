"""Tile sharding across the GPUs of one node: tiles are independent, so batches
of consecutive tiles are dealt round-robin to the ranks, each rank runs gather+forward on its own GPU,
and one fixed-size all-gather per round (RCCL over xGMI on the GPU box, gloo in the CPU tests), overlapped with
the next round's compute, brings the cropped per-tile records to every rank; the stitching rank scatters them into the
volumes.

Device-agnostic on purpose: `run_batch` and `stitch` are callables, so the rendezvous logic is
exercised by world_size-2 gloo tests on CPU with a stand-in producer.  `RecordExchange` is the
round-by-round form (`bench.py` times it step by step); `sharded_records` drives it over a whole map.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def batch_plan(T: int, batch: int):
    """[(first, count)] covering tiles 0..T-1 in the reference's order."""
    return [(f, min(batch, T - f)) for f in range(0, T, batch)]


def rank_batches(T: int, batch: int, rank: int, world: int):
    """Batches of `rank` as (round, first, count); every rank has the same number of rounds."""
    plan = batch_plan(T, batch)
    rounds = (len(plan) + world - 1) // world
    mine = []
    for r in range(rounds):
        k = r * world + rank
        mine.append((r, *plan[k]) if k < len(plan) else (r, 0, 0))
    return mine, rounds


class RecordExchange:
    """Double-buffered all-gather of fixed-size per-tile records.

    post(r, rec, layout): this rank's records of round r ([count, *rec_shape] or None) go into send slot r&1 and the
    all-gather of that slot starts (asynchronously on the collective's own stream with RCCL); then round r-1, whose
    exchange ran beside this round's kernels, is handed to `stitch(records[count], first)` on `stitch_rank`
    (None = every rank) for every rank's batch, in rank order.  `layout` = [(first, count)] per rank for round r.
    flush() finishes the round still in flight.

    xGMI moves ~0.2 GB per round at N = 8 - a few milliseconds that would otherwise sit between two 120-ms rounds on
    every rank.  With the gloo backend and device tensors (rehearsing N > 1 on one GPU) the records are staged through
    the host, synchronously."""

    def __init__(self, batch: int, rec_shape, device, stitch, dtype=torch.float32, group=None, stitch_rank: int | None = 0,
                 force_collective: bool = False, gather_to_root: bool = False):
        self.group = group
        on = dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.batch, self.stitch, self.stitch_rank = batch, stitch, stitch_rank
        self.device = torch.device(device)
        self.backend = dist.get_backend(group) if on else None
        # force_collective: take the collective path even in a group of ONE rank - the way to execute the RCCL branch
        # (device tensors, async_op, work.wait() ordering, slot reuse) on a box with a single GPU
        self.collective = self.world > 1 or (force_collective and on)
        # gather_to_root: only the stitching rank consumes the records, so only it needs them: `dist.gather` (RCCL: grouped
        # send / recv into the root) instead of the all-gather BASELINE.json's north_star names - the same bytes into rank 0 over
        # its seven xGMI links, no traffic between the other ranks, and receive buffers on one rank instead of eight
        self.to_root = bool(gather_to_root) and stitch_rank is not None
        self.root = dist.get_global_rank(group, stitch_rank) if (self.to_root and on and group is not None) else (stitch_rank or 0)
        self.send = [torch.zeros((batch, *rec_shape), dtype=dtype, device=device) for _ in range(2)]
        self.recv = None
        if self.collective and (not self.to_root or self.rank == stitch_rank):
            # one flat receive buffer per slot: [world * batch, ...]; rank rr's records are rows rr*batch ...
            self.recv = [torch.empty((self.world * batch, *rec_shape), dtype=dtype, device=device) for _ in range(2)]
        self.pending = None
        self.collectives = 0          # all-gathers issued (tests assert the collective path really ran)

    def _gather(self, slot):
        if not self.collective:
            return None
        self.collectives += 1
        if self.backend == "gloo" and self.device.type == "cuda":      # rehearsal of N > 1 on a GPU without RCCL: through the host
            if self.to_root:
                host = [torch.empty(self.send[slot].shape, dtype=self.send[slot].dtype) for _ in range(self.world)] if self.recv is not None else None
                dist.gather(self.send[slot].cpu(), gather_list=host, dst=self.root, group=self.group)
                if host is not None:
                    self.recv[slot].copy_(torch.cat(host))
                return None
            host = torch.empty(self.recv[slot].shape, dtype=self.recv[slot].dtype)
            dist.all_gather_into_tensor(host, self.send[slot].cpu(), group=self.group)
            self.recv[slot].copy_(host)
            return None
        if self.to_root:
            parts = list(self.recv[slot].split(self.batch)) if self.recv is not None else None
            return dist.gather(self.send[slot], gather_list=parts, dst=self.root, group=self.group, async_op=True)
        # the same call with RCCL (device tensors, asynchronous on the collective's stream) and in the CPU tests (gloo)
        return dist.all_gather_into_tensor(self.recv[slot], self.send[slot], group=self.group, async_op=True)

    def _finish(self):
        work, slot, layout = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        if self.stitch_rank is None or self.rank == self.stitch_rank:
            for rr, (first, count) in enumerate(layout):
                if count:
                    src = self.recv[slot][rr * self.batch:] if self.collective else self.send[slot]
                    self.stitch(src[:count], first)

    def post(self, r: int, rec, layout):
        slot = r & 1
        if rec is not None and rec.shape[0]:
            self.send[slot][:rec.shape[0]] = rec
        work = self._gather(slot)
        if self.pending is not None:
            self._finish()
        self.pending = (work, slot, layout)

    def flush(self):
        if self.pending is not None:
            self._finish()


def sharded_records(run_batch, stitch, T: int, batch: int, rec_shape, device, dtype=torch.float32, group=None,
                    stitch_rank: int | None = 0, force_collective: bool = False, stats: dict | None = None,
                    gather_to_root: bool = False):
    """run_batch(first, count) -> tensor [count, *rec_shape] on `device`;
    stitch(records [count, *rec_shape], first) is called on `stitch_rank` (None = every rank) for every
    batch of every rank, in global tile order within a round."""
    plan = batch_plan(T, batch)
    seen = {}

    def counted(rec, first):
        seen[first] = seen.get(first, 0) + int(rec.shape[0])
        stitch(rec, first)

    ex = RecordExchange(batch, rec_shape, device, counted, dtype=dtype, group=group, stitch_rank=stitch_rank,
                        force_collective=force_collective, gather_to_root=gather_to_root)
    mine, rounds = rank_batches(T, batch, ex.rank, ex.world)
    for r, first, count in mine:
        layout = [plan[r * ex.world + rr] if r * ex.world + rr < len(plan) else (0, 0) for rr in range(ex.world)]
        ex.post(r, run_batch(first, count) if count else None, layout)
    ex.flush()
    if stitch_rank is None or ex.rank == stitch_rank:
        # every batch of the map reached the stitcher exactly once, whole (a dropped or repeated round would leave a hole in
        # the volumes or overwrite a region silently)
        if seen != dict(plan):
            raise RuntimeError(f"sharded_records: stitched batches {sorted(seen.items())} != plan {plan}")
    if stats is not None:
        stats.update(collectives=ex.collectives, backend=ex.backend, world=ex.world, rounds=rounds,
                     collective="gather" if ex.to_root else "all_gather")
    return rounds
