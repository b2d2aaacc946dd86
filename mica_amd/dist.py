"""Tile sharding across the GPUs of one node: tiles are independent, so batches
of consecutive tiles are dealt round-robin to the ranks, each rank runs gather+forward on its own GPU,
and one fixed-size all-gather per round (RCCL over xGMI on the GPU box, gloo in the CPU tests), overlapped with
the next round's compute, brings the cropped per-tile records to every rank; the stitching rank scatters them into the
volumes.

Device-agnostic on purpose: `run_batch` and `stitch` are callables, so the rendezvous logic is
exercised by world_size-2 gloo tests on CPU with a stand-in producer.
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


def sharded_records(run_batch, stitch, T: int, batch: int, rec_shape, device, dtype=torch.float32, group=None,
                    stitch_rank: int | None = 0):
    """run_batch(first, count) -> tensor [count, *rec_shape] on `device`;
    stitch(records [count, *rec_shape], first) is called on `stitch_rank` (None = every rank) for every
    batch of every rank, in global tile order within a round."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine, rounds = rank_batches(T, batch, rank, world)
    plan = batch_plan(T, batch)
    # Two send/receive slots: the all-gather of round r runs (async, on the collective's own stream) while round r+1
    # computes; round r is stitched once its gather has landed.  xGMI moves ~0.2 GB per round at N = 8 - a few
    # milliseconds that would otherwise sit between two 120-ms rounds on every rank.
    send = [torch.zeros((batch, *rec_shape), dtype=dtype, device=device) for _ in range(2)]
    recv = [[torch.empty_like(send[0]) for _ in range(world)] for _ in range(2)] if world > 1 else None

    def finish(pend):
        work, r, slot = pend
        if work is not None:
            work.wait()
        if stitch_rank is None or rank == stitch_rank:
            for rr in range(world):
                k = r * world + rr
                if k < len(plan):
                    f, c = plan[k]
                    stitch((recv[slot][rr] if world > 1 else send[slot])[:c], f)

    pending = None
    for r, first, count in mine:
        slot = r & 1
        if count:
            send[slot][:count] = run_batch(first, count)
        work = dist.all_gather(recv[slot], send[slot], group=group, async_op=True) if world > 1 else None
        if pending is not None:
            finish(pending)
        pending = (work, r, slot)
    if pending is not None:
        finish(pending)
    return rounds
