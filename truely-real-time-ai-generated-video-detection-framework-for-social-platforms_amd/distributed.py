"""Multi-GPU path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Frames are independent until the drift pass (model.py:47-59 vs 60-66), so the sampled frames of a
clip are sharded contiguously in time across ranks with no data-path collective; the only exchange
is one all-gather of the per-frame result rows ``(emb f32 [512], valid, box f32 [4], rect i32 [4])``
per batch (<= 0.6 MB per rank: latency-bound), after which every rank holds the time-ordered rows and
runs the O(T) drift kernel.  For independent clips per rank (BASELINE configs[3]) no collective is
needed at all (``bench.py --mode streams``).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.distributed as dist

from .engine import Engine, default_engine

_ROW = 512 + 1 + 4 + 4      # emb | valid | box | rect (bit-cast int32)


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous time shard [lo, hi) of n sampled frames for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n: int, world: int):
    """Shard sizes every rank can compute locally: pass them as ``counts`` to skip the size exchange."""
    return [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]


class GatherHandle:
    """An all-gather of result rows in flight (``allgather_embeddings_async``).  ``wait()`` returns the time-ordered rows; with RCCL
    it makes the CURRENT STREAM wait for the collective (the host does not block), with gloo it blocks the host."""

    def __init__(self, work, out, sizes, m, extra, pay=None):
        self.work, self.out, self.sizes, self.m, self.extra = work, out, sizes, m, extra
        self.pay = pay                                   # the send buffer stays alive until the collective has been waited for

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.pay = None
        out, sizes, m = self.out, self.sizes, self.m
        if all(s == m for s in sizes):
            rows = out
        else:
            out = out.view(len(sizes), m, _ROW)
            rows = torch.cat([out[r, :sizes[r]] for r in range(len(sizes))])
        e, v = rows[:, :512].contiguous(), rows[:, 512].to(torch.uint8).contiguous()
        if not self.extra:
            return e, v
        return e, v, rows[:, 513:517].contiguous(), rows[:, 517:521].contiguous().view(torch.int32)


def allgather_embeddings_async(emb: torch.Tensor, valid: torch.Tensor, counts: Optional[Sequence[int]] = None, group=None,
                               box: Optional[torch.Tensor] = None, rect: Optional[torch.Tensor] = None) -> GatherHandle:
    """Start the all-gather of per-rank shards of result rows and return at once (``async_op=True``): a caller that consumes the
    rows one step later never stalls on a slower rank inside a step -- the per-step collective leaves the critical path
    (bench.py).  See :func:`allgather_embeddings` for the arguments."""
    world = dist.get_world_size(group)
    n_loc = int(emb.shape[0])
    if counts is None:
        n_local = torch.tensor([n_loc], dtype=torch.int64, device=emb.device)
        sizes_t = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes_t, n_local, group=group)
        sizes = [int(s.item()) for s in sizes_t]
    else:
        sizes = [int(c) for c in counts]
        if len(sizes) != world or sizes[dist.get_rank(group)] != n_loc:
            raise ValueError(f"counts {sizes} do not describe this rank's shard of {n_loc} rows")
    m = max(sizes)
    # one fused payload row: 512 floats + valid + box + rect (int32 bits carried in a float lane: a gather only copies)
    pay = torch.zeros((m, _ROW), dtype=torch.float32, device=emb.device)
    pay[:n_loc, :512] = emb
    pay[:n_loc, 512] = valid.to(torch.float32)
    if box is not None:
        pay[:n_loc, 513:517] = box
    if rect is not None:
        pay[:n_loc, 517:521] = rect.to(torch.int32).contiguous().view(torch.float32)
    out = torch.empty((world * m, _ROW), dtype=torch.float32, device=emb.device)
    work = dist.all_gather_into_tensor(out, pay, group=group, async_op=True)
    return GatherHandle(work, out, sizes, m, box is not None or rect is not None, pay)


def allgather_embeddings(emb: torch.Tensor, valid: torch.Tensor, counts: Optional[Sequence[int]] = None, group=None,
                         box: Optional[torch.Tensor] = None, rect: Optional[torch.Tensor] = None):
    """All-gather per-rank shards of result rows and return them time-ordered.

    ``counts`` = the shard size of every rank when the caller knows it (``shard_counts``: contiguous shards of a
    clip of known length): the gather is then ONE collective and no host synchronisation.  Without it the sizes
    are exchanged first (one small all-gather + a host read).  Shards may be ragged or empty; they travel padded
    to the largest shard.  Returns ``(emb, valid)`` or, when ``box``/``rect`` are given, ``(emb, valid, box, rect)``.
    """
    return allgather_embeddings_async(emb, valid, counts=counts, group=group, box=box, rect=rect).wait()


def analyze_video_sharded(frames_local, fps: int, frame_count: int, engine: Engine | None = None, group=None,
                          counts: Optional[Sequence[int]] = None) -> dict:
    """Each rank passes ITS contiguous time shard of the sampled frames; every rank returns the clip score and the
    time-ordered rows of the whole clip (emb, valid, box, rect)."""
    eng = engine or default_engine()
    local = eng.detect_embed(frames_local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        emb, valid, box, rect = allgather_embeddings(local["emb"], local["valid"], counts=counts, group=group,
                                                     box=local["box"], rect=local["rect"])
    else:
        emb, valid, box, rect = local["emb"], local["valid"], local["box"], local["rect"]
    d = eng.drift_score(emb, valid, frame_count, fps)
    return {"score": d["score"], "sims": d["sims"], "flags": d["flags"], "run": d["run"], "hits": d["hits"],
            "emb": emb, "valid": valid, "box": box, "rect": rect, "local": local}
