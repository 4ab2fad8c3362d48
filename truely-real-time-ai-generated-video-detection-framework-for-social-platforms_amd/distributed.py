"""Multi-GPU path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Frames are independent until the drift pass (model.py:47-59 vs 60-66), so the sampled frames of a
clip are sharded contiguously in time across ranks with no data-path collective; the only exchange
is one all-gather of ``(emb f32 [n_r,512], valid u8 [n_r])`` per batch (<= 0.5 MB per rank:
latency-bound), after which every rank holds the time-ordered embeddings and runs the O(T) drift
kernel.  For independent clips per rank (BASELINE config 3) no collective is needed at all.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .engine import Engine, default_engine


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous time shard [lo, hi) of n sampled frames for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_embeddings(emb: torch.Tensor, valid: torch.Tensor, counts=None, group=None):
    """All-gather variable-length shards (padded to the max shard) and return time-ordered tensors."""
    world = dist.get_world_size(group)
    n_local = torch.tensor([emb.shape[0]], dtype=torch.int64, device=emb.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    # one fused payload: 512 floats + valid flag as float per row
    pay = torch.zeros((m, 513), dtype=torch.float32, device=emb.device)
    pay[:emb.shape[0], :512] = emb
    pay[:emb.shape[0], 512] = valid.to(torch.float32)
    out = torch.empty((world * m, 513), dtype=torch.float32, device=emb.device)
    dist.all_gather_into_tensor(out, pay, group=group)
    out = out.view(world, m, 513)
    rows = torch.cat([out[r, :sizes[r]] for r in range(world)])
    return rows[:, :512].contiguous(), rows[:, 512].to(torch.uint8).contiguous()


def analyze_video_sharded(frames_local, fps: int, frame_count: int, engine: Engine | None = None, group=None) -> dict:
    """Each rank passes ITS contiguous time shard of the sampled frames; every rank returns the clip score."""
    eng = engine or default_engine()
    local = eng.detect_embed(frames_local)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        emb, valid = allgather_embeddings(local["emb"], local["valid"], group=group)
    else:
        emb, valid = local["emb"], local["valid"]
    d = eng.drift_score(emb, valid, frame_count, fps)
    return {"score": d["score"], "sims": d["sims"], "flags": d["flags"], "run": d["run"], "hits": d["hits"],
            "emb": emb, "valid": valid, "local": local}
