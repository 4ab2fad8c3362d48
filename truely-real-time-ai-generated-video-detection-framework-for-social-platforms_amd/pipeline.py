"""Several batches in flight on one GPU.

Contexts (``Engine`` instances) are independent: own workspaces, thread-local error slot.  Driving two of them
from two host threads on two HIP streams lets the narrow tail of one batch (NMS, per-candidate tails, FaceNet's
1x1-spatial layers) overlap the wide kernels of the next (+7-10 % throughput on MI355X).
The reference processes frames strictly one at a time (server/model.py:42-59); batches are independent until the
drift state machine (model.py:60-66), which consumes their embeddings in order.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, List, Sequence

import torch

from .engine import Engine


def detect_embed_grouped(engine: Engine, batches: Sequence, embed_group: int = 1) -> List[dict]:
    """``detect_embed`` of every batch on ONE engine, with the embedder called once per ``embed_group`` batches: the cascade
    and the crops run per batch (``trl_detect_crop``), the faces of a group are embedded together (``trl_facenet_embed_masked``).
    InceptionResnetV1 at the reference's 80x80 crops is ~100 small dependent launches; their fixed cost amortises over more
    faces (2.22 ms per 256 faces alone, 1.92 at 512, 1.77 at 768 per call).  Every output element is computed by the same
    accumulation chain whatever the grouping: results are bit-identical to per-batch ``detect_embed``."""
    G = max(1, int(embed_group))
    if G == 1:
        return [engine.detect_embed(b) for b in batches]
    outs: List[dict] = []
    for g0 in range(0, len(batches), G):
        part = [engine.detect_crop(b) for b in batches[g0:g0 + G]]
        emb = engine.embed_faces(torch.cat([p["faces"] for p in part]), torch.cat([p["valid"] for p in part]))
        k = 0
        for p in part:
            n = p["valid"].shape[0]
            p["emb"] = emb[k:k + n]
            k += n
            del p["faces"]
            outs.append(p)
    return outs


def detect_embed_pipelined(engines: Sequence[Engine], batches: Iterable, on_result: Callable[[int, dict], None] | None = None,
                           embed_group: int = 1) -> List[dict]:
    """``engines[j]`` processes batches j, j+F, j+2F, ... (F = len(engines)) on its own stream and thread; with
    ``embed_group`` > 1 each engine embeds the faces of that many of ITS consecutive batches in one call (see
    ``detect_embed_grouped``).  Returns the per-batch results in batch order; ``on_result(i, out)`` (optional) is called in
    batch order on the calling thread as results arrive."""
    batches = list(batches)
    F = len(engines)
    if F == 0:
        raise ValueError("need at least one engine")
    if F == 1 or len(batches) <= 1:
        outs = detect_embed_grouped(engines[0], batches, embed_group)
        if on_result:
            for i, out in enumerate(outs):
                on_result(i, out)
        return outs
    dev = engines[0].device
    streams = [torch.cuda.Stream(dev) for _ in range(F)]
    qs = [queue.Queue() for _ in range(F)]

    def worker(j):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                mine = list(range(j, len(batches), F))
                G = max(1, int(embed_group))
                for g0 in range(0, len(mine), G):
                    outs_j = detect_embed_grouped(engines[j], [batches[i] for i in mine[g0:g0 + G]], G)
                    streams[j].synchronize()      # the consumer uses the tensors on another stream
                    for out in outs_j:
                        qs[j].put(out)
        except BaseException as e:                 # surfaced by the consumer
            qs[j].put(e)

    ths = [threading.Thread(target=worker, args=(j,), daemon=True) for j in range(F)]
    for t in ths:
        t.start()
    outs = []
    try:
        for i in range(len(batches)):
            item = qs[i % F].get()
            if isinstance(item, BaseException):
                raise item
            if on_result:
                on_result(i, item)
            outs.append(item)
    finally:
        for t in ths:
            t.join()
    return outs
