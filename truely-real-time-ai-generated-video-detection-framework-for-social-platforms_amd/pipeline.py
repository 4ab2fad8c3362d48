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


def detect_embed_pipelined(engines: Sequence[Engine], batches: Iterable, on_result: Callable[[int, dict], None] | None = None) -> List[dict]:
    """``engines[j]`` processes batches j, j+F, j+2F, ... (F = len(engines)) on its own stream and thread.
    Returns the per-batch ``detect_embed`` results in batch order; ``on_result(i, out)`` (optional) is called in batch
    order on the calling thread as results arrive."""
    batches = list(batches)
    F = len(engines)
    if F == 0:
        raise ValueError("need at least one engine")
    if F == 1 or len(batches) <= 1:
        outs = []
        for i, b in enumerate(batches):
            out = engines[0].detect_embed(b)
            if on_result:
                on_result(i, out)
            outs.append(out)
        return outs
    dev = engines[0].device
    streams = [torch.cuda.Stream(dev) for _ in range(F)]
    qs = [queue.Queue() for _ in range(F)]

    def worker(j):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                for i in range(j, len(batches), F):
                    out = engines[j].detect_embed(batches[i])
                    streams[j].synchronize()      # the consumer uses the tensors on another stream
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
