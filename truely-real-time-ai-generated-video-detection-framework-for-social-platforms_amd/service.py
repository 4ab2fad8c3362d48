"""a persistent analysis service for the FastAPI handlers.

The reference rebuilds both models inside every ``run()`` call (server/model.py:18-19) and calls
``run`` synchronously inside ``async def`` handlers (server/server.py:585,611,813,856), so one
request blocks the whole uvicorn event loop.  Here the engines (weights on the GPU, workspaces)
live for the life of the process and requests are executed by worker threads -- ONE per GPU, since a
context has one call in flight -- while handlers simply ``await``:

    service = AnalysisService()                       # at start-up: the current device
    service = AnalysisService(gpus=range(8))          # ... or every GPU of the node, one engine + worker each
    fake_score = await service.analyze(video_path, output_path)     # in /analyze-video, /analyze-combined

With several GPUs a request goes to the worker with the fewest requests queued or running (ties: lowest ordinal); requests
on one GPU run in arrival order.  Clips are independent (no data-path collective between requests), so the node serves
``len(gpus)`` clips at a time.

Error behaviour is ``run``'s: 0 for unreadable input, exceptions propagate to the handler, which maps
them to HTTP 500 exactly as today (server.py:647-652).
"""
from __future__ import annotations

import asyncio
import concurrent.futures
import threading
from typing import Callable, Iterable, Optional


class _Worker:
    """One GPU: a single worker thread (arrival order), its engine (built on first use, in that thread) and a load counter."""

    def __init__(self, gpu: Optional[int]):
        self.gpu = gpu
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"truely-analysis-{'cur' if gpu is None else gpu}")
        self.load = 0                                  # requests queued or running (guarded by the service lock)
        self.completed = 0
        self._engine = None

    def engine(self):
        if self._engine is None:
            if self.gpu is None:
                from .engine import default_engine
                self._engine = default_engine()
            else:
                import os
                import torch
                from .engine import Engine
                torch.cuda.set_device(self.gpu)
                path = os.environ.get("TRUELY_WEIGHTS")
                self._engine = Engine(open(path, "rb").read() if path else None, device=self.gpu)
        return self._engine


class AnalysisService:
    def __init__(self, run_fn: Optional[Callable[..., int]] = None, warm_up: bool = False, gpus: Optional[Iterable[int]] = None):
        """``gpus``: device ordinals to serve on (one engine + worker thread each); None = the current device only.
        ``run_fn`` (tests, other back-ends): called as ``run_fn(video_in, video_out)``, or ``run_fn(video_in, video_out, device=g)``
        when ``gpus`` is given; default = ``model.run`` on the worker's engine."""
        self._custom = run_fn
        self._workers = [_Worker(g) for g in (list(gpus) if gpus is not None else [None])]
        if not self._workers:
            raise ValueError("gpus must name at least one device")
        self._multi = gpus is not None
        self._lock = threading.Lock()
        self.completed = self.failed = 0
        if warm_up:                                   # build the engines now instead of on each GPU's first request
            for f in [w.pool.submit(w.engine) for w in self._workers]:
                f.result()

    def _job(self, w: _Worker, video_in: str, video_out: str) -> int:
        try:
            if self._custom is not None:
                score = self._custom(video_in, video_out, device=w.gpu) if self._multi else self._custom(video_in, video_out)
            else:
                from .model import run                # lazy: importing the service needs no GPU
                score = run(video_in, video_out, engine=w.engine())
            score = int(score)
        except BaseException:
            with self._lock:
                w.load -= 1
                self.failed += 1
            raise
        with self._lock:
            w.load -= 1
            w.completed += 1
            self.completed += 1
        return score

    def submit(self, video_in: str, video_out: str) -> concurrent.futures.Future:
        """Queue one clip on the least-loaded GPU; requests on a GPU run one at a time in arrival order."""
        with self._lock:
            w = min(self._workers, key=lambda x: x.load)      # min() keeps the first of equals: lowest ordinal
            w.load += 1
        try:
            return w.pool.submit(self._job, w, video_in, video_out)
        except BaseException:
            with self._lock:
                w.load -= 1
            raise

    async def analyze(self, video_in: str, video_out: str) -> int:
        """Await the score without blocking the event loop."""
        return await asyncio.wrap_future(self.submit(video_in, video_out))

    def loads(self):
        """(gpu, queued-or-running, completed) per worker: what a health endpoint would report."""
        with self._lock:
            return [(w.gpu, w.load, w.completed) for w in self._workers]

    def close(self):
        for w in self._workers:
            w.pool.shutdown(wait=True)
