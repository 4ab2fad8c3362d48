"""a persistent analysis service for the FastAPI handlers.

The reference rebuilds both models inside every ``run()`` call (server/model.py:18-19) and calls
``run`` synchronously inside ``async def`` handlers (server/server.py:585,611,813,856), so one
request blocks the whole uvicorn event loop.  Here the engine (weights on the GPU, workspaces)
lives for the life of the process and requests are executed by ONE worker thread (a context has one
in-flight call); handlers simply ``await``:

    service = AnalysisService()                       # at start-up
    fake_score = await service.analyze(video_path, output_path)     # in /analyze-video, /analyze-combined

Error behaviour is ``run``'s: 0 for unreadable input, exceptions propagate to the handler, which maps
them to HTTP 500 exactly as today (server.py:647-652).
"""
from __future__ import annotations

import asyncio
import concurrent.futures
import threading
from typing import Callable, Optional


class AnalysisService:
    def __init__(self, run_fn: Optional[Callable[[str, str], int]] = None, warm_up: bool = False):
        if run_fn is None:
            from .model import run as run_fn          # lazy: importing the service needs no GPU
        self._run = run_fn
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="truely-analysis")
        self._lock = threading.Lock()
        self.completed = 0
        if warm_up:                                   # build the engine now instead of on the first request
            self._pool.submit(self._warm).result()

    @staticmethod
    def _warm():
        from .engine import default_engine
        default_engine()

    def _job(self, video_in: str, video_out: str) -> int:
        score = int(self._run(video_in, video_out))
        with self._lock:
            self.completed += 1
        return score

    def submit(self, video_in: str, video_out: str) -> concurrent.futures.Future:
        """Queue one clip; requests run one at a time in arrival order."""
        return self._pool.submit(self._job, video_in, video_out)

    async def analyze(self, video_in: str, video_out: str) -> int:
        """Await the score without blocking the event loop."""
        return await asyncio.wrap_future(self.submit(video_in, video_out))

    def close(self):
        self._pool.shutdown(wait=True)
