"""Drop-in for the reference's ``server/model.py``: same module-level ``run`` signature, return
range and error convention, with the per-frame work batched onto the MI355X.

    from model import run                                  # server/server.py:35
    fake_score = run(video_path, output_path)              # server/server.py:611,856

Differences from the reference, all behind the same results:
* the models are built once per process (``engine.default_engine``), not per call (model.py:18-19);
* sampled frames are processed in batches through ``trl_detect_embed`` instead of one
  ``mtcnn.detect`` + one ``facenet_model`` call per frame (model.py:47-59);
* the cosine / run-length / score pass (model.py:60-66,86-95) runs as one device kernel over the
  time-ordered embeddings (``trl_drift_score``).
``analyze_video`` is the batched entry point the benchmark and the multi-GPU path use.
"""
from __future__ import annotations

import os
import queue
import threading
import time

import numpy as np
import torch

from . import video_io
from .engine import Engine, default_engine

WINDOW_BYTES = 128 * 720 * 1280 * 3   # sampled BGR bytes per device call inside run() when no output is written: 128 frames at 720p
                                      # (the knee of the batch sweep); 32 sampled frames per call when every
                                      # frame also goes to the encoder, which then sets the pace
BATCH = 32


def analyze_video(frames, fps: int = 30, frame_count: int | None = None, engine: Engine | None = None,
                  batch: int | None = None, engines=None, embed_group: int = 1, embed_engine: Engine | None = None) -> dict:
    """Batched model.py:42-75,86-95 over already-sampled frames ``(n, H, W, 3)`` uint8 BGR
    (numpy or a device tensor).  ``frame_count`` = frames decoded (defaults to n*step).
    ``engines`` (a list of >= 2 contexts on the same GPU) keeps that many batches in flight, driven by this one thread
    (pipeline.detect_embed_overlapped); ``embed_group`` > 1 embeds the faces of that many consecutive batches per embedder call
    on ``embed_engine`` (a further context, created on demand) -- same results, fewer and larger launches."""
    eng = engine or (engines[0] if engines else default_engine())
    n = int(frames.shape[0])
    step = max(1, int(fps / 7))
    if frame_count is None:
        frame_count = n * step
    bs = batch or n
    if engines and len(engines) > 1:
        from .pipeline import detect_embed_overlapped
        outs = detect_embed_overlapped(engines, [frames[i:i + bs] for i in range(0, n, bs)], embed_group=embed_group,
                                       embed_engine=embed_engine)
    else:
        from .pipeline import detect_embed_grouped
        outs = detect_embed_grouped(eng, [frames[i:i + bs] for i in range(0, n, bs)], embed_group)
    out = {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
    d = eng.drift_score(out["emb"], out["valid"], frame_count, fps)
    out.update(score=d["score"], sims=d["sims"], flags=d["flags"], run=d["run"], hits=d["hits"])
    return out


class _RunCtx:
    """What run() keeps between calls, cached on the engine: a second cascade context (two windows in flight), their streams, and
    the pinned / device staging buffers of the reader.  The reference rebuilds its models per call (model.py:18-19); pinning
    host memory per call would cost this path more than the analysis of a short clip (round 3: the NV12 path was 4x slower than
    the BGR one for exactly that reason)."""

    SLOTS = 4

    def __init__(self, eng: Engine):
        self.engines = [eng, eng.clone()]
        self.streams = [torch.cuda.Stream(eng.device) for _ in self.engines]
        self.lock = threading.Lock()                      # one run() at a time per engine (service workers are per GPU anyway)
        self.pinned, self.pinned_np, self.raw_dev, self.bgr_dev = [], [], [], []
        self.key = None

    def buffers(self, rows: int, row_bytes: int, yuv: bool, H: int, W: int):
        """Pinned ring of SLOTS windows of ``rows`` frames, one device staging buffer per engine (+ the converted BGR batch for 4:2:0
        sources).  Grown when a clip needs more, otherwise reused."""
        key = (rows, row_bytes, yuv, H, W)
        if self.key != key:
            dev = self.engines[0].device
            self.pinned = [torch.empty((rows, row_bytes), dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
            self.pinned_np = [t.numpy() for t in self.pinned]
            self.raw_dev = [torch.empty((rows, row_bytes), dtype=torch.uint8, device=dev) for _ in self.engines]
            self.key = key
        return self.pinned, self.pinned_np, self.raw_dev

    def close(self):
        self.pinned = self.pinned_np = self.raw_dev = []
        self.key = None
        self.engines[1].close()


class _WindowReader(threading.Thread):
    """Reads the clip into the pinned ring, one window of frames per slot, on its own thread: file reads (and, for compressed
    clips, decoding) overlap the device work of earlier windows.  Containers with fixed-size frames (TRLV, YUV4MPEG2) are read
    with positioned reads straight into pinned memory, several in parallel, and only the frames that are needed: every frame when
    the output is written, the sampled ones otherwise (model.py:46 -- `cap.read()` has to decode them all, a raw container does
    not).  Other sources go through ``cap.read()``.

    Items on ``full``: (slot, rows, first, nframes, host_frames) per window, an exception, or None at the end of the clip.
    rows = frames in the slot (sampled frames, or every frame of the window when ``all_rows``)."""

    def __init__(self, cap, step: int, win: int, slots_np, all_rows: bool, keep_host: bool, frame_shape):
        super().__init__(name="truely-reader", daemon=True)
        self.cap, self.step, self.win, self.slots_np = cap, step, win, slots_np
        self.all_rows, self.keep_host, self.frame_shape = all_rows, keep_host, frame_shape
        self.free: queue.Queue = queue.Queue()
        self.full: queue.Queue = queue.Queue()
        self.frame_count = 0
        self.stop = False
        self.random = bool(getattr(cap, "stride", 0)) and hasattr(cap, "frame_offset")

    def run(self):
        try:
            (self._run_random if self.random else self._run_sequential)()
            self.full.put(None)
        except BaseException as e:  # noqa: BLE001 - surfaced by the consumer
            self.full.put(e)

    def _slot(self):
        item = self.free.get()
        if item is None or self.stop:
            return None
        slot, ev = item
        if ev is not None:
            ev.synchronize()                              # the copy out of this pinned window has finished
        return slot

    def _run_random(self):
        import concurrent.futures
        cap, step, win = self.cap, self.step, self.win
        fd = cap.f.fileno()
        n, fb = cap.n, cap.frame_bytes
        per = win * step                                  # frames per window
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="truely-read")

        def pread(dst, i):
            got, off = 0, cap.frame_offset(i)
            mv = memoryview(dst)
            while got < fb:
                k = os.preadv(fd, [mv[got:]], off + got)
                if k <= 0:
                    raise IOError(f"short read at frame {i}")
                got += k
        try:
            for first in range(0, n, per):
                slot = self._slot()
                if slot is None:
                    return
                nfr = min(per, n - first)
                buf = self.slots_np[slot]
                host = None
                if self.keep_host:                        # BGR source with output: every frame goes to the writer (its own array: the
                    host = [np.empty(self.frame_shape, np.uint8) for _ in range(nfr)]   # slot is reused), the sampled ones to the GPU too
                    list(pool.map(lambda k: pread(host[k].reshape(-1), first + k), range(nfr)))
                    rows = (nfr + step - 1) // step
                    for r in range(rows):
                        buf[r] = host[r * step].reshape(-1)
                else:
                    idx = list(range(first, first + nfr)) if self.all_rows else list(range(first, first + nfr, step))
                    list(pool.map(lambda a: pread(buf[a[0]], a[1]), enumerate(idx)))
                    rows = len(idx)
                self.frame_count = first + nfr
                self.full.put((slot, rows, first, nfr, host))
        finally:
            pool.shutdown(wait=True)

    def _run_sequential(self):
        cap, step, win = self.cap, self.step, self.win
        per = win * step
        idx, done = 0, False
        while not done:
            slot = self._slot()
            if slot is None:
                return
            buf = self.slots_np[slot]
            first, rows, host = idx, 0, ([] if self.keep_host else None)
            while idx - first < per and cap.isOpened():
                ret, frame = cap.read()
                if not ret:
                    done = True
                    break
                if self.all_rows or idx % step == 0:
                    buf[rows] = np.asarray(frame, np.uint8).reshape(-1)
                    rows += 1
                if host is not None:
                    host.append(frame)
                idx += 1
            if not cap.isOpened():
                done = True
            self.frame_count = idx
            if idx > first:
                self.full.put((slot, rows, first, idx - first, host))


def run(video_path_one: str, video_path_two: str, engine: Engine | None = None) -> int:
    """server/model.py::run as a stream: a reader thread fills pinned windows, two cascade contexts keep two windows in flight
    (pipeline.Overlapped) with the embedder grouped over several windows, the drift state machine is carried from window to window
    on the device (``trl_drift_update``), and the annotated output is drawn and encoded on its own thread.  Memory is bounded by
    the windows in flight, not by the clip.

    Sampled frames (model.py:40,46) travel to the GPU through pinned memory: BGR as they are, 4:2:0 clips (NV12 from a hardware
    decoder, planar I420 from YUV4MPEG2 files) at 1.5 bytes per pixel with the colour conversion on the device
    (``trl_ingest_nv12`` / ``trl_ingest_i420``).  When the output stage is skipped only the sampled frames are read at all from
    containers that allow it; when it is on, every frame reaches the writer (4:2:0 clips: converted on the device, copied back).

    ``engine`` (optional, not in the reference's signature): the context to run on -- a multi-GPU service keeps one per device
    (service.AnalysisService(gpus=[...])); default: the process-wide engine on the current device.  The second context, the
    streams and the pinned buffers are created on the first call and cached on the engine.

    Environment: TRUELY_ANNOTATE=0 writes the frames without boxes / text; TRUELY_WRITE_OUTPUT=0 skips the output stage
    (benchmarking only: the server requires a non-empty file, server.py:612-627)."""
    start_time = time.time()
    # model.py:20-22
    if not os.path.exists(video_path_one) or os.path.getsize(video_path_one) == 0:
        print(f"Error: Input video file {video_path_one} doesn't exist or is empty")
        return 0
    opened = video_io.open_reader(video_path_one)
    if opened is None:   # model.py:24-26
        print(f"Error: OpenCV couldn't open video file {video_path_one}")
        print(f"       ({video_io.describe(video_path_one)})")
        return 0
    cap, fps, width, height = opened
    if width <= 0 or height <= 0 or fps <= 0:   # model.py:30-33
        print(f"Error: Invalid video properties: width={width}, height={height}, fps={fps}")
        cap.release()
        return 0
    try:
        eng = engine or default_engine()
        ctx = eng.__dict__.get("_run_ctx")
        if ctx is None:
            ctx = eng._run_ctx = _RunCtx(eng)
    except BaseException:
        cap.release()
        raise
    with ctx.lock:
        return _run_locked(ctx, cap, fps, width, height, video_path_two, start_time)


def _run_locked(ctx: _RunCtx, cap, fps: int, width: int, height: int, video_path_two: str, start_time: float) -> int:
    from .pipeline import Overlapped
    eng = ctx.engines[0]
    dev = eng.device
    write_out = os.environ.get("TRUELY_WRITE_OUTPUT", "1") != "0"
    step = max(1, int(fps / 7))   # model.py:40
    random_access = bool(getattr(cap, "stride", 0))
    pixfmt = getattr(cap, "raw_pixfmt", None) if random_access else None
    pixfmt = pixfmt or getattr(cap, "pixfmt", "bgr")
    yuv = pixfmt in ("nv12", "i420")
    row_bytes = height * width * 3 // (2 if yuv else 1)
    # window: sampled frames per device call
    if write_out:
        win = BATCH
    else:
        win = max(32, min(256, WINDOW_BYTES // (height * width * 3)))
        total = getattr(cap, "n", 0)
        if total:                                         # a short clip still gets a few windows: reading overlaps the device work
            win = max(16, min(win, -(-((total + step - 1) // step) // 4)))
    all_rows = write_out and yuv                          # 4:2:0 + output: every frame is converted on the device for the writer
    rows = win * step if all_rows else win
    pinned, pinned_np, raw_dev = ctx.buffers(rows, row_bytes, yuv, height, width)
    sink = video_io.open_writer(video_path_two, fps, (width, height)) if write_out else None
    writer = video_io.AsyncWriter(sink, annotate=os.environ.get("TRUELY_ANNOTATE", "1") != "0")
    reader = _WindowReader(cap, step, win, pinned_np, all_rows, keep_host=write_out and not yuv, frame_shape=(height, width, 3))
    for k in range(len(pinned)):
        reader.free.put((k, None))
    state = eng.drift_state()
    meta: dict = {}                                       # window index -> (first, nframes, host frames | device BGR frames)
    seen = [0]                                            # frames decoded up to the last delivered window
    last = [None]

    def on_result(i, out):
        first, nfr, frames = meta.pop(i)
        seen[0] = first + nfr
        d = eng.drift_update(state, out["emb"], out["valid"], first + nfr, fps, want_flags=write_out, sync=False)   # model.py:60-66
        last[0] = d["result"]
        if not write_out:
            return
        sims = d["sims"].cpu().numpy(); flags = d["flags"].cpu().numpy()
        vmask = out["valid"].cpu().numpy(); rect = out["rect"].cpu().numpy()
        notes = {}
        for j in range(len(vmask)):
            if vmask[j] and sims[j] <= 1.5:               # a face with a previous embedding (model.py:60,67-74)
                notes[j * step] = (first + j * step, rect[j], bool(flags[j]))
        if isinstance(frames, torch.Tensor):
            frames = frames.cpu().numpy()
        for k in range(nfr):
            writer.put(frames[k], notes.get(k))

    ov = Overlapped(ctx.engines, on_result=on_result, streams=ctx.streams, embed_group=1 if write_out else 4, collect=False)
    reader.start()
    # Whatever fails in the loop -- an allocation failure, a damaged clip -- the reader and writer threads must end and both
    # files must be closed: a long-lived service (service.AnalysisService) would otherwise leak blocked threads and file handles.
    try:
        wi = 0
        while True:
            item = reader.full.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            slot, nrows, first, nfr, host = item

            def batch(j, slot=slot, nrows=nrows, wi=wi, first=first, nfr=nfr, host=host):
                """Under engine j's stream: pinned window -> device, (4:2:0: colour conversion), the sampled BGR batch."""
                raw = raw_dev[j][:nrows]
                raw.copy_(pinned[slot][:nrows], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                reader.free.put((slot, ev))               # the reader refills the slot once this copy has finished
                if yuv:
                    bgr = eng_j_ingest(j, raw, nrows)
                    if all_rows:
                        meta[wi] = (first, nfr, bgr)
                        return bgr[::step].contiguous()
                    meta[wi] = (first, nfr, None)
                    return bgr
                meta[wi] = (first, nfr, host)
                return raw.view(nrows, height, width, 3)

            def eng_j_ingest(j, raw, nrows):
                return ctx.engines[j].ingest_nv12(raw, height, width, 1, planar=pixfmt == "i420")

            if nrows > 0:
                ov.push(batch)
                wi += 1
            else:                                         # a window without a sampled frame cannot happen (windows start on one)
                reader.free.put((slot, None))
        ov.finish()
    except BaseException:
        reader.stop = True
        reader.free.put(None)
        ov.abandon()
        try:
            writer.close()                   # its own error, if any, must not mask the one in flight
        except Exception:  # noqa: BLE001
            pass
        reader.join(timeout=30)
        cap.release()
        raise
    reader.join()
    cap.release()
    writer.close()
    frame_count = reader.frame_count
    if frame_count == 0:    # model.py:83-85
        print("Error: No frames were processed")
        return 0
    d = eng.drift_update(state, None, None, frame_count, fps)   # model.py:86-95 with the final frame count
    score = int(d["score"])
    print(f"Total Execution Time: {time.time() - start_time} seconds")   # model.py:78-80
    return score
