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
import time

import numpy as np
import torch

from . import video_io
from .engine import Engine, default_engine

BATCH = 32   # sampled frames per device call inside run(): the window holds BATCH * step decoded frames


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


def run(video_path_one: str, video_path_two: str, engine: Engine | None = None) -> int:
    """server/model.py::run, streaming: memory is bounded by ONE batch of decoded frames (BATCH sampled frames and the
    frames between them), not by the clip -- a 10-minute 720p clip held in a Python list, as a literal port would, is ~50 GB.

    Frames are decoded into a window; when the window holds BATCH sampled frames they go through ``trl_detect_embed``, the drift
    state machine is re-run over every embedding seen so far (causal: a frame's flag depends only on earlier frames, so the
    flags of the window are final), and the window's frames are handed to the decoupled writer (video_io.AsyncWriter: drawing +
    encoding on their own thread).  Clips whose container yields NV12 (a hardware decoder's output) take the device ingest
    path: the whole window is copied to the GPU through pinned memory as NV12 (1.5 B/pixel), converted to BGR there
    (``trl_ingest_nv12``), the sampled frames are analysed in place and the BGR frames come back for the writer.

    ``engine`` (optional, not in the reference's signature): the context to run on -- a multi-GPU service keeps one per device
    (service.AnalysisService(gpus=[...])); default: the process-wide engine on the current device.

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
    eng = engine or default_engine()
    nv12 = getattr(cap, "pixfmt", "bgr") == "nv12"
    write_out = os.environ.get("TRUELY_WRITE_OUTPUT", "1") != "0"
    sink = video_io.open_writer(video_path_two, fps, (width, height)) if write_out else None
    writer = video_io.AsyncWriter(sink, annotate=os.environ.get("TRUELY_ANNOTATE", "1") != "0")
    step = max(1, int(fps / 7))   # model.py:40
    frame_count = 0
    window, first = [], 0                   # decoded frames of the current window; index of window[0] in the clip
    embs, valids = [], []                   # per-window device tensors (2 KB per sampled frame)
    uploader = None

    def flush():
        """Analyse the window's sampled frames, then release the window to the writer."""
        nonlocal window, first, uploader
        if not window:
            return
        off = (-first) % step                               # first sampled frame inside the window (model.py:46)
        if nv12:
            from .ingest import Nv12Uploader
            if uploader is None:
                uploader = Nv12Uploader(eng, height, width, BATCH * step)
            if write_out:                                   # every frame of the window is converted: the writer needs them all
                bgr_dev = uploader.upload(np.stack(window), 1)
                sampled = bgr_dev[off::step].contiguous()
            else:                                           # only the sampled frames travel to the GPU at all
                bgr_dev = None
                sampled = uploader.upload(np.stack(window[off::step]), 1) if len(window) > off else None
        else:
            bgr_dev = None
            sampled = np.stack(window[off::step]) if len(window) > off else None
        notes = {}
        if sampled is not None and len(sampled) > 0:
            r = eng.detect_embed(sampled)
            embs.append(r["emb"]); valids.append(r["valid"])
            emb = torch.cat(embs); valid = torch.cat(valids)
            d = eng.drift_score(emb, valid, first + len(window), fps)   # model.py:60-66 over everything seen so far
            k0 = emb.shape[0] - r["emb"].shape[0]
            sims = d["sims"][k0:].cpu().numpy(); flags = d["flags"][k0:].cpu().numpy()
            vmask = r["valid"].cpu().numpy(); rect = r["rect"].cpu().numpy()
            for j in range(len(vmask)):
                if vmask[j] and sims[j] <= 1.5:             # a face with a previous embedding (model.py:60,67-74)
                    notes[off + j * step] = (first + off + j * step, rect[j], bool(flags[j]))
        if write_out:
            frames = bgr_dev.cpu().numpy() if nv12 else window
            for i in range(len(window)):
                writer.put(frames[i], notes.get(i))
        first += len(window)
        window = []

    # Whatever fails in the loop -- TRL_ERR_CAPACITY from a crowded frame, an allocation failure, a damaged clip -- the writer
    # thread must end and both files must be closed: a long-lived service (service.AnalysisService) would otherwise leak one
    # blocked thread and two file handles per failed request.
    try:
        while cap.isOpened():
            ret, frame = cap.read()
            if not ret:
                break
            window.append(frame)
            frame_count += 1
            if len(window) == BATCH * step:
                flush()
        flush()
    except BaseException:
        try:
            writer.close()                   # its own error, if any, must not mask the one in flight
        except Exception:  # noqa: BLE001
            pass
        raise
    finally:
        cap.release()
    writer.close()
    if frame_count == 0:    # model.py:83-85
        print("Error: No frames were processed")
        return 0
    if embs:
        d = eng.drift_score(torch.cat(embs), torch.cat(valids), frame_count, fps)   # model.py:86-95 with the final frame count
        score = int(d["score"])
    else:
        score = 0
    print(f"Total Execution Time: {time.time() - start_time} seconds")   # model.py:78-80
    return score
