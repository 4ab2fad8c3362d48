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

BATCH = 64   # sampled frames per device call inside run()


def analyze_video(frames, fps: int = 30, frame_count: int | None = None, engine: Engine | None = None,
                  batch: int | None = None, engines=None) -> dict:
    """Batched model.py:42-75,86-95 over already-sampled frames ``(n, H, W, 3)`` uint8 BGR
    (numpy or a device tensor).  ``frame_count`` = frames decoded (defaults to n*step).
    ``engines`` (a list of >= 2 contexts on the same GPU) keeps that many batches in flight (pipeline.py)."""
    eng = engine or (engines[0] if engines else default_engine())
    n = int(frames.shape[0])
    step = max(1, int(fps / 7))
    if frame_count is None:
        frame_count = n * step
    bs = batch or n
    if engines and len(engines) > 1:
        from .pipeline import detect_embed_pipelined
        outs = detect_embed_pipelined(engines, [frames[i:i + bs] for i in range(0, n, bs)])
    else:
        outs = [eng.detect_embed(frames[i:i + bs]) for i in range(0, n, bs)]
    out = {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
    d = eng.drift_score(out["emb"], out["valid"], frame_count, fps)
    out.update(score=d["score"], sims=d["sims"], flags=d["flags"], run=d["run"], hits=d["hits"])
    return out


def run(video_path_one: str, video_path_two: str) -> int:
    start_time = time.time()
    # model.py:20-22
    if not os.path.exists(video_path_one) or os.path.getsize(video_path_one) == 0:
        print(f"Error: Input video file {video_path_one} doesn't exist or is empty")
        return 0
    opened = video_io.open_reader(video_path_one)
    if opened is None:   # model.py:24-26
        print(f"Error: OpenCV couldn't open video file {video_path_one}")
        return 0
    cap, fps, width, height = opened
    if width <= 0 or height <= 0 or fps <= 0:   # model.py:30-33
        print(f"Error: Invalid video properties: width={width}, height={height}, fps={fps}")
        cap.release()
        return 0
    eng = default_engine()
    out = video_io.open_writer(video_path_two, fps, (width, height), isinstance(cap, video_io.RawReader))
    step = max(1, int(fps / 7))   # model.py:40
    frame_count = 0
    pending, pending_idx = [], []           # sampled frames waiting for the device
    held = []                               # (index, frame) in decode order, written once annotated
    embs, valids, rects = [], [], []

    def flush():
        if not pending:
            return
        r = eng.detect_embed(np.stack(pending))
        embs.append(r["emb"]); valids.append(r["valid"]); rects.append(r["rect"].cpu().numpy())
        pending.clear(); pending_idx.clear()

    while cap.isOpened():
        ret, frame = cap.read()
        if not ret:
            break
        if frame_count % step == 0:     # model.py:46
            pending.append(frame); pending_idx.append(frame_count)
            if len(pending) == BATCH:
                flush()
        held.append(frame)
        frame_count += 1
    flush()
    cap.release()
    if frame_count == 0:    # model.py:83-85
        out.release()
        print("Error: No frames were processed")
        return 0
    emb = torch.cat(embs); valid = torch.cat(valids); rect = np.concatenate(rects)
    d = eng.drift_score(emb, valid, frame_count, fps)   # model.py:60-66,86-95
    sims = d["sims"].cpu().numpy(); flags = d["flags"].cpu().numpy(); vmask = valid.cpu().numpy()
    # model.py:67-74,77: annotate sampled frames that had a previous embedding, write every frame
    for i, frame in enumerate(held):
        if i % step == 0:
            j = i // step
            if vmask[j] and sims[j] <= 1.5:
                x0, y0, x1, y1 = (int(v) for v in rect[j])
                if flags[j]:
                    video_io.draw_box(frame, x0, y0, x1, y1, (0, 0, 255), 2)
                    video_io.put_text(frame, f"AI Detected - Frame {i}", (10, 30), 1, (0, 0, 255), 2)
                else:
                    video_io.draw_box(frame, x0, y0, x1, y1, (0, 255, 0), 2)
                    video_io.put_text(frame, "Real Frame", (x0, y0 - 10), 0.5, (0, 255, 0), 2)
        out.write(frame)
    out.release()
    print(f"Total Execution Time: {time.time() - start_time} seconds")   # model.py:78-80
    return int(d["score"])
