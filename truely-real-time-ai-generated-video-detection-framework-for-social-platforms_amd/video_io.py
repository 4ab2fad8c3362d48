"""Frame sources / sinks for ``model.run``.

The reference decodes with ``cv2.VideoCapture`` and re-encodes every frame with
``cv2.VideoWriter(fourcc "H264")`` (server/model.py:23,35-36,43,77).  OpenCV is not installed in
the build environment, so two back-ends exist behind the same tiny interface:

* ``cv2`` when importable (real .mp4 in, annotated .mp4 out, like the reference);
* the raw "TRLV" container (uint8 BGR frames + fps header) used by the tests and the benchmark.
GPU-side decode is SURVEY section 8(f) rank 1 ("next"), not built yet.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_MAGIC = b"TRLV0001"

try:  # pragma: no cover - not available offline
    import cv2  # type: ignore
except Exception:  # noqa: BLE001
    cv2 = None


class RawReader:
    def __init__(self, path):
        self.f = open(path, "rb")
        head = self.f.read(32)
        if len(head) < 32 or head[:8] != _MAGIC:
            self.f.close()
            raise ValueError("not a TRLV file")
        self.n, self.height, self.width = struct.unpack("<III", head[8:20])
        (self.fps_f,) = struct.unpack("<d", head[20:28])
        self.i = 0

    def isOpened(self):
        return True

    def read(self):
        if self.i >= self.n:
            return False, None
        buf = self.f.read(self.height * self.width * 3)
        if len(buf) < self.height * self.width * 3:
            return False, None
        self.i += 1
        return True, np.frombuffer(buf, np.uint8).reshape(self.height, self.width, 3).copy()

    def release(self):
        self.f.close()


class RawWriter:
    def __init__(self, path, fps, size):
        self.path, self.w, self.h, self.n = path, size[0], size[1], 0
        self.f = open(path, "wb")
        self.f.write(_MAGIC + struct.pack("<IIId", 0, self.h, self.w, float(fps)) + b"\0" * 4)

    def write(self, frame):
        self.f.write(np.ascontiguousarray(frame, np.uint8).tobytes())
        self.n += 1

    def release(self):
        self.f.seek(8)
        self.f.write(struct.pack("<I", self.n))
        self.f.close()


def write_raw(path, frames: np.ndarray, fps: float):
    w = RawWriter(path, fps, (frames.shape[2], frames.shape[1]))
    for fr in frames:
        w.write(fr)
    w.release()


def open_reader(path):
    """Returns (reader, fps:int, width, height) or None if the file cannot be opened (model.py:23-29)."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == _MAGIC:
        r = RawReader(path)
        return r, int(r.fps_f), r.width, r.height
    if cv2 is not None:  # pragma: no cover
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            return None
        return cap, int(cap.get(cv2.CAP_PROP_FPS)), int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    return None


def open_writer(path, fps, size, like_raw: bool):
    if cv2 is not None and not like_raw:  # pragma: no cover
        return cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"H264"), fps, size)
    return RawWriter(path, fps, size)


def draw_box(frame: np.ndarray, x0, y0, x1, y1, color, thickness=2):
    """cv2.rectangle stand-in (outline centred on the box edge, clipped to the frame)."""
    if cv2 is not None:  # pragma: no cover
        cv2.rectangle(frame, (x0, y0), (x1, y1), color, thickness)
        return
    H, W = frame.shape[:2]
    t0, t1 = thickness // 2, thickness - thickness // 2
    def span(a, lo, hi):
        return max(lo, a - t0), min(hi, a + t1)
    ya, yb = span(y0, 0, H); yc, yd = span(y1, 0, H)
    xa, xb = span(x0, 0, W); xc, xd = span(x1, 0, W)
    xs, xe = max(0, x0 - t0), min(W, x1 + t1)
    ys, ye = max(0, y0 - t0), min(H, y1 + t1)
    frame[ya:yb, xs:xe] = color; frame[yc:yd, xs:xe] = color
    frame[ys:ye, xa:xb] = color; frame[ys:ye, xc:xd] = color


def put_text(frame, text, org, scale, color, thickness):
    if cv2 is not None:  # pragma: no cover
        cv2.putText(frame, text, org, cv2.FONT_HERSHEY_SIMPLEX, scale, color, thickness, cv2.LINE_AA)
    # without OpenCV the Hershey font is unavailable; the label is skipped (annotation is row 8(f)-2)
