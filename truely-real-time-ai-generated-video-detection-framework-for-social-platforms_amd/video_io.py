"""Frame sources / sinks for ``model.run``.

The reference decodes with ``cv2.VideoCapture`` and re-encodes every frame with
``cv2.VideoWriter(fourcc "H264")`` (server/model.py:23,35-36,43,77).  OpenCV is not installed in
the build environment, so two back-ends exist behind the same tiny interface:

* ``cv2`` when importable (real .mp4 in, annotated .mp4 out, like the reference);
* the raw "TRLV" container (fps header + uint8 frames, BGR or -- what a hardware decoder hands over -- NV12) used by the
  tests and the benchmark.  NV12 clips take the device ingest path of ``model.run``;
* YUV4MPEG2 (``.y4m``, 4:2:0 progressive): the uncompressed interchange format every decoder can emit
  (``ffmpeg -i clip.mp4 -pix_fmt yuv420p clip.y4m``).  Planes are repacked to NV12 on the fly and take the same device ingest.
``AsyncWriter`` is the decoupled, skippable annotated-output stage: drawing and encoding run on their
own thread behind a bounded queue, so the analysis loop never waits for the encoder (the reference draws and encodes inline,
server/model.py:67-77).
"""
from __future__ import annotations

import os
import queue
import struct
import threading

import numpy as np

from . import annotate as _annotate

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
        (fmt,) = struct.unpack("<I", head[28:32])
        self.pixfmt = "nv12" if fmt == 1 else "bgr"
        self.frame_bytes = self.height * self.width * 3 // (2 if fmt == 1 else 1)
        self.i = 0

    def isOpened(self):
        return True

    def read(self):
        """(ok, frame): BGR (H, W, 3) for a BGR clip, the flat NV12 frame (H*W*3/2,) for an NV12 clip."""
        if self.i >= self.n:
            return False, None
        buf = self.f.read(self.frame_bytes)
        if len(buf) < self.frame_bytes:
            return False, None
        self.i += 1
        a = np.frombuffer(buf, np.uint8)
        return True, (a.copy() if self.pixfmt == "nv12" else a.reshape(self.height, self.width, 3).copy())

    def release(self):
        self.f.close()


class Y4MReader:
    """YUV4MPEG2 4:2:0 progressive.  ``read()`` yields flat NV12 frames (Y plane, then interleaved U/V) like an NV12 TRLV clip."""
    pixfmt = "nv12"

    def __init__(self, path):
        self.f = open(path, "rb")
        line = self.f.readline(256)
        if not line.startswith(b"YUV4MPEG2 ") or not line.endswith(b"\n"):
            self.f.close()
            raise ValueError("not a YUV4MPEG2 file")
        self.width = self.height = 0
        self.fps_f = 0.0
        chroma, interlace = "420jpeg", "p"
        for tok in line[10:].split():
            t, v = chr(tok[0]), tok[1:].decode("ascii", "replace")
            if t == "W":
                self.width = int(v)
            elif t == "H":
                self.height = int(v)
            elif t == "F":
                num, _, den = v.partition(":")
                self.fps_f = int(num) / max(1, int(den or 1))
            elif t == "C":
                chroma = v
            elif t == "I":
                interlace = v
        if not chroma.startswith("420") or chroma.startswith("420p1") or interlace not in ("p", "?"):
            self.f.close()
            raise ValueError(f"unsupported YUV4MPEG2 stream (chroma {chroma}, interlace {interlace}): 8-bit 4:2:0 progressive only")
        if self.width <= 0 or self.height <= 0 or self.width % 4 or self.height % 2:
            self.f.close()
            raise ValueError(f"YUV4MPEG2 {self.width}x{self.height}: the NV12 ingest needs W % 4 == 0 and even H")
        self.header = len(line)
        self.ysize, self.csize = self.width * self.height, (self.width // 2) * (self.height // 2)
        self.frame_bytes = self.ysize + 2 * self.csize
        self.n = max(0, (os.path.getsize(path) - self.header) // (6 + self.frame_bytes))   # "FRAME\n" + planes (no frame params)
        self.i = 0

    def isOpened(self):
        return True

    def read(self):
        tag = self.f.readline(256)
        if not tag.startswith(b"FRAME"):
            return False, None
        buf = self.f.read(self.frame_bytes)
        if len(buf) < self.frame_bytes:
            return False, None
        self.i += 1
        a = np.frombuffer(buf, np.uint8)
        out = np.empty(self.frame_bytes, np.uint8)
        out[:self.ysize] = a[:self.ysize]
        out[self.ysize::2] = a[self.ysize:self.ysize + self.csize]          # U
        out[self.ysize + 1::2] = a[self.ysize + self.csize:]                # V
        return True, out

    def release(self):
        self.f.close()


def write_y4m(path, nv12_frames: np.ndarray, fps: int, size):
    """Test / tooling helper: flat NV12 frames -> a YUV4MPEG2 file (planar 4:2:0)."""
    w, h = size
    ys = w * h
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{int(fps)}:1 Ip A1:1 C420jpeg\n".encode())
        for fr in nv12_frames:
            fr = np.asarray(fr, np.uint8).reshape(-1)
            f.write(b"FRAME\n")
            f.write(fr[:ys].tobytes()); f.write(fr[ys::2].tobytes()); f.write(fr[ys + 1::2].tobytes())


class RawWriter:
    def __init__(self, path, fps, size, pixfmt: str = "bgr"):
        self.path, self.w, self.h, self.n = path, size[0], size[1], 0
        self.f = open(path, "wb")
        self.f.write(_MAGIC + struct.pack("<IIIdI", 0, self.h, self.w, float(fps), 1 if pixfmt == "nv12" else 0))

    def write(self, frame):
        self.f.write(np.ascontiguousarray(frame, np.uint8).tobytes())
        self.n += 1

    def release(self):
        self.f.seek(8)
        self.f.write(struct.pack("<I", self.n))
        self.f.close()


def write_raw(path, frames: np.ndarray, fps: float, pixfmt: str = "bgr", size=None):
    """BGR clip: frames (n, H, W, 3).  NV12 clip: frames (n, H*W*3/2) plus size=(W, H)."""
    w = RawWriter(path, fps, size or (frames.shape[2], frames.shape[1]), pixfmt)
    for fr in frames:
        w.write(fr)
    w.release()


class AsyncWriter:
    """Decoupled annotated-output stage.  ``put(frame, note)`` hands a frame (and, for sampled frames that were compared with
    their predecessor, ``note = (frame_index, rect, flagged)``) to a worker thread that draws (annotate.py) and encodes.
    ``annotate=False`` skips the drawing, ``sink=None`` skips the stage altogether (benchmarks); the queue is bounded, so a slow
    encoder applies back-pressure instead of growing memory."""

    def __init__(self, sink, annotate: bool = True, depth: int = 64):
        self.sink, self.annotate, self.frames = sink, annotate, 0
        self.err = None
        self.q = queue.Queue(maxsize=depth)
        self.th = threading.Thread(target=self._work, name="truely-writer", daemon=True) if sink is not None else None
        if self.th:
            self.th.start()

    def _work(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.err is not None:
                continue                                  # keep draining so the producer never blocks on a dead stage
            try:
                frame, note = item
                if note is not None and self.annotate:
                    _annotate.annotate(frame, *note)
                self.sink.write(frame)
            except BaseException as e:                    # surfaced by close()
                self.err = e

    def put(self, frame, note=None):
        self.frames += 1
        if self.th:
            self.q.put((frame, note))

    def close(self):
        if self.th:
            self.q.put(None)
            self.th.join()
        if self.sink is not None:
            self.sink.release()
        if self.err is not None:
            raise self.err


def open_reader(path):
    """Returns (reader, fps:int, width, height) or None if the file cannot be opened (model.py:23-29)."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == _MAGIC:
        r = RawReader(path)
        return r, int(r.fps_f), r.width, r.height
    if magic == b"YUV4MPEG":
        try:
            r = Y4MReader(path)
        except ValueError as e:
            print(f"Error: {e}")
            return None
        return r, int(r.fps_f), r.width, r.height
    if cv2 is not None:  # pragma: no cover
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            return None
        return cap, int(cap.get(cv2.CAP_PROP_FPS)), int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    return None


def describe(path) -> str:
    """One line about a clip this build cannot decode (no OpenCV): what the container says it is."""
    from . import mp4probe
    try:
        i = mp4probe.probe(path)
    except Exception:  # noqa: BLE001 - a damaged container is simply "unknown"
        i = None
    if i is None:
        return "unknown container"
    prof = {66: "Baseline", 77: "Main", 88: "Extended", 100: "High"}.get(i.profile_idc, str(i.profile_idc))
    return (f"H.264 {prof}@L{i.level_idc / 10:.1f} {i.width}x{i.height}, {i.fps:.3f} fps, {i.frame_count} frames"
            f"{' (fragmented mp4)' if i.fragmented else ''}: decoding needs opencv-python (cv2.VideoCapture), or hand run() decoder output as an NV12 TRLV clip")


def open_writer(path, fps, size, like_raw: bool):
    if cv2 is not None and not like_raw:  # pragma: no cover
        return cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"H264"), fps, size)
    return RawWriter(path, fps, size)


def draw_box(frame: np.ndarray, x0, y0, x1, y1, color, thickness=2):
    _annotate.rectangle(frame, (x0, y0), (x1, y1), color, thickness)


def put_text(frame, text, org, scale, color, thickness):
    _annotate.put_text(frame, text, org, scale, color, thickness)
