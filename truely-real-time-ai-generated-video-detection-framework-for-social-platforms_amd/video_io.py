"""Frame sources / sinks for ``model.run``.

The reference decodes with ``cv2.VideoCapture`` and re-encodes every frame with
``cv2.VideoWriter(fourcc "H264")`` (server/model.py:23,35-36,43,77).  OpenCV is not installed in
the build environment, so two back-ends exist behind the same tiny interface:

* ``cv2`` when importable (real .mp4 in, annotated .mp4 out, like the reference);
* the raw "TRLV" container (fps header + uint8 frames, BGR or -- what a hardware decoder hands over -- NV12) used by the
  tests and the benchmark.  NV12 clips take the device ingest path of ``model.run``;
* Motion-JPEG in AVI (``AviMjpegWriter`` / ``AviMjpegReader``, Pillow's libjpeg): the annotated OUTPUT when OpenCV is absent -- a
  bounded standard file instead of raw frames -- and the one compressed INPUT this build decodes by itself;
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
        self.data0, self.stride = 32, self.frame_bytes    # random access (model.run's reader reads only what it needs, in parallel)
        self.raw_pixfmt = self.pixfmt                     # layout of the bytes at frame_offset()
        avail = max(0, (os.path.getsize(path) - 32) // self.frame_bytes) if self.frame_bytes else 0
        self.n = min(self.n, avail)                       # a truncated file ends where its bytes end

    def frame_offset(self, i: int) -> int:
        return self.data0 + i * self.stride

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
        # random access to the planar frames as stored (I420: the device ingest takes them without repacking).  Only when every
        # frame header is the bare "FRAME\n" (checked on the first two); otherwise model.run falls back to sequential read()
        self.data0, self.stride, self.raw_pixfmt = self.header + 6, 6 + self.frame_bytes, "i420"
        ok = True
        for k in range(min(self.n, 2)):
            self.f.seek(self.header + k * self.stride)
            ok = ok and self.f.read(6) == b"FRAME\n"
        self.f.seek(self.header)
        if not ok:
            self.stride = 0

    def frame_offset(self, i: int) -> int:
        return self.data0 + i * self.stride

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


class AviMjpegWriter:
    """Motion-JPEG in an AVI 1.0 container (RIFF 'AVI ': hdrl / movi with '00dc' chunks / idx1), frames encoded with Pillow's
    libjpeg on the writer thread.  The output stage of ``run()`` when OpenCV is absent: a bounded, standard file any player opens
    (a 10-minute 720p clip is ~1.5 GB instead of the ~50 GB of raw frames).  The reference writes H.264 through
    ``cv2.VideoWriter`` (server/model.py:35-36,77); H.264 needs OpenCV -- see INTEGRATION.md.  AVI 1.0 sizes are 32-bit: frames
    past ~3.9 GB are dropped with a warning and the file is still closed properly."""
    LIMIT = 0xE8000000

    def __init__(self, path, fps, size, quality: int = 80, threads: int | None = None):
        from PIL import Image                        # noqa: F401  (fail at open time, not on the writer thread)
        import collections
        import concurrent.futures
        self.path, self.w, self.h, self.fps, self.quality = path, int(size[0]), int(size[1]), max(1, int(round(fps))), int(quality)
        self.threads = int(os.environ.get("TRUELY_ENCODE_THREADS", "1")) if threads is None else int(threads)
        self.pool = concurrent.futures.ThreadPoolExecutor(self.threads, thread_name_prefix="truely-jpeg") if self.threads > 1 else None
        self.inflight = collections.deque()
        self.f = open(path, "wb")
        self.index: list = []                        # (offset relative to 'movi', size)
        self.n = self.dropped = 0
        self.max_chunk = 0
        self.f.write(self._header(0, 0, 0))
        self.movi_start = self.f.tell() - 4          # position of the 'movi' fourcc
        self.pos = 4                                 # bytes of the movi list written so far ('movi' itself)

    def _header(self, nframes, movi_bytes, max_chunk):
        w, h, fps = self.w, self.h, self.fps
        avih = struct.pack("<IIIIIIIIII4I", 1000000 // fps, max_chunk * fps, 0, 0x10, nframes, 0, 1, max_chunk, w, h, 0, 0, 0, 0)
        strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"MJPG", 0, 0, 0, 0, 1, fps, 0, nframes, max_chunk, 0xFFFFFFFF, 0, 0, 0, w, h)
        strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
        strl = b"strl" + b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl = b"hdrl" + b"avih" + struct.pack("<I", len(avih)) + avih + b"LIST" + struct.pack("<I", len(strl)) + strl
        head = b"LIST" + struct.pack("<I", len(hdrl)) + hdrl
        idx_bytes = 8 + 16 * nframes
        riff_size = 4 + len(head) + 8 + movi_bytes + idx_bytes
        return b"RIFF" + struct.pack("<I", riff_size & 0xFFFFFFFF) + b"AVI " + head + b"LIST" + struct.pack("<I", movi_bytes) + b"movi"

    def _encode(self, frame) -> bytes:
        from PIL import Image
        import io
        a = np.ascontiguousarray(np.asarray(frame, np.uint8)[:, :, ::-1])      # BGR (cap.read()) -> RGB
        buf = io.BytesIO()
        Image.fromarray(a, "RGB").save(buf, format="JPEG", quality=self.quality, subsampling=2)
        return buf.getvalue()

    def _append(self, data: bytes):
        if self.pos > self.LIMIT:
            if not self.dropped:
                print(f"Warning: {self.path} reached the 4 GB AVI limit; further frames are not written")
            self.dropped += 1
            return
        pad = len(data) & 1
        self.f.write(b"00dc" + struct.pack("<I", len(data)) + data + (b"\0" if pad else b""))
        self.index.append((self.pos, len(data)))
        self.pos += 8 + len(data) + pad
        self.max_chunk = max(self.max_chunk, len(data))
        self.n += 1

    def write(self, frame):
        """Encode + append.  TRUELY_ENCODE_THREADS > 1 encodes on a small pool and appends in call order (at most 2 x threads
        encoded frames wait in memory); measured here Pillow's encoder holds the GIL for most of a frame (x1.3 with four
        threads, ~750 frames/s of 360p per core), so the default is the writer thread alone."""
        if self.pool is None:
            self._append(self._encode(frame))
            return
        self.inflight.append(self.pool.submit(self._encode, np.array(frame, np.uint8, copy=True)))
        while len(self.inflight) > 2 * self.threads:
            self._append(self.inflight.popleft().result())

    def release(self):
        while self.inflight:
            self._append(self.inflight.popleft().result())
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        idx = b"".join(b"00dc" + struct.pack("<III", 0x10, off, size) for off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        self.f.seek(0)
        self.f.write(self._header(self.n, self.pos, self.max_chunk))
        self.f.close()


class AviMjpegReader:
    """Reads the Motion-JPEG AVI files ``AviMjpegWriter`` (or any encoder: ``ffmpeg -c:v mjpeg out.avi``) produces: walks the
    'movi' list, decodes each '00dc' / '00db' chunk with Pillow -> BGR frames like ``cap.read()``.  The one COMPRESSED input
    this build can decode without OpenCV."""
    pixfmt = "bgr"

    def __init__(self, path):
        self.f = open(path, "rb")
        head = self.f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"AVI ":
            self.f.close()
            raise ValueError("not an AVI file")
        fsize = os.path.getsize(path)
        self.width = self.height = 0
        self.fps_f, self.n, self.codec = 0.0, 0, b""
        self.frames: list = []                       # (file offset, size)
        pos, top = 12, 0
        while pos + 8 <= fsize and top < 65536:      # (a sane file has a handful of top-level chunks: a hostile one cannot make this spin)
            top += 1
            self.f.seek(pos)
            cid, csz = struct.unpack("<4sI", self.f.read(8))
            if cid == b"LIST":
                ltype = self.f.read(4)
                if ltype == b"hdrl":
                    self._parse_hdrl(self.f.read(max(0, min(csz - 4, 1 << 20))))
                elif ltype == b"movi":
                    self._walk_movi(pos + 12, min(pos + 8 + csz, fsize))
            pos += 8 + csz + (csz & 1)
        if self.codec.upper() not in (b"MJPG", b"JPEG") or self.width <= 0 or self.height <= 0:
            self.f.close()
            raise ValueError(f"AVI video codec {self.codec!r}: only Motion-JPEG can be decoded without OpenCV")
        self.n = len(self.frames)
        self.i = 0

    def _parse_hdrl(self, b):
        p = 0
        while p + 8 <= len(b):
            cid, csz = struct.unpack_from("<4sI", b, p)
            if cid == b"LIST":
                p += 12
                continue
            body = b[p + 8:p + 8 + csz]
            if cid == b"strh" and len(body) >= 32 and body[:4] == b"vids":
                self.codec = body[4:8]
                scale, rate = struct.unpack_from("<II", body, 20)
                self.fps_f = rate / scale if scale else 0.0
            elif cid == b"strf" and len(body) >= 20 and not self.width:
                self.width, self.height = struct.unpack_from("<ii", body, 4)
                self.height = abs(self.height)
                if not self.codec.strip(b"\0"):
                    self.codec = body[16:20]
            p += 8 + csz + (csz & 1)

    def _walk_movi(self, pos, end):
        while pos + 8 <= end and len(self.frames) < 4_000_000:      # (37 h at 30 fps: anything longer is a damaged or hostile file)
            self.f.seek(pos)
            cid, csz = struct.unpack("<4sI", self.f.read(8))
            if cid == b"LIST":                       # 'rec ' groups
                pos += 12
                continue
            if cid[2:4] in (b"dc", b"db") and csz > 0 and pos + 8 + csz <= end:
                self.frames.append((pos + 8, csz))
            pos += 8 + csz + (csz & 1)

    def isOpened(self):
        return True

    def read(self):
        from PIL import Image
        import io
        if self.i >= self.n:
            return False, None
        off, size = self.frames[self.i]
        self.f.seek(off)
        try:
            im = Image.open(io.BytesIO(self.f.read(size))).convert("RGB")
        except Exception as e:  # noqa: BLE001 - a damaged frame ends the clip like a failed cap.read() -- but not silently
            print(f"Warning: frame {self.i} of {self.n} cannot be decoded ({e}): the clip is analysed up to it")
            return False, None
        self.i += 1
        a = np.asarray(im, np.uint8)
        if a.shape[0] != self.height or a.shape[1] != self.width:
            print(f"Warning: frame {self.i - 1} of {self.n} is {a.shape[1]}x{a.shape[0]}, the header says {self.width}x{self.height}: "
                  f"the clip is analysed up to it")
            return False, None
        return True, np.ascontiguousarray(a[:, :, ::-1])

    def release(self):
        self.f.close()


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
    if magic[:4] == b"RIFF" and cv2 is None:
        try:
            r = AviMjpegReader(path)
        except (ValueError, struct.error, OSError) as e:
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


def open_writer(path, fps, size, like_raw: bool = False):
    """The sink of ``run()``'s annotated output (server/model.py:35-36).  ``*.trlv``: the raw container (tests, byte-exact
    read-back).  Otherwise OpenCV's H.264 writer when importable, like the reference; without OpenCV a Motion-JPEG AVI stream
    (bounded size, plays anywhere -- but it is MJPEG/AVI whatever the file is called: H.264 needs OpenCV)."""
    if str(path).lower().endswith(".trlv"):
        return RawWriter(path, fps, size)
    if cv2 is not None:  # pragma: no cover
        return cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"H264"), fps, size)
    if not str(path).lower().endswith(".avi"):       # the server names the file *.mp4 and serves it as video/mp4: say what it really is
        print(f"Note: OpenCV is not installed, {os.path.basename(str(path))} is written as Motion-JPEG in an AVI container (not H.264)")
    return AviMjpegWriter(path, fps, size)


def draw_box(frame: np.ndarray, x0, y0, x1, y1, color, thickness=2):
    _annotate.rectangle(frame, (x0, y0), (x1, y1), color, thickness)


def put_text(frame, text, org, scale, color, thickness):
    _annotate.put_text(frame, text, org, scale, color, thickness)
