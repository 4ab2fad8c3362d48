"""Host -> device frame ingest.

The reference decodes every frame on the CPU and hands BGR arrays to the detector one at a time
(``cap.read()``, server/model.py:43,46).  Here a decoder's native output -- NV12, 1.5 bytes per pixel instead of 3 --
is staged through PINNED host buffers, copied asynchronously on the engine's stream and converted to the BGR batch
``trl_detect_embed`` consumes by ``trl_ingest_nv12`` (csrc/trl_ingest.hip), which also applies the frame sampling
``i % step == 0`` on the device.  Two staging buffers alternate, so packing batch k+1 on the host overlaps the copy and
the kernels of batch k.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine


def bgr_to_nv12(frames: np.ndarray) -> np.ndarray:
    """uint8 BGR (n, H, W, 3) -> NV12 (n, H*W*3/2): BT.601 limited range, 2x2 chroma mean.  Used to synthesise decoder
    output for tests and the ingest benchmark (there is no video decoder in the build environment)."""
    fr = np.asarray(frames)
    n, H, W, _ = fr.shape
    if H % 2 or W % 4:
        raise ValueError("NV12 needs even H and W % 4 == 0")
    out = np.empty((n, H * W * 3 // 2), np.uint8)
    for i in range(n):
        b, g, r = (fr[i, :, :, k].astype(np.float32) for k in range(3))
        y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
        u = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
        v = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
        out[i, :H * W] = np.clip(np.rint(y), 0, 255).astype(np.uint8).reshape(-1)
        u2 = u.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
        v2 = v.reshape(H // 2, 2, W // 2, 2).mean(axis=(1, 3))
        uv = np.stack([u2, v2], axis=-1)
        out[i, H * W:] = np.clip(np.rint(uv), 0, 255).astype(np.uint8).reshape(-1)
    return out


class Nv12Uploader:
    """Pinned double-buffered NV12 upload + device-side colour conversion / sampling for one engine.

    ``upload(nv12_host, step)`` returns the u8 BGR device batch of frames ``0, step, 2*step, ...``; the returned tensor
    is one of two alternating device buffers, valid until the second-next call (the engine consumes it before then)."""

    def __init__(self, engine: Engine, H: int, W: int, max_frames: int, slots: int = 2):
        self.eng, self.H, self.W, self.max_frames = engine, int(H), int(W), int(max_frames)
        fb = self.H * self.W * 3 // 2
        self.pinned = [torch.empty((max_frames, fb), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev_nv12 = [torch.empty((max_frames, fb), dtype=torch.uint8, device=engine.device) for _ in range(slots)]
        self.done = [None] * slots          # event: the H2D copy out of pinned[i] has finished
        self.free = [None] * slots          # event: the conversion kernel that read dev_nv12[i] has finished
        self.count = [0] * slots
        self.copy_stream = None
        self.i = 0

    def prefetch(self, nv12_host) -> int:
        """Start the host -> device copy of the NEXT batch on the uploader's own copy stream and return its slot; the copy runs
        while the engine's stream is busy with the current batch (PCIe and kernels overlap).  ``convert(slot, step)`` then makes
        the engine's stream wait for the copy and converts."""
        n = int(nv12_host.shape[0])
        if n > self.max_frames:
            raise ValueError(f"batch of {n} frames exceeds the uploader's {self.max_frames}")
        i = self.i
        self.i = (i + 1) % len(self.pinned)
        src = nv12_host if isinstance(nv12_host, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(nv12_host))
        if self.copy_stream is None:
            self.copy_stream = torch.cuda.Stream(self.eng.device)
        if self.free[i] is not None:
            self.copy_stream.wait_event(self.free[i])      # the conversion that last read this device buffer has finished
        if not src.is_pinned():
            if self.done[i] is not None:
                self.done[i].synchronize()
            self.pinned[i][:n].copy_(src)
            src = self.pinned[i][:n]
        with torch.cuda.stream(self.copy_stream):
            self.dev_nv12[i][:n].copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.done[i] = ev
        self.count[i] = n
        return i

    def convert(self, slot: int, step: int = 1) -> torch.Tensor:
        stream = torch.cuda.current_stream(self.eng.device)
        stream.wait_event(self.done[slot])
        out = self.eng.ingest_nv12(self.dev_nv12[slot][:self.count[slot]], self.H, self.W, step)
        ev = torch.cuda.Event()
        ev.record(stream)
        self.free[slot] = ev
        return out

    def upload(self, nv12_host, step: int = 1) -> torch.Tensor:
        n = int(nv12_host.shape[0])
        if n > self.max_frames:
            raise ValueError(f"batch of {n} frames exceeds the uploader's {self.max_frames}")
        i = self.i
        self.i = (i + 1) % len(self.pinned)
        src = nv12_host if isinstance(nv12_host, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(nv12_host))
        if not src.is_pinned():                       # pageable source (a decoder's own buffers): stage through pinned memory
            if self.done[i] is not None:
                self.done[i].synchronize()            # the previous copy out of this staging buffer must be over
            self.pinned[i][:n].copy_(src)
            src = self.pinned[i][:n]
        stream = torch.cuda.current_stream(self.eng.device)
        self.dev_nv12[i][:n].copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        self.done[i] = ev
        return self.eng.ingest_nv12(self.dev_nv12[i][:n], self.H, self.W, step)
