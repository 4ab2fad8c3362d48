"""Persistent device context of the hot path: weights uploaded once, workspaces reused.

The reference rebuilds ``MTCNN()`` and ``InceptionResnetV1(pretrained="vggface2")`` on every
``run()`` call (server/model.py:18-19); here an :class:`Engine` is created once per GPU and shared
by :class:`mtcnn.MTCNN`, :class:`inception_resnet_v1.InceptionResnetV1` and :func:`model.run`.
PyTorch is used for device memory and streams only; every kernel is in libtruely_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from . import weights as _weights


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, blob: bytes | None = None, device: int | None = None, pnet_mode: int | None = None,
                 cap_level: int | None = None, cap_frame: int | None = None, min_face_size: int = 20,
                 thresholds=(0.6, 0.7, 0.7), factor: float = 0.709, max_faces: int = 64, embed_mode: int = 0,
                 embed_precision: str | int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("truely_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.lib = _lib.load()
        cfg = _lib.TrlConfig()
        _lib.check(self.lib.trl_default_config(C.byref(cfg)))
        cfg.device = torch.cuda.current_device() if device is None else int(device)
        cfg.min_face_size = int(min_face_size)
        cfg.thr0, cfg.thr1, cfg.thr2 = (float(t) for t in thresholds)
        cfg.factor = float(factor)
        cfg.max_faces = int(max_faces)
        cfg.embed_mode = int(embed_mode)
        cfg.embed_precision = {"f32": 0, "fp32": 0, "bf16": 1, "fp16": 2, "f16": 2}.get(embed_precision, embed_precision) if isinstance(embed_precision, str) else int(embed_precision)
        if pnet_mode is not None:
            cfg.pnet_mode = int(pnet_mode)
        if cap_level:
            cfg.cap_level = int(cap_level)
        if cap_frame:
            cfg.cap_frame = int(cap_frame)
        self.cfg = cfg
        self.device = torch.device("cuda", cfg.device)
        h = C.c_void_p()
        _lib.check(self.lib.trl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        if blob is None:
            blob = _weights.synthetic_blob(0)   # no checkpoints exist offline
        self._blob = blob
        self._pending = None                 # (outputs, frames) of the call queued by detect_embed_begin
        _lib.check(self.lib.trl_load_weights(self._h, blob, len(blob)))

    def clone(self) -> "Engine":
        """A second context on the same device with the same weights and configuration (its own workspaces: contexts are the
        unit of concurrency -- one batch in flight each)."""
        c = self.cfg
        return Engine(self._blob, device=c.device, pnet_mode=c.pnet_mode, cap_level=c.cap_level, cap_frame=c.cap_frame,
                      min_face_size=c.min_face_size, thresholds=(c.thr0, c.thr1, c.thr2), factor=c.factor, max_faces=c.max_faces,
                      embed_mode=c.embed_mode, embed_precision=c.embed_precision)

    def close(self):
        for name in ("_embedder", "_run_ctx"):            # contexts / buffers cached on this engine by pipeline.Overlapped / model.run
            o = self.__dict__.pop(name, None)
            if o is not None and hasattr(o, "close"):
                o.close()
        if getattr(self, "_h", None):
            self.lib.trl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _frames(self, frames) -> torch.Tensor:
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 (n, H, W, 3) BGR")
        return frames.to(self.device, non_blocking=True).contiguous()

    # server/model.py:47 for a batch
    def mtcnn_detect(self, frames, landmarks: bool = False):
        """(boxes [n,max_faces,4], probs [n,max_faces], counts [n]) and, with ``landmarks=True``, points [n,max_faces,10]
        (x0..x4, y0..y4) as the fourth element -- `mtcnn.detect(frame, landmarks=True)`."""
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        mf = self.cfg.max_faces
        boxes = torch.empty((n, mf, 4), dtype=torch.float32, device=self.device)
        probs = torch.empty((n, mf), dtype=torch.float32, device=self.device)
        counts = torch.empty((n,), dtype=torch.int32, device=self.device)
        if landmarks:
            points = torch.empty((n, mf, 10), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.trl_mtcnn_detect_landmarks(self._h, _ptr(fr), n, H, W, _ptr(boxes), _ptr(probs), _ptr(points),
                                                           _ptr(counts), self._stream()))
            return boxes, probs, counts, points
        _lib.check(self.lib.trl_mtcnn_detect(self._h, _ptr(fr), n, H, W, _ptr(boxes), _ptr(probs), _ptr(counts), self._stream()))
        return boxes, probs, counts

    # server/model.py:59 for a batch; faces f32 (n, h, w, 3) NHWC in [0,1]
    def facenet_embed(self, faces: torch.Tensor) -> torch.Tensor:
        faces = faces.to(self.device, torch.float32).contiguous()
        n, h, w, _ = faces.shape
        emb = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_facenet_embed(self._h, _ptr(faces), n, h, w, _ptr(emb), self._stream()))
        return emb

    # server/model.py:47-59 for a batch of sampled frames
    def detect_embed(self, frames):
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        d = self.device
        out = {"box": torch.empty((n, 4), dtype=torch.float32, device=d), "prob": torch.empty((n,), dtype=torch.float32, device=d),
               "rect": torch.empty((n, 4), dtype=torch.int32, device=d), "valid": torch.empty((n,), dtype=torch.uint8, device=d),
               "emb": torch.empty((n, 512), dtype=torch.float32, device=d)}
        _lib.check(self.lib.trl_detect_embed(self._h, _ptr(fr), n, H, W, _ptr(out["box"]), _ptr(out["prob"]), _ptr(out["rect"]),
                                             _ptr(out["valid"]), _ptr(out["emb"]), self._stream()))
        return out

    # detect_embed / detect_crop split into "queue" and "finish" (trl_detect_embed_begin / _end): one host thread can keep
    # several engines busy, each on its own stream, without a thread per engine (pipeline.detect_embed_overlapped)
    def detect_embed_begin(self, frames, crop: bool = False, faces: torch.Tensor | None = None, valid: torch.Tensor | None = None):
        """Queue detect_embed (or, with ``crop=True``, detect_crop) on the current stream and return at once; the outputs are
        valid after :meth:`detect_embed_end`.  An engine holds one call in flight.  ``faces`` / ``valid`` (optional, crop mode):
        caller-owned contiguous destination buffers -- a slot of a ring whose neighbours hold other batches' crops, so several
        batches can be embedded in one call without a copy (pipeline.detect_embed_overlapped)."""
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        d = self.device
        out = {"box": torch.empty((n, 4), dtype=torch.float32, device=d), "prob": torch.empty((n,), dtype=torch.float32, device=d),
               "rect": torch.empty((n, 4), dtype=torch.int32, device=d)}
        if valid is None:
            valid = torch.empty((n,), dtype=torch.uint8, device=d)
        elif valid.shape != (n,) or valid.dtype != torch.uint8 or not valid.is_contiguous():
            raise ValueError("valid must be a contiguous uint8 (n,) tensor")
        out["valid"] = valid
        if crop:
            S = 80 if self.cfg.embed_mode == 0 else 160
            if faces is None:
                faces = torch.empty((n, S, S, 3), dtype=torch.float32, device=d)
            elif faces.shape != (n, S, S, 3) or faces.dtype != torch.float32 or not faces.is_contiguous():
                raise ValueError(f"faces must be a contiguous float32 ({n}, {S}, {S}, 3) tensor")
            out["faces"] = faces
            _lib.check(self.lib.trl_detect_crop_begin(self._h, _ptr(fr), n, H, W, _ptr(out["box"]), _ptr(out["prob"]), _ptr(out["rect"]),
                                                      _ptr(out["valid"]), _ptr(out["faces"]), self._stream()))
        else:
            out["emb"] = torch.empty((n, 512), dtype=torch.float32, device=d)
            _lib.check(self.lib.trl_detect_embed_begin(self._h, _ptr(fr), n, H, W, _ptr(out["box"]), _ptr(out["prob"]), _ptr(out["rect"]),
                                                       _ptr(out["valid"]), _ptr(out["emb"]), self._stream()))
        self._pending = (out, fr)            # the frame tensor must outlive the queued kernels
        return out

    def detect_embed_end(self):
        """Finish the call queued by :meth:`detect_embed_begin`: the one host synchronisation, the capacity check and (rarely) the re-run."""
        if self._pending is None:
            raise RuntimeError("detect_embed_end() without a call queued by detect_embed_begin()")
        out, _fr = self._pending
        try:
            _lib.check(self.lib.trl_detect_embed_end(self._h))
        finally:
            self._pending = None
        return out

    # the two halves of detect_embed: callers that embed the faces of several batches in ONE embedder call (pipeline.py)
    def detect_crop(self, frames):
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        d = self.device
        S = 80 if self.cfg.embed_mode == 0 else 160
        out = {"box": torch.empty((n, 4), dtype=torch.float32, device=d), "prob": torch.empty((n,), dtype=torch.float32, device=d),
               "rect": torch.empty((n, 4), dtype=torch.int32, device=d), "valid": torch.empty((n,), dtype=torch.uint8, device=d),
               "faces": torch.empty((n, S, S, 3), dtype=torch.float32, device=d)}
        _lib.check(self.lib.trl_detect_crop(self._h, _ptr(fr), n, H, W, _ptr(out["box"]), _ptr(out["prob"]), _ptr(out["rect"]),
                                            _ptr(out["valid"]), _ptr(out["faces"]), self._stream()))
        return out

    def embed_faces(self, faces: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
        """Embeddings of prepared crops (detect_crop's ``faces``), zero rows where ``valid`` is 0."""
        faces = faces.to(self.device, torch.float32).contiguous()
        valid = valid.to(self.device, torch.uint8).contiguous()
        n, h, w, _ = faces.shape
        emb = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_facenet_embed_masked(self._h, _ptr(faces), _ptr(valid), n, h, w, _ptr(emb), self._stream()))
        return emb

    # NV12 decoder output -> sampled BGR batch on the device (model.py:43,46)
    def ingest_nv12(self, nv12, H: int, W: int, step: int, planar: bool = False) -> torch.Tensor:
        """``planar``: the frames are I420 (Y, U, V planes) instead of NV12 (Y plane, interleaved UV)."""
        if isinstance(nv12, np.ndarray):
            nv12 = torch.from_numpy(np.ascontiguousarray(nv12))
        nv12 = nv12.to(self.device).contiguous()
        if nv12.dtype != torch.uint8 or nv12.dim() != 2 or nv12.shape[1] != H * W * 3 // 2:
            raise ValueError("nv12 must be uint8 (n, H*W*3/2)")
        n_in = int(nv12.shape[0])
        n_out = (n_in + step - 1) // step
        out = torch.empty((n_out, H, W, 3), dtype=torch.uint8, device=self.device)
        k = C.c_int()
        fn = self.lib.trl_ingest_i420 if planar else self.lib.trl_ingest_nv12
        _lib.check(fn(self._h, _ptr(nv12), n_in, H, W, int(step), _ptr(out), C.byref(k), self._stream()))
        assert k.value == n_out
        return out

    # server/model.py:60-66,86-95
    def drift_score(self, emb: torch.Tensor, valid: torch.Tensor, frame_count: int, fps: int):
        emb = emb.to(self.device, torch.float32).contiguous()
        valid = valid.to(self.device, torch.uint8).contiguous()
        n = int(valid.shape[0])
        sims = torch.empty((n,), dtype=torch.float32, device=self.device)
        flags = torch.empty((n,), dtype=torch.uint8, device=self.device)
        res = torch.zeros((4,), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.trl_drift_score(self._h, _ptr(emb), _ptr(valid), n, int(frame_count), int(fps), _ptr(sims), _ptr(flags),
                                            _ptr(res), self._stream()))
        r = res.cpu().tolist()
        return {"score": r[0], "run": r[1], "hits": r[2], "total": r[3], "sims": sims, "flags": flags}

    DRIFT_STATE_BYTES = 2064

    def drift_state(self) -> torch.Tensor:
        """A fresh carry for :meth:`drift_update` (one per clip)."""
        return torch.zeros((self.DRIFT_STATE_BYTES,), dtype=torch.uint8, device=self.device)

    def drift_update(self, state: torch.Tensor, emb: torch.Tensor | None, valid: torch.Tensor | None, frame_count: int, fps: int,
                     want_flags: bool = True, sync: bool = True):
        """model.py:60-66 for the NEXT window of a clip (``trl_drift_update``): ``state`` carries previous embedding / run / hits.
        ``emb`` = None (or empty): only the score for ``frame_count`` is recomputed.  ``sync=False`` returns device tensors only
        (``result`` = [score, run, hits, total]) and does not wait."""
        n = 0 if emb is None else int(emb.shape[0])
        d = self.device
        if n:
            emb = emb.to(d, torch.float32).contiguous(); valid = valid.to(d, torch.uint8).contiguous()
        sims = torch.empty((n,), dtype=torch.float32, device=d)
        flags = torch.empty((n,), dtype=torch.uint8, device=d) if want_flags else None
        res = torch.empty((4,), dtype=torch.int32, device=d)
        _lib.check(self.lib.trl_drift_update(self._h, _ptr(state), _ptr(emb) if n else C.c_void_p(0), _ptr(valid) if n else C.c_void_p(0), n,
                                             int(frame_count), int(fps), _ptr(sims), _ptr(flags), _ptr(res), self._stream()))
        out = {"sims": sims, "flags": flags, "result": res}
        if sync:
            r = res.cpu().tolist()
            out.update(score=r[0], run=r[1], hits=r[2], total=r[3])
        return out

    # ---- inspection hooks (parity tests) ----
    def stage_boxes(self, stage: int, frame: int, max_rows: int | None = None) -> np.ndarray:
        k = C.c_int()
        if max_rows is None:                              # every row the stage produced
            _lib.check(self.lib.trl_debug_stage_boxes(self._h, stage, frame, None, 0, C.byref(k)))
            max_rows = max(1, k.value)
        buf = np.zeros((max_rows, 5), np.float32)
        _lib.check(self.lib.trl_debug_stage_boxes(self._h, stage, frame, buf.ctypes.data_as(C.c_void_p), max_rows, C.byref(k)))
        return buf[:min(k.value, max_rows)].copy()

    def pyramid_level(self, frame, level: int) -> torch.Tensor:
        """Test hook: pyramid level of one frame as the fused PNet kernel reads it, (h, w, 3) float32."""
        fr = self._frames(frame[None] if getattr(frame, "ndim", 4) == 3 else frame)
        _, H, W, _ = fr.shape
        m = 12.0 / self.cfg.min_face_size
        out = torch.empty((int(H * m + 1) * int(W * m + 1) * 3,), dtype=torch.float32, device=self.device)
        h, w = C.c_int(), C.c_int()
        _lib.check(self.lib.trl_debug_pyramid_level(self._h, _ptr(fr), H, W, int(level), _ptr(out), C.byref(h), C.byref(w), self._stream()))
        return out[:h.value * w.value * 3].view(h.value, w.value, 3)

    def batch_capacity(self, t2_per_frame: float = 0.0, t3_per_frame: float = 0.0) -> int:
        """Test hook: set the optimistic R-/O-Net candidate capacities (per frame) and return the attempts the last call took."""
        k = C.c_int()
        _lib.check(self.lib.trl_debug_batch_capacity(self._h, float(t2_per_frame), float(t3_per_frame), C.byref(k)))
        return k.value

    def option(self, key: str, value: int):
        """Test hook (``trl_debug_option``): "rnet_chunk" / "onet_chunk" (this context), "no_fnconv" (process-wide)."""
        _lib.check(self.lib.trl_debug_option(self._h, key.encode(), int(value)))

    def nms_tiers(self, small: int = 0, full: int = 0):
        """Test hook: LDS tiers (candidates per list) of the sort + NMS kernels; lists longer than ``full`` take the
        global-memory spill tier.  Results never depend on the tiers."""
        _lib.check(self.lib.trl_debug_nms_tiers(self._h, int(small), int(full)))

    def list_stats(self) -> dict:
        """Candidate-list statistics of the last call (attempts, lists in the spill tier, capacities, largest counts)."""
        t = (C.c_longlong * 8)()
        _lib.check(self.lib.trl_debug_list_stats(self._h, t))
        keys = ("attempts", "spill_lists", "spill_used", "spill_cap", "cap_frame", "slots_per_frame", "max_level_count", "max_frame_total")
        return dict(zip(keys, (int(v) for v in t)))

    def pnet_run(self, run: int = 0):
        """Test / tuning hook: tiles per cursor fetch of the fused PNet launch (0 = automatic); > 1 exercises the halo carry."""
        _lib.check(self.lib.trl_debug_pnet_run(self._h, int(run)))

    def stage_totals(self):
        """(boxes that entered R-Net, boxes that entered O-Net) over the whole batch of the last call."""
        t = (C.c_int32 * 2)()
        _lib.check(self.lib.trl_debug_stage_totals(self._h, t))
        return int(t[0]), int(t[1])

    def poison_workspaces(self, byte: int = 0xFF):
        """Test hook: fill the activation workspaces with a byte pattern (0xFF = NaNs)."""
        _lib.check(self.lib.trl_debug_poison(self._h, int(byte)))

    def level_counts(self, frame: int):
        a = np.zeros(32, np.int32); b = np.zeros(32, np.int32)
        L = C.c_int()
        _lib.check(self.lib.trl_debug_level_counts(self._h, frame, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(L)))
        return a[:L.value].tolist(), b[:L.value].tolist()

    CAND_DTYPE = np.dtype([("box", np.float32, 4), ("score", np.float32), ("reg", np.float32, 4), ("cell", np.int32)])

    def level_keep(self, frame: int, level: int):
        """Test hook: (records in APPEND order, pick list) of one (frame, level) of the last call: the pick list holds indices into
        the records, in pick order (descending score) -- the per-level batched_nms(0.5) of detect_face()."""
        k = C.c_int()
        _lib.check(self.lib.trl_debug_level_cands(self._h, int(frame), int(level), None, 0, C.byref(k)))
        rec = np.zeros((max(1, k.value),), self.CAND_DTYPE)
        _lib.check(self.lib.trl_debug_level_cands(self._h, int(frame), int(level), rec.ctypes.data_as(C.c_void_p), len(rec), C.byref(k)))
        rec = rec[:k.value]
        _lib.check(self.lib.trl_debug_level_keep(self._h, int(frame), int(level), None, 0, C.byref(k)))
        idx = np.zeros((max(1, k.value),), np.int32)
        _lib.check(self.lib.trl_debug_level_keep(self._h, int(frame), int(level), idx.ctypes.data_as(C.c_void_p), len(idx), C.byref(k)))
        return rec, idx[:k.value]

    def level_cands(self, frame: int, level: int) -> np.ndarray:
        """Test hook: the candidate records the PNet kernel appended for (frame, level) in the last call, sorted by cell."""
        k = C.c_int()
        _lib.check(self.lib.trl_debug_level_cands(self._h, int(frame), int(level), None, 0, C.byref(k)))
        buf = np.zeros((max(1, k.value),), self.CAND_DTYPE)
        _lib.check(self.lib.trl_debug_level_cands(self._h, int(frame), int(level), buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(k)))
        rows = buf[:min(k.value, len(buf))]
        return rows[np.argsort(rows["cell"], kind="stable")].copy()

    def pnet_level(self, frame, level: int):
        fr = self._frames(frame[None] if frame.ndim == 3 else frame)
        _, H, W, _ = fr.shape
        prob = torch.empty((H * W,), dtype=torch.float32, device=self.device)
        reg = torch.empty((H * W * 4,), dtype=torch.float32, device=self.device)
        oh, ow = C.c_int(), C.c_int()
        _lib.check(self.lib.trl_debug_pnet_level(self._h, _ptr(fr), H, W, level, _ptr(prob), _ptr(reg), C.byref(oh), C.byref(ow), self._stream()))
        k = oh.value * ow.value
        return prob[:k].reshape(oh.value, ow.value), reg[:4 * k].reshape(oh.value, ow.value, 4)

    def rnet(self, crops: torch.Tensor) -> torch.Tensor:
        crops = crops.to(self.device, torch.float32).contiguous()
        out = torch.empty((crops.shape[0], 6), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_debug_rnet(self._h, _ptr(crops), crops.shape[0], _ptr(out), self._stream()))
        return out

    def onet(self, crops: torch.Tensor) -> torch.Tensor:
        crops = crops.to(self.device, torch.float32).contiguous()
        out = torch.empty((crops.shape[0], 16), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_debug_onet(self._h, _ptr(crops), crops.shape[0], _ptr(out), self._stream()))
        return out

    def front_net(self, frame, boxes: np.ndarray, net: int) -> torch.Tensor:
        """Test hook: R-Net (net=24 -> [nb, 6]) or O-Net (net=48 -> [nb, 16]) through the production front kernel + tail on the
        given boxes (x1, y1, x2, y2) of one frame."""
        fr = self._frames(frame[None] if getattr(frame, "ndim", 4) == 3 else frame)
        _, H, W, _ = fr.shape
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
        out = torch.empty((len(b), 6 if net == 24 else 16), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_debug_front_net(self._h, _ptr(fr), H, W, b.ctypes.data_as(C.c_void_p), len(b), int(net), _ptr(out), self._stream()))
        return out

    def crop_resize(self, frames, rect: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        rect = rect.to(self.device, torch.int32).contiguous(); valid = valid.to(self.device, torch.uint8).contiguous()
        out = torch.empty((n, 80, 80, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_debug_crop_resize(self._h, _ptr(fr), n, H, W, _ptr(rect), _ptr(valid), _ptr(out), self._stream()))
        return out

    def crop_aligned(self, frames, pts: torch.Tensor, valid: torch.Tensor, S: int = 160, rgb: bool = True) -> torch.Tensor:
        """Embedding mode 3's crop alone: five-point similarity alignment of each frame's face (pts [n,10] = x0..x4, y0..y4)."""
        fr = self._frames(frames)
        n, H, W, _ = fr.shape
        pts = pts.to(self.device, torch.float32).contiguous(); valid = valid.to(self.device, torch.uint8).contiguous()
        if pts.shape != (n, 10) or valid.shape != (n,):
            raise ValueError("pts must be (n, 10) and valid (n,)")
        out = torch.empty((n, S, S, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.trl_debug_crop_aligned(self._h, _ptr(fr), n, H, W, _ptr(pts), _ptr(valid), int(S), int(bool(rgb)), _ptr(out),
                                                   self._stream()))
        return out

    def levels(self, H: int, W: int) -> int:
        """Number of pyramid levels MTCNN.detect builds for an (H, W) frame (detect_face.py scale loop)."""
        m = 12.0 / self.cfg.min_face_size
        minl, k = min(H, W) * m, 0
        while minl >= 12:
            k += 1
            minl *= self.cfg.factor
        return k

    def pnet_span(self, reset: bool = False):
        """(milliseconds, launches): execution spans of the fused PNet launches of this context since the last reset, summed on
        the device (first workgroup start to last workgroup end: the duration rocprofv3 reports for the kernel)."""
        ms, k = C.c_double(), C.c_int32()
        _lib.check(self.lib.trl_debug_pnet_span(self._h, 1 if reset else 0, C.byref(ms), C.byref(k)))
        return float(ms.value), int(k.value)

    def timings(self):
        t = (C.c_float * 4)()
        _lib.check(self.lib.trl_debug_timings(self._h, t))
        k = C.c_float()
        _lib.check(self.lib.trl_debug_pnet_kernel_ms(self._h, C.byref(k)))
        return {"pnet_ms": t[0], "call_ms": t[1], "pnet_launches": int(t[2]), "pyramid_ms": t[3], "pnet_kernel_ms": float(k.value)}


_default: Engine | None = None


def default_engine() -> Engine:
    """Process-wide engine on the current device (synthetic weights unless TRUELY_WEIGHTS points at a TRLW blob)."""
    global _default
    if _default is None:
        import os
        path = os.environ.get("TRUELY_WEIGHTS")
        blob = open(path, "rb").read() if path else None
        _default = Engine(blob)
    return _default
