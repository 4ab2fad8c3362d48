"""ctypes front-end of the CPU oracle (oracle/trl_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.  PARITY UNPINNED (see trl_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtrl_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "trl_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["clean", "all"] if force else []))
    return _SO


class Params(C.Structure):
    _fields_ = [("min_face_size", C.c_int), ("thr0", C.c_float), ("thr1", C.c_float), ("thr2", C.c_float),
                ("factor", C.c_double)]


class Trace(C.Structure):
    _fields_ = [("max_boxes", C.c_int), ("n_scales", C.c_int), ("n_cand_scale", C.c_int * 32),
                ("n_keep_scale", C.c_int * 32),
                ("n1", C.c_int), ("boxes1", C.POINTER(C.c_float)),
                ("n2", C.c_int), ("boxes2", C.POINTER(C.c_float)),
                ("n3", C.c_int), ("boxes3", C.POINTER(C.c_float)),
                ("points3", C.POINTER(C.c_float))]


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, blob: bytes, threads: int | None = None):
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_default_params.restype = Params
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_dot512.restype = C.c_float
        L.orc_dot512.argtypes = [C.POINTER(C.c_float)] * 2
        self.ctx = L.orc_create(blob, len(blob))
        if not self.ctx:
            raise RuntimeError(L.orc_last_error().decode())
        if threads:
            L.orc_set_threads(int(threads))
        self.params = L.orc_default_params()

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.orc_destroy(C.c_void_p(self.ctx))
            self.ctx = None

    # ---- primitives ----
    def expf(self, x: float) -> float:
        return float(self.lib.orc_expf(C.c_float(x)))

    def dot512(self, a, b) -> float:
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        return float(self.lib.orc_dot512(_p(a), _p(b)))

    def scales(self, H, W, minsize=20, factor=0.709):
        sc = (C.c_double * 32)(); hs = (C.c_int * 32)(); ws = (C.c_int * 32)()
        n = self.lib.orc_scales(H, W, minsize, C.c_double(factor), sc, hs, ws, 32)
        return [(sc[i], hs[i], ws[i]) for i in range(n)]

    def area_resample_norm(self, img, y0, y1, x0, x1, oh, ow):
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape[:2]
        out = np.empty((oh, ow, 3), np.float32)
        self.lib.orc_area_resample_norm(_p(img, C.c_uint8), H, W, int(y0), int(y1), int(x0), int(x1), int(oh), int(ow), _p(out))
        return out

    def pnet_level(self, lvl):
        lvl = np.ascontiguousarray(lvl, np.float32); h, w = lvl.shape[:2]
        oh0, ow0 = (h - 1) // 2 - 4, (w - 1) // 2 - 4
        prob = np.empty((oh0 * ow0,), np.float32); reg = np.empty((oh0 * ow0, 4), np.float32)
        oh = C.c_int(); ow = C.c_int()
        self.lib.orc_pnet_level(C.c_void_p(self.ctx), _p(lvl), h, w, _p(prob), _p(reg), C.byref(oh), C.byref(ow))
        assert (oh.value, ow.value) == (oh0, ow0)
        return prob.reshape(oh0, ow0), reg.reshape(oh0, ow0, 4)

    def rnet(self, crops):
        crops = np.ascontiguousarray(crops, np.float32); n = crops.shape[0]
        prob = np.empty((n,), np.float32); reg = np.empty((n, 4), np.float32)
        self.lib.orc_rnet(C.c_void_p(self.ctx), _p(crops), n, _p(prob), _p(reg))
        return prob, reg

    def onet(self, crops):
        crops = np.ascontiguousarray(crops, np.float32); n = crops.shape[0]
        prob = np.empty((n,), np.float32); reg = np.empty((n, 4), np.float32); pts = np.empty((n, 10), np.float32)
        self.lib.orc_onet(C.c_void_p(self.ctx), _p(crops), n, _p(prob), _p(reg), _p(pts))
        return prob, reg, pts

    def nms_iou(self, boxes, scores, thr):
        boxes = np.ascontiguousarray(boxes, np.float32); scores = np.ascontiguousarray(scores, np.float32)
        keep = np.empty((max(1, len(scores)),), np.int32)
        n = self.lib.orc_nms_iou(_p(boxes), _p(scores), len(scores), C.c_float(thr), _p(keep, C.c_int))
        return keep[:n].copy()

    def nms_min(self, boxes, scores, thr):
        boxes = np.ascontiguousarray(boxes, np.float32); scores = np.ascontiguousarray(scores, np.float32)
        keep = np.empty((max(1, len(scores)),), np.int32)
        n = self.lib.orc_nms_min(_p(boxes), _p(scores), len(scores), C.c_float(thr), _p(keep, C.c_int))
        return keep[:n].copy()

    def resize_linear_u8(self, img, y0, y1, x0, x1, oh=80, ow=80):
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape[:2]
        out = np.empty((oh, ow, 3), np.uint8)
        self.lib.orc_resize_linear_u8(_p(img, C.c_uint8), H, W, int(y0), int(y1), int(x0), int(x1), int(oh), int(ow), _p(out, C.c_uint8))
        return out

    def facenet(self, x):
        x = np.ascontiguousarray(x, np.float32); n, H, W, _ = x.shape
        emb = np.empty((n, 512), np.float32)
        self.lib.orc_facenet(C.c_void_p(self.ctx), _p(x), n, H, W, _p(emb))
        return emb

    def selftest_recip_div(self, kmax):
        """Mismatches between the device's reciprocal division and IEEE division over all bins <= kmax x kmax."""
        self.lib.orc_selftest_recip_div.restype = C.c_long
        self.lib.orc_selftest_recip_div.argtypes = [C.c_int]
        return int(self.lib.orc_selftest_recip_div(int(kmax)))

    def nv12_to_bgr(self, nv12, H, W):
        nv12 = np.ascontiguousarray(nv12, np.uint8)
        out = np.empty((H, W, 3), np.uint8)
        self.lib.orc_nv12_to_bgr(_p(nv12, C.c_uint8), int(H), int(W), _p(out, C.c_uint8))
        return out

    # ---- cascade ----
    def detect(self, frame, max_out=64, trace=False, max_trace=8192):
        frame = np.ascontiguousarray(frame, np.uint8); H, W = frame.shape[:2]
        boxes = np.zeros((max_out, 4), np.float32); probs = np.zeros((max_out,), np.float32)
        tr = None; bufs = None
        if trace:
            mb = int(max_trace)
            bufs = [np.zeros((mb, 5), np.float32) for _ in range(3)] + [np.zeros((mb, 10), np.float32)]
            tr = Trace(); tr.max_boxes = mb
            tr.boxes1, tr.boxes2, tr.boxes3, tr.points3 = (_p(b) for b in bufs)
        n = self.lib.orc_detect(C.c_void_p(self.ctx), _p(frame, C.c_uint8), H, W, C.byref(self.params),
                                _p(boxes), _p(probs), max_out, C.byref(tr) if tr else None)
        k = min(n, max_out)
        res = (boxes[:k].copy(), probs[:k].copy()) if n > 0 else (None, None)
        if trace:
            if max(tr.n1, tr.n2, tr.n3) > mb:
                raise ValueError(f"trace buffers hold {mb} boxes per stage, the frame produced {tr.n1}/{tr.n2}/{tr.n3}: pass max_trace")
            t = {"n_scales": tr.n_scales, "n_cand_scale": list(tr.n_cand_scale[:tr.n_scales]),
                 "n_keep_scale": list(tr.n_keep_scale[:tr.n_scales]),
                 "boxes1": bufs[0][:tr.n1].copy(), "boxes2": bufs[1][:tr.n2].copy(),
                 "boxes3": bufs[2][:tr.n3].copy(), "points3": bufs[3][:tr.n3].copy()}
            return res + (t,)
        return res

    def detect_embed(self, frames, want_faces=False):
        frames = np.ascontiguousarray(frames, np.uint8); n, H, W, _ = frames.shape
        box = np.zeros((n, 4), np.float32); prob = np.zeros((n,), np.float32)
        rect = np.zeros((n, 4), np.int32); valid = np.zeros((n,), np.uint8)
        emb = np.zeros((n, 512), np.float32)
        faces = np.zeros((n, 80, 80, 3), np.uint8) if want_faces else None
        self.lib.orc_detect_embed(C.c_void_p(self.ctx), _p(frames, C.c_uint8), n, H, W, C.byref(self.params),
                                  _p(box), _p(prob), _p(rect, C.c_int32), _p(valid, C.c_uint8), _p(emb),
                                  _p(faces, C.c_uint8) if want_faces else None)
        out = {"box": box, "prob": prob, "rect": rect, "valid": valid, "emb": emb}
        if want_faces:
            out["faces"] = faces
        return out

    def detect_embed_mode(self, frames, mode):
        frames = np.ascontiguousarray(frames, np.uint8); n, H, W, _ = frames.shape
        box = np.zeros((n, 4), np.float32); prob = np.zeros((n,), np.float32)
        rect = np.zeros((n, 4), np.int32); valid = np.zeros((n,), np.uint8); emb = np.zeros((n, 512), np.float32)
        self.lib.orc_detect_embed_mode(C.c_void_p(self.ctx), _p(frames, C.c_uint8), n, H, W, C.byref(self.params), int(mode),
                                       _p(box), _p(prob), _p(rect, C.c_int32), _p(valid, C.c_uint8), _p(emb))
        return {"box": box, "prob": prob, "rect": rect, "valid": valid, "emb": emb}

    def crop_area_std(self, img, rect, S=160, rgb=False):
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape[:2]
        out = np.empty((S, S, 3), np.float32)
        x0, y0, x1, y1 = (int(v) for v in rect)
        self.lib.orc_crop_area_std(_p(img, C.c_uint8), H, W, x0, y0, x1, y1, int(S), int(bool(rgb)), _p(out))
        return out

    def crop_aligned(self, img, pts, S=160, rgb=True):
        """Five-point similarity-aligned crop (embedding mode 3); pts = x0..x4, y0..y4 in frame coordinates."""
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape[:2]
        pts = np.ascontiguousarray(pts, np.float32).reshape(10)
        out = np.empty((S, S, 3), np.float32)
        self.lib.orc_crop_aligned(_p(img, C.c_uint8), H, W, _p(pts), int(S), int(bool(rgb)), _p(out))
        return out

    def align_params(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(10)
        prm = np.empty((6,), np.float64)
        self.lib.orc_align_params(_p(pts), prm.ctypes.data_as(C.POINTER(C.c_double)))
        return prm

    def drift_score(self, emb, valid, frame_count, fps):
        emb = np.ascontiguousarray(emb, np.float32); valid = np.ascontiguousarray(valid, np.uint8)
        n = len(valid)
        sims = np.zeros((n,), np.float32); flags = np.zeros((n,), np.uint8)
        run = C.c_int(); hits = C.c_int()
        self.lib.orc_drift_score.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_int, C.c_long, C.c_int,
                                             C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int),
                                             C.POINTER(C.c_int)]
        score = self.lib.orc_drift_score(_p(emb), _p(valid, C.c_uint8), n, int(frame_count), int(fps),
                                         _p(sims), _p(flags, C.c_uint8), C.byref(run), C.byref(hits))
        return {"score": int(score), "sims": sims, "flags": flags, "run": run.value, "hits": hits.value}
