"""Drop-in for ``facenet_pytorch.MTCNN`` as the reference uses it:

    mtcnn = MTCNN()                      # server/model.py:18
    boxes, _ = mtcnn.detect(frame)       # server/model.py:47

``detect`` keeps the library's return convention: for one ``(H, W, 3)`` uint8 frame it returns
``(boxes, probs)`` with ``boxes`` a float32 ``(k, 4)`` array sorted largest-area first
(``select_largest=True``) or ``(None, [None])`` when no face is found (``landmarks=True`` adds the ``(k, 5, 2)``
O-Net landmarks as a third element, ``None`` without a face); for a batch
``(n, H, W, 3)`` it returns object arrays of those.  The P/R/O-Net cascade runs in
libtruely_hip.so (see csrc/trl_cascade.hip, csrc/trl_pnet.hip)."""
from __future__ import annotations

import numpy as np
import torch

from .engine import Engine, default_engine


class MTCNN:
    def __init__(self, image_size=160, margin=0, min_face_size=20, thresholds=(0.6, 0.7, 0.7), factor=0.709,
                 post_process=True, select_largest=True, selection_method=None, keep_all=False, device=None,
                 engine: Engine | None = None):
        if not select_largest or selection_method not in (None, "largest"):
            raise NotImplementedError("only select_largest=True (the MTCNN() default used by server/model.py:18) is built")
        self.image_size, self.margin, self.post_process, self.keep_all = image_size, margin, post_process, keep_all
        self.min_face_size, self.thresholds, self.factor = min_face_size, list(thresholds), factor
        defaults = (min_face_size == 20 and tuple(thresholds) == (0.6, 0.7, 0.7) and factor == 0.709)
        if engine is not None:
            self.engine = engine
        elif defaults:
            self.engine = default_engine()
        else:   # non-default cascade parameters need their own context (same weights)
            base = default_engine()
            self.engine = Engine(getattr(base, "_blob", None), device=base.cfg.device, min_face_size=min_face_size,
                                 thresholds=tuple(thresholds), factor=factor)

    def eval(self):
        return self

    def to(self, device):
        return self

    def detect(self, img, landmarks: bool = False):
        if isinstance(img, torch.Tensor):
            arr = img
            single = arr.dim() == 3
            if single:
                arr = arr[None]
        elif isinstance(img, (list, tuple)):
            arr = np.stack([np.asarray(i) for i in img]); single = False
        else:
            arr = np.asarray(img)
            single = arr.ndim == 3
            if single:
                arr = arr[None]
        res = self.engine.mtcnn_detect(arr, landmarks=landmarks)
        boxes, probs, counts = res[0].cpu().numpy(), res[1].cpu().numpy(), res[2].cpu().numpy()
        # facenet-pytorch returns points as (k, 5, 2) = (x_j, y_j); the device rows are x0..x4, y0..y4
        points = res[3].cpu().numpy().reshape(len(counts), -1, 2, 5).transpose(0, 1, 3, 2) if landmarks else None
        out_b, out_p, out_l = [], [], []
        for i in range(len(counts)):
            k = int(counts[i])
            if k == 0:
                out_b.append(None); out_p.append([None]); out_l.append(None)
            else:
                out_b.append(boxes[i, :k].copy()); out_p.append(probs[i, :k].copy())
                if landmarks:
                    out_l.append(points[i, :k].copy())
        if single:
            return (out_b[0], out_p[0], out_l[0]) if landmarks else (out_b[0], out_p[0])
        if landmarks:
            return np.array(out_b, dtype=object), np.array(out_p, dtype=object), np.array(out_l, dtype=object)
        return np.array(out_b, dtype=object), np.array(out_p, dtype=object)
