"""Decision-level comparison of the two CPU restatements + near-threshold audit (test infrastructure).

The cascade is data dependent: a 1-ulp move of a PNet probability near 0.6, an R-/O-Net probability near 0.7 or an IoU
near 0.5 / 0.7 flips a candidate ("hard parts").  The C oracle (oracle/trl_oracle.c) accumulates
every conv as one k-ascending fmaf chain; torch's CPU kernels (oracle/torch_ref.py, the ops facenet-pytorch itself is
built on) use their own blocking and summation order.  `audit_frame` runs both on one frame and

* compares every DECISION by identity, not by tolerance: per pyramid level the set of PNet cells with prob >= thr0 and
  the per-level NMS picks (cell indices), the cross-level NMS picks, the candidates passing R-Net / O-Net thresholds
  and the stage-2 / stage-3 NMS picks (indices into the previous stage's list);
* re-derives the C oracle's decisions from its own primitives (so the cascade code is also checked against them);
* records how close any compared quantity came to its threshold under torch's accumulation order.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _pad(b, w, h):
    t = np.trunc(b[:, :4]).astype(np.int32)
    x, y, ex, ey = t[:, 0].copy(), t[:, 1].copy(), t[:, 2].copy(), t[:, 3].copy()
    x[x < 1] = 1; y[y < 1] = 1; ex[ex > w] = w; ey[ey > h] = h
    return y, ey, x, ex


def _crops(oracle, frame, boxes, size):
    H, W = frame.shape[:2]
    y, ey, x, ex = _pad(boxes, W, H)
    out = []
    for k in range(len(y)):
        assert ey[k] > y[k] - 1 and ex[k] > x[k] - 1
        out.append(oracle.area_resample_norm(frame, y[k] - 1, ey[k], x[k] - 1, ex[k], size, size))
    return np.stack(out) if out else np.zeros((0, size, size, 3), F32)


def c_oracle_decisions(oracle, frame, thr=(0.6, 0.7, 0.7), minsize=20):
    """The C oracle's decisions, re-derived stage by stage from its own primitives and its own stage boxes."""
    H, W = frame.shape[:2]
    _b, _p, tr = oracle.detect(frame, trace=True)
    dec = {"cand_cells": [], "keep_cells": [], "trace": tr}
    rows = []
    for l, (sc, h, w) in enumerate(oracle.scales(H, W, minsize)):
        p, r = oracle.pnet_level(oracle.area_resample_norm(frame, 0, H, 0, W, h, w))
        cells = np.flatnonzero(p.reshape(-1) >= F32(thr[0]))
        ys, xs = np.divmod(cells, p.shape[1])
        scf = F32(sc)
        q = np.stack([np.floor((F32(2) * xs.astype(F32) + F32(1)) / scf), np.floor((F32(2) * ys.astype(F32) + F32(1)) / scf),
                      np.floor((F32(2) * xs.astype(F32) + F32(12)) / scf), np.floor((F32(2) * ys.astype(F32) + F32(12)) / scf)], 1)
        score = p.reshape(-1)[cells]
        pick = oracle.nms_iou(q, score, 0.5) if len(cells) else np.zeros(0, np.int32)
        dec["cand_cells"].append(cells); dec["keep_cells"].append(cells[pick])
        assert len(cells) == tr["n_cand_scale"][l] and len(pick) == tr["n_keep_scale"][l], "cascade disagrees with its primitives"
        rows.append(np.concatenate([q[pick], score[pick, None]], 1))
    allb = np.concatenate(rows) if rows else np.zeros((0, 5), F32)
    dec["pick1"] = oracle.nms_iou(allb[:, :4], allb[:, 4], 0.7) if len(allb) else np.zeros(0, np.int32)
    b1 = tr["boxes1"]
    assert len(b1) == len(dec["pick1"]), "stage-1 list length (a candidate with an empty clipped box was dropped?)"
    if len(b1) == 0:
        return dec
    p2, r2 = oracle.rnet(_crops(oracle, frame, b1, 24))
    dec["score2"] = p2
    dec["pass2"] = np.flatnonzero(p2 > F32(thr[1]))
    if len(dec["pass2"]) == 0:
        return dec
    dec["pick2"] = dec["pass2"][oracle.nms_iou(b1[dec["pass2"], :4], p2[dec["pass2"]], 0.7)]
    b2 = tr["boxes2"]
    assert len(b2) == len(dec["pick2"])
    p3, r3, _pts = oracle.onet(_crops(oracle, frame, b2, 48))
    dec["score3"] = p3
    dec["pass3"] = np.flatnonzero(p3 > F32(thr[2]))
    if len(dec["pass3"]) == 0:
        return dec
    bb = b2[dec["pass3"], :4].copy(); mv = r3[dec["pass3"]]
    bw = bb[:, 2] - bb[:, 0] + F32(1); bh = bb[:, 3] - bb[:, 1] + F32(1)
    bb = np.stack([bb[:, 0] + mv[:, 0] * bw, bb[:, 1] + mv[:, 1] * bh, bb[:, 2] + mv[:, 2] * bw, bb[:, 3] + mv[:, 3] * bh], 1).astype(F32)
    dec["pick3"] = dec["pass3"][oracle.nms_min(bb, p3[dec["pass3"]], 0.7)]
    assert len(tr["boxes3"]) == len(dec["pick3"])
    return dec


def audit_frame(oracle, tref, frame):
    """Returns (audit dict of torch-side margins, summary) after asserting that both restatements decide identically."""
    tt = {}
    tb, tp = tref.detect(frame, tt)
    cd = c_oracle_decisions(oracle, frame, minsize=tref.minsize)
    tr = cd["trace"]
    L = len(cd["cand_cells"])
    assert len(tt.get("cand_cells", [])) == L
    for l in range(L):
        assert np.array_equal(tt["cand_cells"][l], cd["cand_cells"][l]), f"level {l}: PNet candidate set differs"
        assert np.array_equal(tt["keep_cells"][l], cd["keep_cells"][l]), f"level {l}: per-level NMS picks differ"
    n_dec = {"levels": L, "cand": int(sum(len(c) for c in cd["cand_cells"])), "keep": int(sum(len(c) for c in cd["keep_cells"]))}
    for key in ("pick1", "pass2", "pick2", "pass3", "pick3"):
        a, b = tt.get(key), cd.get(key)
        assert (a is None or len(a) == 0) == (b is None or len(b) == 0), f"{key}: one side stopped early"
        if a is not None and b is not None:
            assert np.array_equal(np.asarray(a, np.int64), np.asarray(b, np.int64)), f"{key} differs"
            n_dec[key] = int(len(a))
    for s, tol in ((1, 1e-3), (2, 1e-2), (3, 1e-2)):
        a = tt.get(f"boxes{s}")
        if a is not None:
            b = tr[f"boxes{s}"]
            assert a.shape == b.shape and np.abs(a - b).max() < tol, f"boxes{s}"
    assert (tb is None) == (_b_none(tr))
    return tt.get("audit", {}), n_dec


def _b_none(tr):
    return len(tr["boxes3"]) == 0


def merge_audits(audits):
    out = {}
    for a in audits:
        for k, v in a.items():
            o = out.setdefault(k, {"n": 0, "min_margin": float("inf"), "within_1e-5": 0, "within_1e-6": 0})
            o["n"] += v["n"]; o["min_margin"] = min(o["min_margin"], v["min_margin"])
            o["within_1e-5"] += v["within_1e-5"]; o["within_1e-6"] += v["within_1e-6"]
    return out
