"""Dev-time calibration of the SYNTHETIC weights (test infrastructure; uses the oracle).

Random weights make a useless cascade: PNet either fires nowhere or everywhere and the random
FaceNet maps every crop to nearly the same direction (cosine > 0.99 between unrelated crops).
This script measures, on seeded synthetic frames, the face-logit distributions of the three
MTCNN heads and the pre-BN embedding statistics, and writes
``<package>/synthetic_calibration.npz`` holding
  * one face-logit bias offset per net (so ~0.2 % of PNet cells, ~12 % of R-Net and ~40 % of
    O-Net candidates pass -- the candidate counts of a trained cascade), and
  * last_bn running_mean / running_var that whiten the 512-d embedding (what training does).
The product only reads that .npz as data; it never imports this file.

Run:  python -m oracle.calibrate_synthetic
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "truely-real-time-ai-generated-video-detection-framework-for-social-platforms_amd")


def _load(name):
    spec = importlib.util.spec_from_file_location("trlcal_" + name, os.path.join(PKG, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["trlcal_" + name] = m
    spec.loader.exec_module(m)
    return m


def logit(p):
    p = np.clip(p.astype(np.float64), 1e-9, 1 - 1e-9)
    return np.log(p / (1 - p))


def pad(boxes, W, H):
    b = np.trunc(boxes[:, :4]).astype(np.int32)
    x = np.maximum(b[:, 0], 1); y = np.maximum(b[:, 1], 1)
    ex = np.minimum(b[:, 2], W); ey = np.minimum(b[:, 3], H)
    return y, ey, x, ex


def main(seed=0, H=720, W=1280, nframes=6):
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    from oracle.torch_ref import TorchRef
    weights = _load("weights"); synthetic = _load("synthetic")
    frames = np.concatenate([synthetic.synthetic_frames(nframes // 2, H, W, seed=s) for s in (0, 1)])
    cal = {"pnet": 0.0, "rnet": 0.0, "onet": 0.0}
    T0 = np.log(0.6 / 0.4); T1 = np.log(0.7 / 0.3)

    # -- PNet: 0.2 % of all pyramid cells pass thr0
    o = Oracle(weights.pack_state_dicts(*weights.synthetic_state_dicts(seed, cal, use_file=False)))
    ds = []
    for f in frames:
        for (_s, h, w) in o.scales(H, W):
            p, _ = o.pnet_level(o.area_resample_norm(f, 0, H, 0, W, h, w))
            ds.append(logit(p.ravel()))
    ds = np.concatenate(ds)
    cal["pnet"] = float(T0 - np.quantile(ds, 1 - 0.002))
    print("pnet offset", cal["pnet"], "cells", len(ds) // len(frames))

    # -- RNet: 12 % of stage-1 candidates pass thr1
    o = Oracle(weights.pack_state_dicts(*weights.synthetic_state_dicts(seed, cal, use_file=False)))
    ds = []
    for f in frames:
        _b, _p, tr = o.detect(f, trace=True)
        b1 = tr["boxes1"]
        print("  stage1:", tr["n_cand_scale"], "->", len(b1))
        y, ey, x, ex = pad(b1, W, H)
        crops = np.stack([o.area_resample_norm(f, y[k] - 1, ey[k], x[k] - 1, ex[k], 24, 24) for k in range(len(b1))])
        ds.append(logit(o.rnet(crops)[0]))
    ds = np.concatenate(ds)
    cal["rnet"] = float(T1 - np.quantile(ds, 1 - 0.12))
    print("rnet offset", cal["rnet"])

    # -- ONet: 40 % of stage-2 candidates pass thr2
    o = Oracle(weights.pack_state_dicts(*weights.synthetic_state_dicts(seed, cal, use_file=False)))
    ds = []
    for f in frames:
        _b, _p, tr = o.detect(f, trace=True)
        b2 = tr["boxes2"]
        print("  stage2:", len(tr["boxes1"]), "->", len(b2))
        if len(b2) == 0:
            continue
        y, ey, x, ex = pad(b2, W, H)
        crops = np.stack([o.area_resample_norm(f, y[k] - 1, ey[k], x[k] - 1, ex[k], 48, 48) for k in range(len(b2))])
        ds.append(logit(o.onet(crops)[0]))
    ds = np.concatenate(ds)
    cal["onet"] = float(T1 - np.quantile(ds, 1 - 0.40))
    print("onet offset", cal["onet"])

    # -- FaceNet last_bn: whiten the pre-BN 512-d features over detected face crops
    sds = weights.synthetic_state_dicts(seed, cal, use_file=False)
    o = Oracle(weights.pack_state_dicts(*sds))
    tr = TorchRef(*sds)
    faces = []
    for s in range(4):
        fr = synthetic.synthetic_frames(8, 360, 640, seed=100 + s)
        r = o.detect_embed(fr, want_faces=True)
        faces.append(r["faces"][r["valid"] == 1])
        print("  faces seed", 100 + s, int(r["valid"].sum()), "of", len(fr))
    faces = np.concatenate(faces)
    import torch
    feats = []
    hook = tr.facenet.last_linear.register_forward_hook(lambda m, i, out: feats.append(out.detach().numpy().copy()))
    with torch.no_grad():
        for f in faces:
            tr.embed(f)
    hook.remove()
    feats = np.concatenate(feats).astype(np.float64)
    mean = feats.mean(0).astype(np.float32)
    var = np.maximum(feats.var(0), 1e-6).astype(np.float32)
    out = os.path.join(PKG, "synthetic_calibration.npz")
    np.savez(out, seed=np.int64(seed), pnet=np.float32(cal["pnet"]), rnet=np.float32(cal["rnet"]),
             onet=np.float32(cal["onet"]), last_bn_mean=mean, last_bn_var=var)
    print("wrote", out, cal, "faces used", len(faces))


if __name__ == "__main__":
    main()
