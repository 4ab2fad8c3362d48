"""Seeded synthetic clips (there is no dataset, decoder or network in the build environment).

BASELINE.json config[1]: "Synthetic 720p 1-face frames".  A frame is uint8 BGR (H, W, 3) like
``cv2.VideoCapture.read()`` returns (server/model.py:43): a smooth low-amplitude background plus
``faces`` face-like blobs (ellipse, two dark eyes, a mouth) whose position drifts slowly from
frame to frame so consecutive embeddings are close, as in a real talking-head clip.
Only numpy's PCG64 generator is used, so the bytes are identical on every machine.
"""
from __future__ import annotations

import numpy as np


def _upsample_bilinear(small: np.ndarray, H: int, W: int) -> np.ndarray:
    hs, ws = small.shape[:2]
    ys = np.linspace(0, hs - 1, H, dtype=np.float32)
    xs = np.linspace(0, ws - 1, W, dtype=np.float32)
    y0 = np.minimum(ys.astype(np.int32), hs - 2); fy = (ys - y0)[:, None, None]
    x0 = np.minimum(xs.astype(np.int32), ws - 2); fx = (xs - x0)[None, :, None]
    rows = small[y0] * (1 - fy) + small[y0 + 1] * fy            # (H, ws, 3)
    return rows[:, x0] * (1 - fx) + rows[:, x0 + 1] * fx        # (H, W, 3)


def _draw_face(img: np.ndarray, cx: float, cy: float, r: float, tone: np.ndarray) -> None:
    H, W = img.shape[:2]
    x0, x1 = max(0, int(cx - 1.2 * r)), min(W, int(cx + 1.2 * r) + 1)
    y0, y1 = max(0, int(cy - 1.5 * r)), min(H, int(cy + 1.5 * r) + 1)
    if x1 <= x0 or y1 <= y0:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
    u, v = (xx - cx) / r, (yy - cy) / (1.3 * r)
    d = u * u + v * v
    m = np.clip((1.0 - d) * 6.0, 0.0, 1.0)[..., None]
    shade = (1.0 - 0.25 * d)[..., None]
    face = tone[None, None, :] * shade
    for ex in (-0.38, 0.38):                                     # eyes
        e = np.exp(-(((u - ex) / 0.16) ** 2 + ((v + 0.25) / 0.10) ** 2))[..., None]
        face = face * (1 - 0.8 * e)
    mo = np.exp(-((u / 0.35) ** 2 + ((v - 0.45) / 0.07) ** 2))[..., None]   # mouth
    face = face * (1 - 0.6 * mo)
    no = np.exp(-((u / 0.08) ** 2 + ((v - 0.08) / 0.22) ** 2))[..., None]   # nose ridge
    face = face * (1 + 0.12 * no)
    img[y0:y1, x0:x1] = img[y0:y1, x0:x1] * (1 - m) + face * m


def synthetic_frames(n: int, H: int, W: int, seed: int = 0, faces: int = 1) -> np.ndarray:
    """uint8 (n, H, W, 3).  ``faces`` blobs per frame (``faces<0``: seeded 3..5 per frame)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, H, W, 3), dtype=np.uint8)
    hs, ws = H // 24 + 2, W // 24 + 2
    bg0 = rng.uniform(70, 170, (hs, ws, 3)).astype(np.float32)
    bg1 = rng.uniform(70, 170, (hs, ws, 3)).astype(np.float32)
    tile = rng.integers(-4, 5, (H, W, 3)).astype(np.float32)    # fine texture, rolled per frame
    kmax = 5 if faces < 0 else faces
    base = np.stack([rng.uniform(0.2, 0.8, kmax) * W, rng.uniform(0.25, 0.75, kmax) * H], 1)
    rad = rng.uniform(0.09, 0.2, kmax) * min(H, W)
    tones = rng.uniform(110, 215, (kmax, 3)).astype(np.float32)
    vel = rng.normal(0, 0.6, (kmax, 2))
    for i in range(n):
        t = i / max(1, n - 1)
        img = _upsample_bilinear(bg0 * (1 - t) + bg1 * t, H, W)
        img += np.roll(tile, (int(rng.integers(0, H)), int(rng.integers(0, W))), (0, 1))
        k = kmax if faces >= 0 else int(rng.integers(3, 6))
        for j in range(k):
            jit = rng.normal(0, 0.8, 2)
            cx, cy = base[j] + vel[j] * i + jit
            _draw_face(img, float(cx), float(cy), float(rad[j]), tones[j])
        out[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out
