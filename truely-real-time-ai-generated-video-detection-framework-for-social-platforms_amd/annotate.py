"""Frame annotation of the reference's output video (server/model.py:67-74) without OpenCV:

    cv2.rectangle(frame, (x1, y1), (x2, y2), color, 2)
    cv2.putText(frame, "AI Detected - Frame N", (10, 30), cv2.FONT_HERSHEY_SIMPLEX, 1, (0, 0, 255), 2, cv2.LINE_AA)
    cv2.putText(frame, "Real Frame", (x1, y1 - 10), cv2.FONT_HERSHEY_SIMPLEX, 0.5, (0, 255, 0), 2, cv2.LINE_AA)

When ``cv2`` is importable it is used and the output is OpenCV's, byte for byte.  Otherwise this module draws:

* ``rectangle``: four axis-aligned thick lines.  OpenCV's ThickLine fills the polygon offset by thickness/2 on either side
  of the ideal line INCLUDING its boundary pixels, so thickness t covers offsets -(t//2) .. +(t//2): 1 px for t = 1, 3 px for
  t = 2 or 3 (RECALLED from modules/imgproc/src/drawing.cpp; there is no OpenCV here to pin it).
* ``put_text``: a single-stroke vector font in the Hershey "simplex" proportions (cap height 21 units, baseline at the text
  origin, scaled by fontScale like OpenCV) for the characters the reference prints (digits, the letters of "AI Detected -
  Frame" / "Real Frame"), rasterised as anti-aliased thick strokes by exact distance-to-segment coverage.  The glyph
  outlines are this project's own stroke tables in the Hershey style, NOT OpenCV's data: text is legible and lands where
  OpenCV puts it, but is not pixel-identical (the scores, boxes and every decision are unaffected -- annotation is cosmetic).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - not available offline
    import cv2  # type: ignore
except Exception:  # noqa: BLE001
    cv2 = None

CAP = 21.0          # Hershey simplex cap height in font units; OpenCV renders 1 unit = fontScale pixels

# strokes: list of polylines, coordinates in font units, x to the right from the glyph's left edge, y UP from the baseline
_G = {
    " ": (16, []),
    "-": (26, [[(4, 9), (22, 9)]]),
    "A": (18, [[(1, 0), (9, 21), (17, 0)], [(4, 7), (14, 7)]]),
    "I": (8, [[(4, 0), (4, 21)]]),
    "D": (21, [[(4, 0), (4, 21), (11, 21), (14, 20), (16, 18), (17, 16), (18, 13), (18, 8), (17, 5), (16, 3), (14, 1), (11, 0), (4, 0)]]),
    "F": (18, [[(4, 0), (4, 21), (17, 21)], [(4, 11), (12, 11)]]),
    "R": (21, [[(4, 0), (4, 21), (13, 21), (16, 20), (17, 19), (18, 17), (18, 15), (17, 13), (16, 12), (13, 11), (4, 11)], [(11, 11), (18, 0)]]),
    "a": (19, [[(15, 14), (15, 0)], [(15, 11), (13, 13), (11, 14), (8, 14), (6, 13), (4, 11), (3, 8), (3, 6), (4, 3), (6, 1), (8, 0), (11, 0), (13, 1), (15, 3)]]),
    "c": (18, [[(15, 11), (13, 13), (11, 14), (8, 14), (6, 13), (4, 11), (3, 8), (3, 6), (4, 3), (6, 1), (8, 0), (11, 0), (13, 1), (15, 3)]]),
    "d": (19, [[(15, 21), (15, 0)], [(15, 11), (13, 13), (11, 14), (8, 14), (6, 13), (4, 11), (3, 8), (3, 6), (4, 3), (6, 1), (8, 0), (11, 0), (13, 1), (15, 3)]]),
    "e": (18, [[(3, 8), (15, 8), (15, 10), (14, 12), (13, 13), (11, 14), (8, 14), (6, 13), (4, 11), (3, 8), (3, 6), (4, 3), (6, 1), (8, 0), (11, 0), (13, 1), (15, 3)]]),
    "l": (8, [[(4, 21), (4, 0)]]),
    "m": (30, [[(4, 14), (4, 0)], [(4, 10), (7, 13), (9, 14), (12, 14), (14, 13), (15, 10), (15, 0)], [(15, 10), (18, 13), (20, 14), (23, 14), (25, 13), (26, 10), (26, 0)]]),
    "r": (13, [[(4, 14), (4, 0)], [(4, 8), (5, 11), (7, 13), (9, 14), (12, 14)]]),
    "t": (12, [[(5, 21), (5, 4), (6, 1), (8, 0), (10, 0)], [(2, 14), (9, 14)]]),
    "0": (20, [[(9, 21), (6, 20), (4, 17), (3, 12), (3, 9), (4, 4), (6, 1), (9, 0), (11, 0), (14, 1), (16, 4), (17, 9), (17, 12), (16, 17), (14, 20), (11, 21), (9, 21)]]),
    "1": (20, [[(6, 17), (8, 18), (11, 21), (11, 0)]]),
    "2": (20, [[(4, 16), (4, 17), (5, 19), (6, 20), (8, 21), (12, 21), (14, 20), (15, 19), (16, 17), (16, 15), (15, 13), (13, 10), (3, 0), (17, 0)]]),
    "3": (20, [[(5, 21), (16, 21), (10, 13), (13, 13), (15, 12), (16, 11), (17, 8), (17, 6), (16, 3), (14, 1), (11, 0), (8, 0), (5, 1), (4, 2), (3, 4)]]),
    "4": (20, [[(13, 21), (3, 7), (18, 7)], [(13, 21), (13, 0)]]),
    "5": (20, [[(15, 21), (5, 21), (4, 12), (5, 13), (8, 14), (11, 14), (14, 13), (16, 11), (17, 8), (17, 6), (16, 3), (14, 1), (11, 0), (8, 0), (5, 1), (4, 2), (3, 4)]]),
    "6": (20, [[(16, 18), (15, 20), (12, 21), (10, 21), (7, 20), (5, 17), (4, 12), (4, 7), (5, 3), (7, 1), (10, 0), (11, 0), (14, 1), (16, 3), (17, 6), (17, 7), (16, 10), (14, 12), (11, 13), (10, 13), (7, 12), (5, 10), (4, 7)]]),
    "7": (20, [[(17, 21), (7, 0)], [(3, 21), (17, 21)]]),
    "8": (20, [[(8, 21), (5, 20), (4, 18), (4, 16), (5, 14), (7, 13), (11, 12), (14, 11), (16, 9), (17, 7), (17, 4), (16, 2), (15, 1), (12, 0), (8, 0), (5, 1), (4, 2), (3, 4), (3, 7), (4, 9), (6, 11), (9, 12), (13, 13), (15, 14), (16, 16), (16, 18), (15, 20), (12, 21), (8, 21)]]),
    "9": (20, [[(16, 14), (15, 11), (13, 9), (10, 8), (9, 8), (6, 9), (4, 11), (3, 14), (3, 15), (4, 18), (6, 20), (9, 21), (10, 21), (13, 20), (15, 18), (16, 14), (16, 9), (15, 4), (13, 1), (10, 0), (8, 0), (5, 1), (4, 3)]]),
}


def text_size(text: str, scale: float):
    """(width, height) in pixels like cv2.getTextSize: sum of glyph advances, cap height."""
    return sum(_G.get(ch, _G[" "])[0] for ch in text) * scale, CAP * scale


def _blend_segment(img, x0, y0, x1, y1, half, color):
    """Anti-aliased thick segment: coverage = clamp(half + 0.5 - distance, 0, 1) at pixel centres."""
    H, W = img.shape[:2]
    xa, xb = int(np.floor(min(x0, x1) - half - 1)), int(np.ceil(max(x0, x1) + half + 1))
    ya, yb = int(np.floor(min(y0, y1) - half - 1)), int(np.ceil(max(y0, y1) + half + 1))
    xa, ya, xb, yb = max(xa, 0), max(ya, 0), min(xb, W - 1), min(yb, H - 1)
    if xa > xb or ya > yb:
        return
    yy, xx = np.mgrid[ya:yb + 1, xa:xb + 1].astype(np.float32)
    dx, dy = x1 - x0, y1 - y0
    L2 = dx * dx + dy * dy
    t = np.clip(((xx - x0) * dx + (yy - y0) * dy) / L2, 0.0, 1.0) if L2 > 0 else np.zeros_like(xx)
    d = np.hypot(xx - (x0 + t * dx), yy - (y0 + t * dy))
    a = np.clip(half + 0.5 - d, 0.0, 1.0)[..., None]
    reg = img[ya:yb + 1, xa:xb + 1].astype(np.float32)
    # strokes of one glyph overlap at the joints: combine by max coverage, i.e. never blend an already inked pixel back
    out = reg * (1 - a) + np.asarray(color, np.float32) * a
    img[ya:yb + 1, xa:xb + 1] = np.clip(np.rint(out), 0, 255).astype(np.uint8)


def put_text(frame: np.ndarray, text: str, org, scale: float, color, thickness: int = 1) -> None:
    """cv2.putText(frame, text, org, FONT_HERSHEY_SIMPLEX, scale, color, thickness, LINE_AA): org = bottom-left of the text."""
    if cv2 is not None:  # pragma: no cover
        cv2.putText(frame, text, org, cv2.FONT_HERSHEY_SIMPLEX, scale, color, thickness, cv2.LINE_AA)
        return
    x, y = float(org[0]), float(org[1])
    half = max(thickness, 1) / 2.0
    for ch in text:
        adv, strokes = _G.get(ch, _G[" "])
        for line in strokes:
            for (ax, ay), (bx, by) in zip(line[:-1], line[1:]):
                _blend_segment(frame, x + ax * scale, y - ay * scale, x + bx * scale, y - by * scale, half, color)
        x += adv * scale


def rectangle(frame: np.ndarray, pt1, pt2, color, thickness: int = 1) -> None:
    """cv2.rectangle(frame, pt1, pt2, color, thickness) for thickness >= 1 (outline), clipped to the frame."""
    if cv2 is not None:  # pragma: no cover
        cv2.rectangle(frame, pt1, pt2, color, thickness)
        return
    H, W = frame.shape[:2]
    (x0, y0), (x1, y1) = pt1, pt2
    x0, x1 = min(x0, x1), max(x0, x1)
    y0, y1 = min(y0, y1), max(y0, y1)
    h = max(thickness, 1) // 2

    def fill(ya, yb, xa, xb):      # inclusive pixel ranges
        ya, xa, yb, xb = max(ya, 0), max(xa, 0), min(yb, H - 1), min(xb, W - 1)
        if ya <= yb and xa <= xb:
            frame[ya:yb + 1, xa:xb + 1] = color

    fill(y0 - h, y0 + h, x0 - h, x1 + h)
    fill(y1 - h, y1 + h, x0 - h, x1 + h)
    fill(y0 - h, y1 + h, x0 - h, x0 + h)
    fill(y0 - h, y1 + h, x1 - h, x1 + h)


def annotate(frame: np.ndarray, index: int, rect, flagged: bool) -> None:
    """server/model.py:67-74 for one sampled frame that was compared with its predecessor."""
    x0, y0, x1, y1 = (int(v) for v in rect)
    if flagged:
        rectangle(frame, (x0, y0), (x1, y1), (0, 0, 255), 2)
        put_text(frame, f"AI Detected - Frame {index}", (10, 30), 1, (0, 0, 255), 2)
    else:
        rectangle(frame, (x0, y0), (x1, y1), (0, 255, 0), 2)
        put_text(frame, "Real Frame", (x0, y0 - 10), 0.5, (0, 255, 0), 2)
