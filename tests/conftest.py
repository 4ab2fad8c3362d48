import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import truely_amd  # noqa: E402  (import shim for the hyphenated package directory)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, loudly
    from truely_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libtrl_oracle.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def state_dicts():
    return truely_amd.weights.synthetic_state_dicts(0)


@pytest.fixture(scope="session")
def blob(state_dicts):
    return truely_amd.weights.pack_state_dicts(*state_dicts)


@pytest.fixture(scope="session")
def oracle(blob):
    from oracle.oracle import Oracle
    return Oracle(blob)


@pytest.fixture(scope="session")
def engine(blob):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from truely_amd.engine import Engine
    return Engine(blob)


@pytest.fixture(scope="session")
def engine_generic(blob):
    """Same weights, PNet through the generic layer kernels (validation path)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from truely_amd.engine import Engine
    return Engine(blob, pnet_mode=1)


def frames_small(n=6, H=180, W=320, seed=3):
    return truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed)


def assert_greedy_nms_fixed_point(rec, keep_idx, oh, ow, thr, R=7):
    """Greedy NMS in priority order (score descending, ties by cell index ascending) yields the UNIQUE set K with
        x in K  <=>  no y in K of higher priority has IoU(x, y) > thr          (induction over the priority order),
    so checking that equivalence for every candidate proves a keep set equal to the sequential algorithm's -- in O(n * window)
    for the candidates of ONE pyramid level, which sit on the PNet cell grid and can only overlap within a few cells.  float32
    arithmetic in torchvision's operation order (areas without +1, inter / (a + b - inter) > thr)."""
    n = len(rec)
    cell = rec["cell"].astype(np.int64)
    assert len(np.unique(cell)) == n and cell.min() >= 0 and cell.max() < oh * ow
    order = keep_idx.astype(np.int64)
    assert len(np.unique(order)) == len(order) and (order >= 0).all() and (order < n).all()
    ks, kc = rec["score"][order], cell[order]                                      # pick order = priority order
    assert ((ks[:-1] > ks[1:]) | ((ks[:-1] == ks[1:]) & (kc[:-1] < kc[1:]))).all()
    P = R
    S = np.full((oh + 2 * P, ow + 2 * P), -np.inf, np.float32)
    Hs = np.zeros((oh + 2 * P, ow + 2 * P), bool)
    K = np.zeros_like(Hs)
    B = np.zeros((4, oh + 2 * P, ow + 2 * P), np.float32)
    yy, xx = cell // ow + P, cell % ow + P
    S[yy, xx] = rec["score"]; Hs[yy, xx] = True
    K[yy[order], xx[order]] = True
    for q in range(4):
        B[q][yy, xx] = rec["box"][:, q]
    c = (slice(P, P + oh), slice(P, P + ow))
    x1, y1, x2, y2 = (B[q][c] for q in range(4))
    area = (x2 - x1) * (y2 - y1)
    sup = np.zeros((oh, ow), bool)
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            if dy == 0 and dx == 0:
                continue
            nb = (slice(P + dy, P + dy + oh), slice(P + dx, P + dx + ow))
            nx1, ny1, nx2, ny2 = (B[q][nb] for q in range(4))
            w = np.maximum(np.minimum(x2, nx2) - np.maximum(x1, nx1), np.float32(0))
            h = np.maximum(np.minimum(y2, ny2) - np.maximum(y1, ny1), np.float32(0))
            inter = w * h
            both = Hs[c] & Hs[nb]
            if max(abs(dy), abs(dx)) == R:
                assert not (both & (inter > 0)).any(), "window too small for this level"
                continue
            with np.errstate(invalid="ignore", divide="ignore"):
                ovr = inter / (area + (nx2 - nx1) * (ny2 - ny1) - inter)
            beats = (S[nb] > S[c]) | ((S[nb] == S[c]) & ((dy < 0) or (dy == 0 and dx < 0)))
            sup |= both & K[nb] & beats & (ovr > np.float32(thr))
    ok = Hs[c] & (K[c] == ~sup)
    assert ok.sum() == n, f"{n - ok.sum()} of {n} keep decisions differ from greedy NMS"
