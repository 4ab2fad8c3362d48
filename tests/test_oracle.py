"""CPU tests of the oracle itself: (1) against the committed golden vectors, (2) against torch CPU
ops it restates (oracle/torch_ref.py), (3) hand-derived known answers for the score state machine
(server/model.py:60-66,86-95).  No GPU."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import truely_amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "clip_*.npz"))))
def test_oracle_reproduces_golden(oracle, path):
    z = np.load(path)
    n, H, W, seed = int(z["n"]), int(z["H"]), int(z["W"]), int(z["seed"])
    if H >= 720:
        n = 1   # keep the CPU suite short
    faces = int(z["faces_per_frame"]) if "faces_per_frame" in z else 1
    fr = truely_amd.synthetic.synthetic_frames(int(z["n"]), H, W, seed=seed, faces=faces)[:n]
    if n == int(z["n"]):
        assert int(fr.astype(np.uint64).sum()) == int(z["frames_crc"]), "synthetic frame generator changed"
    r = oracle.detect_embed(fr, want_faces=True)
    for k in ("box", "prob", "rect", "valid", "emb", "faces"):
        assert np.array_equal(r[k], z[k][:n]), k
    for i in range(n):
        _b, _p, tr = oracle.detect(fr[i], trace=True)
        assert tr["n_cand_scale"] == z[f"f{i}_cand"].tolist()
        assert tr["n_keep_scale"] == z[f"f{i}_keep"].tolist()
        for s in (1, 2, 3):
            assert np.array_equal(tr[f"boxes{s}"], z[f"f{i}_boxes{s}"])
    if n == int(z["n"]):
        d = oracle.drift_score(r["emb"], r["valid"], n * 4, 30)
        assert d["score"] == int(z["score"]) and np.array_equal(d["sims"], z["sims"])


def test_primitives_golden(oracle):
    z = np.load(os.path.join(GOLD, "primitives.npz"))
    assert np.array_equal(oracle.facenet(z["facenet_in"]), z["facenet_out"])
    p, r = oracle.rnet(z["rnet_in"])
    assert np.array_equal(p, z["rnet_prob"]) and np.array_equal(r, z["rnet_reg"])
    p, r, t = oracle.onet(z["onet_in"])
    assert np.array_equal(p, z["onet_prob"]) and np.array_equal(r, z["onet_reg"]) and np.array_equal(t, z["onet_pts"])


# ---- against torch CPU (the ops the reference's libraries are built on) -------------------------------
@pytest.fixture(scope="module")
def tref(state_dicts):
    from oracle.torch_ref import TorchRef
    return TorchRef(*state_dicts)


def test_area_resample_bit_exact_vs_torch(oracle):
    fr = truely_amd.synthetic.synthetic_frames(1, 180, 320, seed=5)[0]
    im = torch.as_tensor(fr.copy()).unsqueeze(0).permute(0, 3, 1, 2).float()
    for (h, w) in [(109, 193), (24, 24), (48, 48), (7, 11)]:
        a = oracle.area_resample_norm(fr, 0, 180, 0, 320, h, w)
        b = ((F.interpolate(im, size=(h, w), mode="area") - 127.5) * 0.0078125)[0].permute(1, 2, 0).numpy()
        assert np.array_equal(a, b)
    # upsampling crop (bins of width 1)
    a = oracle.area_resample_norm(fr, 10, 21, 30, 39, 24, 24)
    b = ((F.interpolate(im[:, :, 10:21, 30:39], size=(24, 24), mode="area") - 127.5) * 0.0078125)[0].permute(1, 2, 0).numpy()
    assert np.array_equal(a, b)


def test_networks_close_to_torch(oracle, tref):
    rng = np.random.default_rng(1)
    lvl = rng.uniform(-1, 1, (57, 83, 3)).astype(np.float32)
    p, r = oracle.pnet_level(lvl)
    with torch.no_grad():
        reg, probs = tref.pnet(torch.as_tensor(lvl).permute(2, 0, 1).unsqueeze(0))
    assert np.abs(p - probs[0, 1].numpy()).max() < 2e-6 and np.abs(r - reg[0].permute(1, 2, 0).numpy()).max() < 2e-6
    c24 = rng.uniform(-1, 1, (5, 24, 24, 3)).astype(np.float32)
    p, r = oracle.rnet(c24)
    with torch.no_grad():
        r2, p2 = tref.rnet(torch.as_tensor(c24).permute(0, 3, 1, 2))
    assert np.abs(p - p2[:, 1].numpy()).max() < 2e-6 and np.abs(r - r2.numpy()).max() < 5e-6
    c48 = rng.uniform(-1, 1, (3, 48, 48, 3)).astype(np.float32)
    p, r, t = oracle.onet(c48)
    with torch.no_grad():
        r2, l2, p2 = tref.onet(torch.as_tensor(c48).permute(0, 3, 1, 2))
    assert np.abs(p - p2[:, 1].numpy()).max() < 2e-6 and np.abs(r - r2.numpy()).max() < 5e-6 and np.abs(t - l2.numpy()).max() < 5e-6
    x = rng.uniform(0, 1, (2, 80, 80, 3)).astype(np.float32)
    e = oracle.facenet(x)
    with torch.no_grad():
        e2 = tref.facenet(torch.as_tensor(x).permute(0, 3, 1, 2)).numpy()
    assert np.abs(e - e2).max() < 1e-5   # north-star tolerance is 1e-4
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-6)


def test_cascade_matches_torch_restatement(oracle, tref):
    """Whole detect(): C oracle vs the torch restatement (independent code, torch's conv order)."""
    fr = truely_amd.synthetic.synthetic_frames(2, 360, 640, seed=11)
    for f in fr:
        tr = {}
        boxes, probs = tref.detect(f, tr)
        b2, p2, t2 = oracle.detect(f, trace=True)
        assert t2["boxes1"].shape == tr["boxes1"].shape
        assert np.abs(t2["boxes1"] - tr["boxes1"]).max() < 1e-3
        assert (boxes is None) == (b2 is None)
        if boxes is not None:
            assert boxes.shape == b2.shape and np.abs(boxes - b2).max() < 1e-2 and np.abs(probs - p2).max() < 1e-5


def test_nms_against_bruteforce(oracle):
    rng = np.random.default_rng(3)
    n = 300
    xy = rng.uniform(0, 200, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 60, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    scores[10] = scores[20]   # a tie: stable order keeps index 10 first
    from oracle.torch_ref import _nms_iou, _nms_min
    k1 = oracle.nms_iou(boxes, scores, 0.5)
    k2 = _nms_iou(torch.as_tensor(boxes), torch.as_tensor(scores), 0.5).numpy()
    assert np.array_equal(k1, k2)
    k3 = oracle.nms_min(boxes, scores, 0.7)
    assert np.array_equal(k3, _nms_min(boxes, scores, 0.7))
    assert len(oracle.nms_iou(boxes[:0], scores[:0], 0.5)) == 0


def test_resize_linear_properties(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    # identity when the crop already is 80x80 (OpenCV copies)
    assert np.array_equal(oracle.resize_linear_u8(img, 10, 90, 20, 100), img[10:90, 20:100])
    # exact 2x down-scale equals the 2x2 box average with rounding (OpenCV's INTER_AREA fast path)
    crop = img[0:120:1, 0:160:1][:160 // 2 * 0 + 120, :160]
    sub = img[:160 // 2 * 0 + 160 // 2 * 0 + 120, :160]
    ref = (sub[0::2, 0::2].astype(np.int32) + sub[1::2, 0::2] + sub[0::2, 1::2] + sub[1::2, 1::2] + 2) >> 2
    got = oracle.resize_linear_u8(img, 0, 120, 0, 160, 60, 80)
    assert np.array_equal(got, ref.astype(np.uint8))
    # constant image stays constant
    c = np.full((50, 70, 3), 137, np.uint8)
    assert (oracle.resize_linear_u8(c, 0, 50, 0, 70) == 137).all()
    # float bilinear agrees within 1 LSB
    x = torch.as_tensor(img[5:100, 7:150].copy()).permute(2, 0, 1).unsqueeze(0).float()
    fl = F.interpolate(x, size=(80, 80), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(oracle.resize_linear_u8(img, 5, 100, 7, 150).astype(np.float32) - fl).max() <= 1.01


def test_expf_and_dot(oracle):
    xs = np.linspace(-30, 0, 301, dtype=np.float32)
    got = np.array([oracle.expf(float(x)) for x in xs], np.float32)
    assert np.abs(got / np.exp(xs.astype(np.float64)) - 1).max() < 3e-7
    a = np.random.default_rng(0).standard_normal(512).astype(np.float32)
    b = np.random.default_rng(1).standard_normal(512).astype(np.float32)
    assert abs(oracle.dot512(a, b) - float(a.astype(np.float64) @ b.astype(np.float64))) < 1e-4


# ---- known answers for model.py:60-66,86-95, derived by hand from the reference source ------------------
def _emb(n, same=True):
    rng = np.random.default_rng(7)
    if same:
        e = np.tile(rng.standard_normal(512).astype(np.float32), (n, 1))
    else:
        e = rng.standard_normal((n, 512)).astype(np.float32)     # unrelated vectors: cosine ~ 0 < 0.99
    return e / np.linalg.norm(e, axis=1, keepdims=True)


def test_score_all_identical_is_zero(oracle):
    d = oracle.drift_score(_emb(240), np.ones(240, np.uint8), 960, 30)
    assert d["score"] == 0 and d["run"] == 0 and d["hits"] == 0


def test_score_all_different_long_clip(oracle):
    # 960 frames @30 fps -> step 4 -> 240 sampled, 239 comparisons, run>15 first at the 16th:
    # hits = 239-15 = 224, pct = 93.33, run_final = 239, conf = 100, N=960 > 900 -> int(min(93.33+50,100)) = 100
    d = oracle.drift_score(_emb(240, same=False), np.ones(240, np.uint8), 960, 30)
    assert (d["hits"], d["run"], d["score"]) == (224, 239, 100)


def test_score_short_clip_weight(oracle):
    # 100 frames @25 fps: step = int(25/7) = 3 -> total = ceil(100/3) = 34 sampled; all different:
    # 33 comparisons, hits = 33-15 = 18, pct = 52.94, conf = min(52.94*33/15, 100) = 100, N=100 <= 750 -> w = 0.3
    # score = int(min(52.94 + 30, 100)) = 82
    d = oracle.drift_score(_emb(34, same=False), np.ones(34, np.uint8), 100, 25)
    assert (d["hits"], d["run"], d["score"]) == (18, 33, 82)


def test_score_invalid_frames_do_not_touch_state(oracle):
    # frames without a face neither reset the run nor replace `previous` (model.py:48-75)
    e = _emb(40, same=False)
    v = np.ones(40, np.uint8); v[5:10] = 0
    d = oracle.drift_score(e, v, 160, 30)
    assert d["run"] == 34 and d["hits"] == 34 - 15      # 35 valid frames -> 34 comparisons
    assert (d["sims"][5:10] == 2.0).all() and d["sims"][0] == 2.0


def test_score_run_resets_on_similar_frame(oracle):
    e = _emb(60, same=False)
    e[30] = e[29]                                         # one identical pair -> run resets at i=30
    d = oracle.drift_score(e, np.ones(60, np.uint8), 240, 30)
    assert d["run"] == 29                                 # comparisons 31..59
    assert d["hits"] == (29 - 15) + (29 - 15)             # runs of 29 before and 29 after the reset
    assert oracle.drift_score(e[:0], np.zeros(0, np.uint8), 0, 30)["score"] == 0   # model.py:83-85


def test_nv12_to_bgr_known_values(oracle):
    """BT.601 limited range: (Y,U,V) = (16,128,128) -> black, (235,128,128) -> white, saturation at the ends."""
    H, W = 2, 4
    def frame(y, u, v):
        return np.array([y] * (H * W) + [u, v] * (W // 2), np.uint8)
    assert (oracle.nv12_to_bgr(frame(16, 128, 128), H, W) == 0).all()
    assert (oracle.nv12_to_bgr(frame(235, 128, 128), H, W) == 255).all()
    assert (oracle.nv12_to_bgr(frame(0, 128, 128), H, W) == 0).all()
    px = oracle.nv12_to_bgr(frame(81, 90, 240), H, W)[0, 0]           # BT.601 "red": B,G,R
    assert px[2] >= 250 and px[0] <= 5 and px[1] <= 5
    # against the floating-point BT.601 definition within 1 LSB
    rng = np.random.default_rng(2)
    nv = rng.integers(16, 236, (8 * 8 * 3 // 2,), dtype=np.uint8)
    got = oracle.nv12_to_bgr(nv, 8, 8).astype(np.float64)
    y = nv[:64].reshape(8, 8).astype(np.float64) - 16
    uv = nv[64:].reshape(4, 4, 2).astype(np.float64) - 128
    u = np.repeat(np.repeat(uv[..., 0], 2, 0), 2, 1); v = np.repeat(np.repeat(uv[..., 1], 2, 0), 2, 1)
    ref = np.stack([1.164 * y + 2.018 * u, 1.164 * y - 0.813 * v - 0.391 * u, 1.164 * y + 1.596 * v], -1).clip(0, 255)
    assert np.abs(got - ref).max() <= 1.0


def test_reciprocal_division_is_exact(oracle):
    """The pyramid kernel divides bin sums by kh and kw through precomputed reciprocals + two fmas
    (csrc/trl_pnet.hip:pyr_div, enabled for bins <= 96).  Every (kh, kw, sum) triple must reproduce the two
    IEEE divisions of the reference expression bit for bit -- exhaustive, ~5.5e9 cases, a few seconds."""
    assert oracle.selftest_recip_div(96) == 0


def test_greedy_nms_fixed_point_checker(oracle):
    """tests/conftest.py::assert_greedy_nms_fixed_point (the size-independent check the GPU suite applies to per-level pick lists of
    up to 736 k entries) against the oracle's own NMS on a PNet-like cell grid with many score ties: it accepts the sequential
    algorithm's picks and rejects a pick list with one entry removed or one suppressed box added."""
    from conftest import assert_greedy_nms_fixed_point
    rng = np.random.default_rng(0)
    oh, ow, sc = 40, 55, np.float32(0.6)
    cell = np.nonzero(rng.uniform(size=oh * ow) < 0.7)[0].astype(np.int32)
    cy, cx = (cell // ow).astype(np.float32), (cell % ow).astype(np.float32)
    score = rng.choice(np.linspace(0.3, 0.99, 50).astype(np.float32), size=len(cell))
    box = np.stack([np.floor((2 * cx + 1) / sc), np.floor((2 * cy + 1) / sc), np.floor((2 * cx + 12) / sc), np.floor((2 * cy + 12) / sc)], 1).astype(np.float32)
    keep = oracle.nms_iou(box, score, 0.5)                 # stable descending sort: ties -> lower index = lower cell first
    rec = np.zeros(len(cell), np.dtype([("box", np.float32, 4), ("score", np.float32), ("reg", np.float32, 4), ("cell", np.int32)]))
    rec["box"], rec["score"], rec["cell"] = box, score, cell
    assert 100 < len(keep) < len(cell)
    assert_greedy_nms_fixed_point(rec, keep.astype(np.int32), oh, ow, 0.5)
    with pytest.raises(AssertionError):
        assert_greedy_nms_fixed_point(rec, np.delete(keep, 5).astype(np.int32), oh, ow, 0.5)
    dropped = np.setdiff1d(np.arange(len(cell)), keep)[:1]
    pos = np.searchsorted(-score[keep], -score[dropped[0]], side="right")
    with pytest.raises(AssertionError):
        assert_greedy_nms_fixed_point(rec, np.insert(keep, pos, dropped[0]).astype(np.int32), oh, ow, 0.5)
