"""GPU tests at BASELINE.json's full sizes through size-independent properties, plus the larger
configurations (1080p multi-face, 4K) on single frames against the oracle."""
import numpy as np
import pytest
import torch

import truely_amd
from conftest import assert_greedy_nms_fixed_point

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clip720():
    return truely_amd.synthetic.synthetic_frames(256, 720, 1280, seed=0)


def test_full_batch_determinism_and_batch_independence(engine, clip720):
    """configs[1] at full size: candidates are appended with atomics in arbitrary order, yet every output is
    reproducible bit for bit, and a frame's result does not depend on which batch it travels in."""
    a = engine.detect_embed(clip720)
    b = engine.detect_embed(clip720)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(a[k], b[k]), k
    assert int(a["valid"].sum()) >= 200                     # "1-face frames": almost every frame yields a face
    idx = [0, 17, 101, 255]
    sub = engine.detect_embed(clip720[idx])
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(sub[k], a[k][idx]), k
    rev = engine.detect_embed(clip720[::-1].copy())          # reversed frame order
    assert torch.equal(rev["emb"].flip(0), a["emb"]) and torch.equal(rev["rect"].flip(0), a["rect"])
    # embeddings are unit vectors; invalid rows are exactly zero
    nrm = a["emb"].norm(dim=1)
    v = a["valid"].bool()
    assert torch.allclose(nrm[v], torch.ones_like(nrm[v]), atol=1e-5) and (nrm[~v] == 0).all()


def test_full_batch_sampled_against_oracle(engine, oracle, clip720):
    engine.poison_workspaces(0xFF)
    out = engine.detect_embed(clip720)
    for i in (3, 128, 250):
        ref = oracle.detect_embed(clip720[i:i + 1])
        assert np.array_equal(out["rect"][i].cpu().numpy(), ref["rect"][0])
        assert out["valid"][i].item() == ref["valid"][0]
        assert np.array_equal(out["emb"][i].cpu().numpy(), ref["emb"][0])


def test_drift_is_a_function_of_time_order_only(engine, clip720):
    """Splitting the clip into shards and concatenating embeddings (the multi-GPU path) changes nothing."""
    full = engine.detect_embed(clip720[:64])
    parts = [engine.detect_embed(clip720[i:i + 16]) for i in range(0, 64, 16)]
    emb = torch.cat([p["emb"] for p in parts]); valid = torch.cat([p["valid"] for p in parts])
    d0 = engine.drift_score(full["emb"], full["valid"], 256, 30)
    d1 = engine.drift_score(emb, valid, 256, 30)
    assert d0["score"] == d1["score"] and torch.equal(d0["sims"], d1["sims"]) and torch.equal(d0["flags"], d1["flags"])


def test_config0_whole_clip_through_run(engine, oracle, tmp_path, monkeypatch):
    """BASELINE configs[0] at its stated shape through the drop-in boundary: a 960-frame 640x360 30 fps clip, `run()` samples
    every 4th frame (240 analysed), streams them in windows and writes all 960 frames.  The reference's sample .mp4 cannot be
    decoded here, so the clip is seeded synthetic frames in the raw container: 12 distinct frames, each held for 80 frames, so
    the oracle's per-frame results (frames are independent up to the drift scan) give the expected score of the whole clip."""
    from truely_amd import engine as eng_mod, model, video_io
    monkeypatch.setattr(eng_mod, "_default", engine)
    monkeypatch.setenv("TRUELY_ANNOTATE", "0")
    H, W, fps, N = 360, 640, 30, 960
    uniq = truely_amd.synthetic.synthetic_frames(12, H, W, seed=21)
    src, dst = str(tmp_path / "c0.trlv"), str(tmp_path / "c0_out.trlv")
    wr = video_io.RawWriter(src, fps, (W, H))
    for i in range(N):
        wr.write(uniq[i // 80])
    wr.release()
    score = model.run(src, dst)
    r = oracle.detect_embed(uniq)
    idx = np.arange(0, N, 4) // 80                                   # the 240 sampled frames
    d = oracle.drift_score(r["emb"][idx], r["valid"][idx], N, fps)
    assert len(idx) == 240 and score == d["score"]
    rd, ofps, ow, oh = video_io.open_reader(dst)
    assert (ofps, ow, oh, rd.n) == (fps, W, H, N)
    # and as ONE 240-frame batch (bench.py --config 0): same embeddings as the oracle's, frame for frame
    out = engine.detect_embed(uniq[idx])
    assert np.array_equal(out["emb"].cpu().numpy(), r["emb"][idx]) and np.array_equal(out["valid"].cpu().numpy(), r["valid"][idx])


_FRAME_CACHE = {}


def _frame(H, W, faces, seed):
    """One seeded frame (the numpy generator needs ~15 s for a 4K frame: shared between tests)."""
    key = (H, W, faces, seed)
    if key not in _FRAME_CACHE:
        _FRAME_CACHE[key] = truely_amd.synthetic.synthetic_frames(1, H, W, seed=seed, faces=faces)
    return _FRAME_CACHE[key]


@pytest.fixture(scope="module")
def clip1080():
    """BASELINE configs[2] at full size: 128 x 1080p frames with 3-5 faces each (bench.make_clip: 16 generated frames,
    the rest horizontal rolls of them -- different bytes, same statistics)."""
    import bench
    return bench.make_clip(bench.CONFIGS[2], 128, seed=0)


def test_config2_full_batch_properties(engine, blob, clip1080):
    """configs[2] as stated (1080p multi-face stream, batch=128): reproducible bit for bit, independent of the batch a
    frame travels in, and the bf16-MFMA embedder leaves every DECISION (boxes, rects, valid mask) untouched while its
    embeddings stay within the stated tolerance of the f32 path (cosine >= 0.99; NOT the 1e-4 bar of the default)."""
    from truely_amd.engine import Engine
    a = engine.detect_embed(clip1080)
    b = engine.detect_embed(clip1080)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(a[k], b[k]), k
    assert int(a["valid"].sum()) >= 100
    idx = [0, 15, 16, 77, 127]
    sub = engine.detect_embed(clip1080[idx])
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(sub[k], a[k][idx]), k
    counts = engine.mtcnn_detect(clip1080[:16])[2].cpu().numpy()
    assert counts.max() >= 2                               # multi-face frames: several boxes survive the cascade
    assert engine.levels(1080, 1920) == 12 and len(engine.level_counts(0)[0]) == 12
    eng16 = Engine(blob, embed_precision="bf16")
    c = eng16.detect_embed(clip1080)
    for k in ("box", "prob", "rect", "valid"):
        assert torch.equal(c[k], a[k]), k
    v = a["valid"].bool()
    cos = (c["emb"][v] * a["emb"][v]).sum(1)
    assert float(cos.min()) >= 0.99, float(cos.min())
    assert (c["emb"][~v] == 0).all()
    c2 = eng16.detect_embed(clip1080)
    assert torch.equal(c2["emb"], c["emb"])                # the bf16 path is deterministic too


def test_config2_sampled_against_oracle(engine, oracle, clip1080):
    engine.poison_workspaces(0xFF)
    out = engine.detect_embed(clip1080)
    for i in (5, 64, 120):
        ref = oracle.detect_embed(clip1080[i:i + 1])
        assert np.array_equal(out["box"][i].cpu().numpy(), ref["box"][0])
        assert np.array_equal(out["rect"][i].cpu().numpy(), ref["rect"][0])
        assert out["valid"][i].item() == ref["valid"][0]
        assert np.array_equal(out["emb"][i].cpu().numpy(), ref["emb"][0])


def test_config4_as_stated_4k_twelve_levels_fp16_batch(blob):
    """BASELINE configs[4] in ONE run: a batch of 4K frames, MTCNN pyramid of 12 scales (min_face_size=40), the fp16-MFMA
    InceptionResnetV1 embedder.  Against the oracle run with the same MTCNN(min_face_size=40) parameters: every decision of the
    detector (which stays f32: an fp16 detector cannot keep box / NMS index parity) is bit-exact -- per-level
    candidate and keep counts, the boxes after every stage, the crop rectangles, the valid mask -- and the fp16 embeddings meet
    the stated tolerance of the reduced-precision mode (cosine >= 0.9999 on detected faces, max |diff| <= 3e-3; the f32 embedder
    on the same batch is bit-exact)."""
    from oracle.oracle import Oracle
    from truely_amd.engine import Engine
    base = _frame(2160, 3840, 1, 32)[0]
    fr = np.stack([base, np.roll(base, 37, axis=1), np.roll(base, 977, axis=1), np.roll(np.roll(base, 301, axis=1), 55, axis=0)])
    orc = Oracle(blob)
    orc.params.min_face_size = 40
    ref = orc.detect_embed(fr)
    eng16 = Engine(blob, min_face_size=40, cap_level=3072, cap_frame=3072, embed_precision="fp16")
    eng32 = Engine(blob, min_face_size=40, cap_level=3072, cap_frame=3072)
    assert eng16.levels(2160, 3840) == 12
    for eng in (eng16, eng32):
        eng.detect_embed(fr)
        eng.poison_workspaces(0xFF)
        out = eng.detect_embed(fr)
        for i in range(len(fr)):
            _b, _p, tr = orc.detect(fr[i], trace=True)
            cand, keep = eng.level_counts(i)
            assert len(cand) == 12 and cand == tr["n_cand_scale"] and keep == tr["n_keep_scale"], i
            for s in (1, 2, 3):
                assert np.array_equal(eng.stage_boxes(s, i), tr[f"boxes{s}"]), (i, s)
        for k in ("box", "prob", "rect", "valid"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
        emb = out["emb"].cpu().numpy()
        v = ref["valid"].astype(bool)
        assert v.sum() >= 2
        if eng is eng32:
            assert np.array_equal(emb, ref["emb"])
        else:
            assert (emb[~v] == 0).all()
            assert ((emb[v] * ref["emb"][v]).sum(1)).min() >= 0.9999
            assert np.abs(emb[v] - ref["emb"][v]).max() <= 3e-3
            d = eng.drift_score(out["emb"], out["valid"], len(fr) * 4, 30)
            dref = orc.drift_score(ref["emb"], ref["valid"], len(fr) * 4, 30)
            assert np.abs(d["sims"].cpu().numpy() - np.asarray(dref["sims"])).max() <= 5e-3


@pytest.mark.parametrize("H,W,faces,seed", [(1080, 1920, -1, 31), (2160, 3840, 1, 32)])
def test_large_frames_against_oracle(engine, blob, oracle, H, W, faces, seed):
    """configs[2] (1080p, 3-5 faces) and configs[4] (4K, min_face_size=20 -> 14 pyramid levels), fp32 path."""
    fr = _frame(H, W, faces, seed)    # (the 4K frame has ~3x the candidates of the default START capacity: the lists grow)
    engine.detect_embed(fr)          # sizes the workspaces ...
    engine.poison_workspaces(0xFF)   # ... which are then NaN-filled, like the LDS: see test_results_do_not_depend_on_stale_memory
    out = engine.detect_embed(fr)
    _b, _p, tr = oracle.detect(fr[0], trace=True)
    cand, keep = engine.level_counts(0)
    assert cand == tr["n_cand_scale"] and keep == tr["n_keep_scale"]
    assert len(cand) == (12 if H == 1080 else 14)
    for s in (1, 2, 3):
        assert np.array_equal(engine.stage_boxes(s, 0), tr[f"boxes{s}"])
    ref = oracle.detect_embed(fr)
    assert np.array_equal(out["rect"].cpu().numpy(), ref["rect"]) and np.array_equal(out["emb"].cpu().numpy(), ref["emb"])


def _check_crowded(eng, orc, fr):
    """One frame, every record of the cascade against the oracle (lists of any length), then the delivered outputs."""
    eng.poison_workspaces(0xFF)
    out = eng.detect_embed(fr)
    _b, _p, tr = orc.detect(fr[0], trace=True, max_trace=1 << 20)
    cand, keep = eng.level_counts(0)
    assert cand == tr["n_cand_scale"] and keep == tr["n_keep_scale"], (cand, tr["n_cand_scale"])
    for s in (1, 2, 3):
        got = eng.stage_boxes(s, 0)
        assert got.shape == tr[f"boxes{s}"].shape and np.array_equal(got, tr[f"boxes{s}"]), s
    ref = orc.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    return tr, eng.list_stats()


def test_default_engine_on_a_720p_uniform_noise_frame(blob, oracle):
    """Content that defeats every fixed capacity: uniform noise at 720p fires ~4,800 PNet cells at the finest level alone, ~5,600
    boxes enter R-Net and ~2,500 survive O-Net.  The reference (`mtcnn.detect`, server/model.py:47) returns boxes for ANY frame;
    a DEFAULT context does too: lists grow (re-run), the long ones are sorted / suppressed in global memory, results = oracle."""
    from truely_amd.engine import Engine
    eng = Engine(blob)
    fr = np.random.default_rng(1).integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8)
    tr, st = _check_crowded(eng, oracle, fr)
    assert max(tr["n_cand_scale"]) > 2048 and len(tr["boxes1"]) > 2048 and len(tr["boxes3"]) > 2048   # every stage beyond the LDS tier
    assert st["attempts"] >= 2 and st["spill_lists"] >= 4, st
    _check_crowded(eng, oracle, fr)
    assert eng.list_stats()["attempts"] == 1              # the capacities now follow the content
    calm = truely_amd.synthetic.synthetic_frames(1, 720, 1280, seed=0)
    _check_crowded(eng, oracle, calm)                     # ... and a calm frame on the grown context is unchanged


def test_default_engine_on_a_4k_frame_at_a_low_pnet_threshold(blob):
    """A 4K frame (min_face_size 20: 14 levels, 739 k PNet cells at the finest) at thr0 = 0.55: ~30 k candidates at the finest
    level alone, far past the LDS tier -- default capacities, results = oracle.  (At thr0 = 0.3 this frame fires 736,593 of the
    739,312 cells of level 0 and ~1 M boxes enter R-Net: the device path runs it -- tools/crowded_timing.py, DESIGN.md -- but the
    CPU oracle needs the better part of an hour for it, so the parity case sits at the threshold the oracle finishes in seconds.)"""
    from oracle.oracle import Oracle
    from truely_amd.engine import Engine
    eng = Engine(blob, thresholds=(0.55, 0.7, 0.7))
    orc = Oracle(blob)
    orc.params.thr0 = 0.55
    fr = _frame(2160, 3840, 1, 32)
    tr, st = _check_crowded(eng, orc, fr)
    assert tr["n_cand_scale"][0] > 10000 and st["spill_lists"] >= 1, (tr["n_cand_scale"], st)


def test_default_engine_on_the_pathological_4k_frame(blob):
    """The frame the reference would still accept and no CPU check finishes on: 4K at thr0 = 0.3 fires 736,593 of the finest
    level's 739,312 PNet cells (1.5 M candidates over 14 levels, 394 k boxes into R-Net).  A DEFAULT engine runs it, and what can be
    checked at this size without an O(n^2) reference is checked exactly: the candidate records of three levels against the
    oracle's PNet maps (counts, scores, regressions, generateBoundingBox boxes), and the per-level NMS picks of those levels --
    lists of up to 736 k entries through the spill tier -- against the DEFINITION of greedy NMS (a fixed point that is unique)."""
    from oracle.oracle import Oracle
    from truely_amd.engine import Engine
    eng = Engine(blob, thresholds=(0.3, 0.7, 0.7))
    orc = Oracle(blob)
    fr = _frame(2160, 3840, 1, 32)
    out = eng.detect_embed(fr)
    st = eng.list_stats()
    assert st["max_level_count"] > 700000 and st["spill_lists"] >= 8, st
    cand, keep = eng.level_counts(0)
    scales = orc.scales(2160, 3840)
    assert len(cand) == len(scales) == 14
    for lvl in (0, 2, 6):
        sc, h, w = scales[lvl]
        prob, reg = orc.pnet_level(orc.area_resample_norm(fr[0], 0, 2160, 0, 3840, h, w))
        oh, ow = prob.shape
        rec, idx = eng.level_keep(0, lvl)
        assert len(rec) == cand[lvl] == int((prob >= np.float32(0.3)).sum()) and len(idx) == keep[lvl], lvl
        cy, cx = rec["cell"] // ow, rec["cell"] % ow
        assert np.array_equal(rec["score"], prob[cy, cx]) and np.array_equal(rec["reg"], reg[cy, cx]), lvl
        fs = np.float32(sc)
        fx, fy = cx.astype(np.float32), cy.astype(np.float32)
        exp = np.stack([np.floor((np.float32(2) * fx + np.float32(1)) / fs), np.floor((np.float32(2) * fy + np.float32(1)) / fs),
                        np.floor((np.float32(2) * fx + np.float32(12)) / fs), np.floor((np.float32(2) * fy + np.float32(12)) / fs)], axis=1)
        assert np.array_equal(rec["box"], exp), lvl
        assert_greedy_nms_fixed_point(rec, idx, oh, ow, 0.5)
    # Stage 1b (batched_nms 0.7 over all levels, 394 k boxes, one list through the spill tier) and the stage-1 boxes.  Boxes of
    # different levels cannot reach IoU 0.7 (IoU <= min area / max area, checked from the records), and inside a level the 0.5
    # survivors cannot either: the 0.7 pass keeps everything, so the stage-1 list is every level's picks merged by score (ties:
    # level order, then pick order), regressed with the PNet offsets (w, h without +1), squared (rerec), minus the boxes whose
    # clipped window is empty -- all of it float32 arithmetic that numpy reproduces operation by operation.
    recs = [eng.level_keep(0, l) for l in range(len(cand))]
    picks = [r[i] for r, i in recs]
    amin = [float(((p["box"][:, 2] - p["box"][:, 0]) * (p["box"][:, 3] - p["box"][:, 1])).min()) if len(p) else None for p in picks]
    amax = [float(((p["box"][:, 2] - p["box"][:, 0]) * (p["box"][:, 3] - p["box"][:, 1])).max()) if len(p) else None for p in picks]
    live = [l for l in range(len(picks)) if len(picks[l])]
    for a in live:
        for b in live:
            if a < b:
                assert amax[a] / amin[b] < 0.7, (a, b)      # finer level a: smaller boxes
    allp = np.concatenate([picks[l] for l in live])
    order = np.argsort(-allp["score"].astype(np.float64), kind="stable")          # descending score, ties in concatenation order
    c = allp[order]
    x1, y1, x2, y2 = (c["box"][:, q] for q in range(4))
    rw, rh = x2 - x1, y2 - y1
    bx1, by1 = x1 + c["reg"][:, 0] * rw, y1 + c["reg"][:, 1] * rh
    bx2, by2 = x2 + c["reg"][:, 2] * rw, y2 + c["reg"][:, 3] * rh
    h, w = by2 - by1, bx2 - bx1
    side = np.maximum(w, h)
    half = np.float32(0.5)
    qx1 = bx1 + w * half - side * half
    qy1 = by1 + h * half - side * half
    qx2, qy2 = qx1 + side, qy1 + side
    tx, ty, tex, tey = (np.trunc(v).astype(np.int64) for v in (qx1, qy1, qx2, qy2))
    ok = (np.minimum(tey, 2160) > np.maximum(ty, 1) - 1) & (np.minimum(tex, 3840) > np.maximum(tx, 1) - 1)
    exp1 = np.stack([qx1, qy1, qx2, qy2, c["score"]], axis=1)[ok].astype(np.float32)
    got1 = eng.stage_boxes(1, 0)
    assert got1.shape == exp1.shape and len(got1) > 300000
    assert np.array_equal(got1, exp1)
    assert out["valid"].shape == (1,)


def test_frames_without_candidates(engine, oracle):
    """A flat frame: PNet fires nowhere or NMS leaves nothing -> detect() is None, valid = 0, zero embedding."""
    engine.poison_workspaces(0xFF)   # NaN-filled workspaces and LDS: see test_results_do_not_depend_on_stale_memory
    fr = np.full((2, 240, 320, 3), 127, np.uint8)
    out = engine.detect_embed(fr)
    ref = oracle.detect_embed(fr)
    assert np.array_equal(out["valid"].cpu().numpy(), ref["valid"])
    assert np.array_equal(out["emb"].cpu().numpy(), ref["emb"])
    d = engine.drift_score(out["emb"], out["valid"], 8, 30)
    assert d["score"] == oracle.drift_score(ref["emb"], ref["valid"], 8, 30)["score"]


def test_minimum_frame_size(engine, oracle):
    """12x12 is the smallest frame with a pyramid level (min(h,w)*0.6 >= 12 fails below 20 px)."""
    for (H, W) in [(20, 20), (21, 37)]:
        fr = truely_amd.synthetic.synthetic_frames(2, H, W, seed=5)
        out = engine.detect_embed(fr)
        ref = oracle.detect_embed(fr)
        assert np.array_equal(out["valid"].cpu().numpy(), ref["valid"]) and np.array_equal(out["rect"].cpu().numpy(), ref["rect"])
