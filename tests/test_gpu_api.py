"""GPU tests of the drop-in API surface (MTCNN.detect, InceptionResnetV1.__call__, model.run,
analyze_video) and of the committed golden vectors through the HIP path."""
import glob
import os

import numpy as np
import pytest
import torch

import truely_amd
from conftest import frames_small

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "clip_*.npz"))))
def test_hip_path_reproduces_golden(engine, path):
    z = np.load(path)
    n, H, W, seed = int(z["n"]), int(z["H"]), int(z["W"]), int(z["seed"])
    fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed, faces=int(z["faces_per_frame"]) if "faces_per_frame" in z else 1)
    out = engine.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert np.array_equal(out[k].cpu().numpy(), z[k]), k
    for i in range(n):
        cand, keep = engine.level_counts(i)
        assert cand == z[f"f{i}_cand"].tolist() and keep == z[f"f{i}_keep"].tolist()
        for s in (1, 2, 3):
            assert np.array_equal(engine.stage_boxes(s, i), z[f"f{i}_boxes{s}"])
    d = engine.drift_score(out["emb"], out["valid"], n * 4, 30)
    assert d["score"] == int(z["score"]) and np.array_equal(d["sims"].cpu().numpy(), z["sims"])


def test_mtcnn_detect_api(engine, oracle):
    from truely_amd.mtcnn import MTCNN
    m = MTCNN(engine=engine)
    fr = truely_amd.synthetic.synthetic_frames(3, 360, 640, seed=11)
    boxes, probs = m.detect(fr[0])                      # single frame -> unbatched, like facenet-pytorch
    rb, rp = oracle.detect(fr[0])
    assert (boxes is None) == (rb is None)
    if boxes is not None:
        assert boxes.dtype == np.float32 and np.array_equal(boxes, rb) and np.array_equal(probs, rp)
    bb, pp = m.detect(fr)                               # batch -> object arrays
    assert len(bb) == 3
    for i in range(3):
        rb, rp = oracle.detect(fr[i])
        assert (bb[i] is None) == (rb is None)
        if rb is not None:
            assert np.array_equal(bb[i], rb)
    none_b, none_p = m.detect(np.full((64, 64, 3), 128, np.uint8))
    if none_b is None:
        assert none_p == [None]


def test_inception_resnet_api(engine, oracle):
    from truely_amd.inception_resnet_v1 import InceptionResnetV1
    net = InceptionResnetV1(pretrained="vggface2", engine=engine).eval()
    x = torch.rand(2, 3, 80, 80)                        # what to_tensor(...).unsqueeze(0) yields, batched
    y = net(x)
    assert y.shape == (2, 512) and y.device.type == "cpu"
    ref = oracle.facenet(x.permute(0, 2, 3, 1).contiguous().numpy())
    assert np.array_equal(y.numpy(), ref)


def test_run_end_to_end_raw_container(engine, oracle, tmp_path, monkeypatch):
    from truely_amd import engine as eng_mod, model, video_io
    monkeypatch.setattr(eng_mod, "_default", engine)
    H, W, fps = 180, 320, 30
    fr = truely_amd.synthetic.synthetic_frames(40, H, W, seed=3)
    src, dst = str(tmp_path / "in.trlv"), str(tmp_path / "out.trlv")
    video_io.write_raw(src, fr, fps)
    score = model.run(src, dst)
    # oracle: sample every 4th frame, same path
    sampled = fr[::4]
    r = oracle.detect_embed(sampled)
    d = oracle.drift_score(r["emb"], r["valid"], 40, fps)
    assert score == d["score"] and 0 <= score <= 100
    assert os.path.getsize(dst) > 0                     # server.py:612-627 requires a non-empty output
    rd, ofps, ow, oh = video_io.open_reader(dst)
    assert (ofps, ow, oh, rd.n) == (fps, W, H, 40)      # every frame is written (model.py:77)


def test_analyze_video_batching_invariant(engine):
    from truely_amd.model import analyze_video
    fr = frames_small(10, 180, 320, seed=8)
    a = analyze_video(fr, fps=30, engine=engine)
    b = analyze_video(fr, fps=30, engine=engine, batch=3)    # ragged batches 3+3+3+1
    assert a["score"] == b["score"]
    assert torch.equal(a["emb"], b["emb"]) and torch.equal(a["valid"], b["valid"]) and torch.equal(a["sims"], b["sims"])


def test_candidate_lists_grow_instead_of_failing(blob, engine):
    """detect_face() has no candidate limit (server/model.py:47 never refuses a frame), so a context whose START capacities
    are far too small for the content re-runs the call with larger lists and delivers what a roomy context delivers."""
    from truely_amd.engine import Engine
    small = Engine(blob, cap_level=64, cap_frame=64)
    fr = truely_amd.synthetic.synthetic_frames(2, 720, 1280, seed=0)
    got, ref = small.detect_embed(fr), engine.detect_embed(fr)
    st = small.list_stats()
    assert st["attempts"] >= 2 and st["max_level_count"] > 64
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(got[k], ref[k]), k
    small.detect_embed(fr)
    assert small.list_stats()["attempts"] == 1           # the capacities follow the content: the next call fits at once


def test_bad_arguments_are_rejected(engine):
    from truely_amd._lib import TrlError
    with pytest.raises(ValueError):
        engine.detect_embed(np.zeros((1, 8, 8, 4), np.uint8))
    with pytest.raises(TrlError):
        engine.detect_embed(np.zeros((1, 8, 8, 3), np.uint8))      # smaller than the 12x12 PNet field
    with pytest.raises(TrlError):
        engine.facenet_embed(torch.zeros(1, 40, 40, 3))            # too small for the stem


def test_malformed_weight_blobs_are_rejected(blob):
    """trl_load_weights trusts nothing in the blob's directory: bad magic, a truncated blob, an entry whose byte range leaves the
    blob, a shape that does not account for the entry's bytes and a missing tensor are all TRL_ERR_WEIGHTS, not a crash."""
    import struct
    from truely_amd.engine import Engine
    from truely_amd._lib import TrlError
    from truely_amd import weights
    (n,) = struct.unpack_from("<I", blob, 8)
    ent = weights._ENTRY

    def patched(i, **kw):
        b = bytearray(blob)
        f = list(ent.unpack_from(b, 16 + i * ent.size))
        for k, v in kw.items():
            f[{"ndim": 1, "d0": 2, "d1": 3, "off": 6, "nb": 7}[k]] = v
        ent.pack_into(b, 16 + i * ent.size, *f)
        return bytes(b)
    name, ndim, d0, d1, _, _, off, nb = ent.unpack_from(blob, 16)
    bad = [b"XXXXXXXX" + blob[8:], blob[:16 + ent.size * n - 8], patched(0, off=len(blob) - 4), patched(0, off=2 ** 63, nb=2 ** 63),
           patched(0, d0=d0 + 1), patched(0, nb=nb - 4)]
    t = weights.unpack_tensors(blob)
    t.pop("pnet.conv4_2.w")
    bad.append(weights.pack_tensors(t))
    for b in bad:
        with pytest.raises(TrlError) as e:
            Engine(b)
        assert e.value.status == -3, str(e.value)               # TRL_ERR_WEIGHTS


def test_nv12_ingest_with_sampling(engine, oracle):
    """NV12 -> BGR of every step-th frame on the device equals the oracle per frame."""
    rng = np.random.default_rng(12)
    n, H, W, step = 9, 48, 64, 4
    nv12 = rng.integers(0, 256, (n, H * W * 3 // 2), dtype=np.uint8)
    nv12[0, :16] = [0, 255, 16, 235, 15, 17, 234, 236, 128, 1, 254, 100, 200, 50, 75, 3]   # range edges
    out = engine.ingest_nv12(nv12, H, W, step).cpu().numpy()
    assert out.shape == (3, H, W, 3)                                    # frames 0, 4, 8 (model.py:46)
    for j, i in enumerate(range(0, n, step)):
        assert np.array_equal(out[j], oracle.nv12_to_bgr(nv12[i], H, W)), i
    with pytest.raises(Exception):
        engine.ingest_nv12(nv12[:, :-3], H, W, step)


def test_i420_ingest_equals_nv12_ingest(engine, oracle):
    """Planar 4:2:0 (YUV4MPEG2 files, software decoders) converts on the device without host-side repacking: same bytes as the
    NV12 path on the interleaved copy of the same chroma, and as the oracle."""
    rng = np.random.default_rng(13)
    n, H, W, step = 7, 36, 52, 3
    nv12 = rng.integers(0, 256, (n, H * W * 3 // 2), dtype=np.uint8)
    i420 = nv12.copy()
    i420[:, H * W:H * W + H * W // 4] = nv12[:, H * W::2]
    i420[:, H * W + H * W // 4:] = nv12[:, H * W + 1::2]
    a = engine.ingest_nv12(nv12, H, W, step).cpu().numpy()
    b = engine.ingest_nv12(i420, H, W, step, planar=True).cpu().numpy()
    assert a.shape == (3, H, W, 3) and np.array_equal(a, b)
    assert np.array_equal(b[1], oracle.nv12_to_bgr(nv12[3], H, W))


def test_run_with_the_output_stage_skipped(engine, oracle, tmp_path, monkeypatch):
    """TRUELY_WRITE_OUTPUT=0 (the analysis alone): only the sampled frames are read from containers with fixed-size frames, in
    windows over two contexts with the embedder grouped over four windows -- several groups, a ragged last window, and a clip
    shorter than one window.  BGR, NV12 and YUV4MPEG2 clips: the score is the oracle's on the sampled frames."""
    from truely_amd import engine as eng_mod, model, video_io
    from truely_amd.ingest import bgr_to_nv12
    monkeypatch.setattr(eng_mod, "_default", engine)
    monkeypatch.setenv("TRUELY_WRITE_OUTPUT", "0")
    H, W, fps = 96, 128, 30
    fr = truely_amd.synthetic.synthetic_frames(30, H, W, seed=3)
    fr = np.concatenate([fr, fr[::-1], fr])[:87 * 4 - 2]                 # 346 frames -> 87 sampled
    garbage = np.random.default_rng(0).integers(0, 256, fr.shape, dtype=np.uint8)
    clip = np.where((np.arange(len(fr)) % 4 == 0)[:, None, None, None], fr, garbage)    # the frames in between are never analysed
    nv = bgr_to_nv12(clip)
    a, b, c = str(tmp_path / "in.trlv"), str(tmp_path / "in_nv12.trlv"), str(tmp_path / "in.y4m")
    video_io.write_raw(a, clip, fps)
    video_io.write_raw(b, nv, fps, pixfmt="nv12", size=(W, H))
    video_io.write_y4m(c, nv, fps, (W, H))
    r = oracle.detect_embed(clip[::4])
    exp = oracle.drift_score(r["emb"], r["valid"], len(clip), fps)["score"]
    bgr = np.stack([oracle.nv12_to_bgr(f, H, W) for f in nv[::4]])
    r2 = oracle.detect_embed(bgr)
    exp_yuv = oracle.drift_score(r2["emb"], r2["valid"], len(clip), fps)["score"]
    for win_bytes in (16 * H * W * 3, 10 ** 9):                          # 16-frame windows (6 windows: 4 + 2 in two embedder groups), one window
        monkeypatch.setattr(model, "WINDOW_BYTES", win_bytes)
        assert model.run(a, str(tmp_path / "none.avi")) == exp
        assert model.run(b, str(tmp_path / "none.avi")) == exp_yuv
        assert model.run(c, str(tmp_path / "none.avi")) == exp_yuv
    assert not os.path.exists(str(tmp_path / "none.avi"))


def test_aligned_crop_kernel_bit_exact(engine, oracle):
    """Embedding mode 3's crop on chosen landmark sets: upright, rotated, scaled, partly outside the frame (replicated borders),
    degenerate (all five points equal: scale 0, every sample is one pixel) -- device = oracle bit for bit, and an exact
    template (identity transform, up to the f32 rounding of the points) reproduces the frame's top-left 160x160 pixels."""
    rng = np.random.default_rng(5)
    H, W = 300, 420
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    tx = np.array([54.706571428571436, 105.04542857142857, 80.036, 59.35614285714286, 101.04271428571428])
    ty = np.array([73.85185714285714, 73.57342857142856, 102.48085714285713, 131.9507142857143, 131.72014285714286])

    def place(scale, deg, cx, cy, jitter=0.0):
        t = np.deg2rad(deg)
        x = (tx - 80.0) * scale, (ty - 102.7) * scale
        px = np.cos(t) * x[0] - np.sin(t) * x[1] + cx + rng.normal(0, jitter, 5)
        py = np.sin(t) * x[0] + np.cos(t) * x[1] + cy + rng.normal(0, jitter, 5)
        return np.concatenate([px, py]).astype(np.float32)
    sets = [np.concatenate([tx, ty]).astype(np.float32), place(1.0, 0, 200, 150), place(0.6, 17, 210, 140, 1.5), place(1.7, -33, 100, 220, 2.0),
            place(2.5, 5, 10, 10), place(1.2, 180, 400, 280, 1.0), place(0.3, 90, 415, 3), np.full(10, 123.25, np.float32),
            place(40.0, 45, 5000, -3000)]
    pts = torch.from_numpy(np.stack(sets))
    n = len(sets)
    valid = torch.ones(n, dtype=torch.uint8); valid[3] = 0
    frames = np.broadcast_to(img, (n, H, W, 3)).copy()
    for rgb in (True, False):
        got = engine.crop_aligned(frames, pts, valid, rgb=rgb).cpu().numpy()
        for i in range(n):
            ref = oracle.crop_aligned(img, sets[i], rgb=rgb) if valid[i] else np.zeros((160, 160, 3), np.float32)
            assert np.array_equal(got[i], ref), (i, rgb)
    ident = engine.crop_aligned(frames[:1], pts[:1], valid[:1], rgb=False).cpu().numpy()[0]
    assert np.abs(ident - (img[:160, :160].astype(np.float32) - 127.5) / 128).max() < 1e-3


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_native_embedding_mode(blob, oracle, mode):
    """160x160 area-resampled, standardised (optionally RGB) crops behind a flag (modes 1, 2) and the
    landmark-aligned RGB crop (mode 3); the default (mode 0) stays the reference's 80x80 BGR /255 path."""
    from truely_amd.engine import Engine
    eng = Engine(blob, embed_mode=mode)
    fr = truely_amd.synthetic.synthetic_frames(3, 360, 640, seed=11)
    out = eng.detect_embed(fr)
    ref = oracle.detect_embed_mode(fr, mode)
    assert ref["valid"].sum() >= 1
    assert np.array_equal(out["valid"].cpu().numpy(), ref["valid"]) and np.array_equal(out["rect"].cpu().numpy(), ref["rect"])
    emb = out["emb"].cpu().numpy()
    assert np.abs(emb - ref["emb"]).max() <= 1e-4
    assert np.array_equal(emb, ref["emb"])
    base = oracle.detect_embed(fr)["emb"]
    assert not np.array_equal(base, ref["emb"])        # a different crop pipeline, not the parity default


def test_bf16_embedder_mode(blob, oracle):
    """trl_config.embed_precision = 1 (BASELINE configs[2], 'bf16 MFMA'): InceptionResnetV1 on bf16 activations /
    weights with f32 accumulation.  The detector stays f32-exact (identical boxes / rects / valid mask); the
    embeddings must agree with the f32 oracle within the tolerance stated here (NOT the 1e-4 bar of the default path):
    cosine >= 0.999 / max |diff| <= 3e-2 on unit vectors for random inputs, cosine >= 0.99 for detected faces, and
    the clip-level drift similarities within 5e-2 (measured 2.2e-2 with the seeded random weights, whose
    embeddings of two different synthetic faces are far less separated than a trained network's)."""
    from truely_amd.engine import Engine
    eng = Engine(blob, embed_precision="bf16")
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 1, (6, 80, 80, 3)).astype(np.float32)
    ref = oracle.facenet(x)
    got = eng.facenet_embed(torch.from_numpy(x)).cpu().numpy()
    cos = (got * ref).sum(1)
    assert np.all(np.isfinite(got)) and np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    assert cos.min() >= 0.999, cos
    assert np.abs(got - ref).max() <= 3e-2, np.abs(got - ref).max()
    x160 = rng.uniform(0, 1, (2, 160, 160, 3)).astype(np.float32)
    cos160 = (eng.facenet_embed(torch.from_numpy(x160)).cpu().numpy() * oracle.facenet(x160)).sum(1)
    assert cos160.min() >= 0.999, cos160
    # whole path: detection identical, embeddings / similarities close
    fr = truely_amd.synthetic.synthetic_frames(6, 360, 640, seed=11)
    out, exp = eng.detect_embed(fr), oracle.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid"):
        assert np.array_equal(out[k].cpu().numpy(), exp[k]), k
    v = exp["valid"].astype(bool)
    assert v.sum() >= 2
    e = out["emb"].cpu().numpy()
    assert ((e[v] * exp["emb"][v]).sum(1)).min() >= 0.99
    d = eng.drift_score(out["emb"], out["valid"], len(fr) * 4, 30)
    dref = oracle.drift_score(exp["emb"], exp["valid"], len(fr) * 4, 30)
    sims, sref = d["sims"].cpu().numpy(), np.asarray(dref["sims"])
    assert np.abs(sims - sref).max() <= 5e-2


def test_fp16_embedder_mode(blob, oracle):
    """trl_config.embed_precision = 2 (BASELINE configs[4], 'fp16'): the same reduced-precision embedder on
    v_mfma_f32_32x32x16_f16.  Three more mantissa bits than bf16, so the stated tolerance is tighter: cosine >= 0.99999 and
    max |diff| <= 3e-3 on unit vectors for random inputs, cosine >= 0.9999 on detected faces, drift similarities within 5e-3.
    Decisions (boxes, rects, valid mask) are the f32 path's, bit for bit.  No value overflows fp16's range (all finite)."""
    from truely_amd.engine import Engine
    eng = Engine(blob, embed_precision="fp16")
    rng = np.random.default_rng(9)
    x = rng.uniform(0, 1, (6, 80, 80, 3)).astype(np.float32)
    ref = oracle.facenet(x)
    got = eng.facenet_embed(torch.from_numpy(x)).cpu().numpy()
    cos = (got * ref).sum(1)
    assert np.all(np.isfinite(got)) and np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    assert cos.min() >= 0.99999, cos
    assert np.abs(got - ref).max() <= 3e-3, np.abs(got - ref).max()
    bf = Engine(blob, embed_precision="bf16").facenet_embed(torch.from_numpy(x)).cpu().numpy()
    assert np.abs(got - ref).max() < np.abs(bf - ref).max()          # and it IS closer than bf16
    fr = truely_amd.synthetic.synthetic_frames(6, 360, 640, seed=11)
    out, exp = eng.detect_embed(fr), oracle.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid"):
        assert np.array_equal(out[k].cpu().numpy(), exp[k]), k
    v = exp["valid"].astype(bool)
    e = out["emb"].cpu().numpy()
    assert ((e[v] * exp["emb"][v]).sum(1)).min() >= 0.9999
    d = eng.drift_score(out["emb"], out["valid"], len(fr) * 4, 30)
    dref = oracle.drift_score(exp["emb"], exp["valid"], len(fr) * 4, 30)
    assert np.abs(d["sims"].cpu().numpy() - np.asarray(dref["sims"])).max() <= 5e-3


def test_two_contexts_on_two_threads(blob):
    """bench.py keeps two batches in flight: two contexts, two HIP streams, two host threads.  Results must be the
    bits a single context produces (no shared mutable state in the library besides per-context workspaces)."""
    import threading
    from truely_amd.engine import Engine
    sets = [truely_amd.synthetic.synthetic_frames(12, 360, 640, seed=s) for s in (41, 42)]
    single = Engine(blob)
    ref = [{k: v.cpu().numpy() for k, v in single.detect_embed(fr).items()} for fr in sets]
    engs = [Engine(blob), Engine(blob)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got, errs = [None, None], []

    def worker(j):
        try:
            with torch.cuda.stream(streams[j]):
                for _ in range(6):
                    out = engs[j].detect_embed(sets[j])
                streams[j].synchronize()
                got[j] = {k: v.cpu().numpy() for k, v in out.items()}
        except BaseException as e:
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for j in range(2):
        for k in ("box", "prob", "rect", "valid", "emb"):
            assert np.array_equal(got[j][k], ref[j][k]), (j, k)


def test_analyze_video_with_batches_in_flight(blob):
    """model.analyze_video(engines=[e0, e1]): two contexts, two streams, two threads -- same bits as the sequential call."""
    from truely_amd.engine import Engine
    from truely_amd.model import analyze_video
    fr = truely_amd.synthetic.synthetic_frames(40, 360, 640, seed=51)
    seq = analyze_video(fr, fps=30, engine=Engine(blob), batch=8)
    par = analyze_video(fr, fps=30, engines=[Engine(blob), Engine(blob)], batch=8)
    assert seq["score"] == par["score"] and seq["hits"] == par["hits"]
    for k in ("box", "prob", "rect", "valid", "emb", "sims", "flags"):
        assert torch.equal(seq[k].cpu(), par[k].cpu()), k


def test_run_streams_in_bounded_windows(engine, oracle, tmp_path, monkeypatch):
    """model.run holds one window of frames, not the clip: with a 2-sampled-frame window a 40-frame clip takes 5 flushes and the
    score, the annotations' inputs and the output stream equal the single-batch result."""
    from truely_amd import engine as eng_mod, model, video_io
    monkeypatch.setattr(eng_mod, "_default", engine)
    H, W, fps = 180, 320, 30
    fr = truely_amd.synthetic.synthetic_frames(40, H, W, seed=3)
    src = str(tmp_path / "in.trlv")
    video_io.write_raw(src, fr, fps)
    monkeypatch.setenv("TRUELY_ANNOTATE", "0")
    scores = []
    for batch in (32, 2, 3):
        monkeypatch.setattr(model, "BATCH", batch)
        dst = str(tmp_path / f"out{batch}.trlv")
        scores.append(model.run(src, dst))
        rd, _f, _w, _h = video_io.open_reader(dst)
        assert rd.n == 40
        for i in range(40):
            ok, got = rd.read()
            assert ok and np.array_equal(got, fr[i])            # un-annotated output = the decoded frames, in order
    assert scores[0] == scores[1] == scores[2]
    r = oracle.detect_embed(fr[::4])
    assert scores[0] == oracle.drift_score(r["emb"], r["valid"], 40, fps)["score"]


def test_run_nv12_device_ingest(engine, oracle, tmp_path, monkeypatch):
    """A clip whose container yields NV12 (decoder output) is converted on the device (trl_ingest_nv12), sampled there and scored;
    the written frames are the BGR conversion, byte for byte what the oracle's BT.601 restatement gives."""
    from truely_amd import engine as eng_mod, model, video_io
    from truely_amd.ingest import bgr_to_nv12
    monkeypatch.setattr(eng_mod, "_default", engine)
    monkeypatch.setattr(model, "BATCH", 3)
    monkeypatch.setenv("TRUELY_ANNOTATE", "0")
    H, W, fps = 180, 320, 30
    nv = bgr_to_nv12(truely_amd.synthetic.synthetic_frames(22, H, W, seed=3))
    src, dst = str(tmp_path / "in_nv12.trlv"), str(tmp_path / "out.trlv")
    video_io.write_raw(src, nv, fps, pixfmt="nv12", size=(W, H))
    score = model.run(src, dst)
    bgr = np.stack([oracle.nv12_to_bgr(f, H, W) for f in nv])
    r = oracle.detect_embed(bgr[::4])
    assert score == oracle.drift_score(r["emb"], r["valid"], 22, fps)["score"]
    rd, _f, _w, _h = video_io.open_reader(dst)
    assert rd.n == 22 and rd.pixfmt == "bgr"
    for i in range(22):
        ok, got = rd.read()
        assert ok and np.array_equal(got, bgr[i])


def test_run_reads_y4m(engine, tmp_path, monkeypatch):
    """The same clip as YUV4MPEG2 (what `ffmpeg -pix_fmt yuv420p clip.y4m` writes) and as an NV12 TRLV file: same score, same
    written frames -- the planes are repacked to NV12 and take the device ingest."""
    from truely_amd import engine as eng_mod, model, video_io
    from truely_amd.ingest import bgr_to_nv12
    monkeypatch.setattr(eng_mod, "_default", engine)
    monkeypatch.setattr(model, "BATCH", 4)
    monkeypatch.setenv("TRUELY_ANNOTATE", "0")
    H, W, fps = 180, 320, 30
    nv = bgr_to_nv12(truely_amd.synthetic.synthetic_frames(22, H, W, seed=3))
    a, b = str(tmp_path / "in.trlv"), str(tmp_path / "in.y4m")
    video_io.write_raw(a, nv, fps, pixfmt="nv12", size=(W, H))
    video_io.write_y4m(b, nv, fps, (W, H))
    sa, sb = model.run(a, str(tmp_path / "oa.trlv")), model.run(b, str(tmp_path / "ob.trlv"))
    assert sa == sb
    assert open(str(tmp_path / "oa.trlv"), "rb").read() == open(str(tmp_path / "ob.trlv"), "rb").read()


def test_run_writes_annotations(engine, tmp_path, monkeypatch):
    """Sampled frames that were compared with a predecessor carry a box (model.py:67-74); the others are untouched."""
    from truely_amd import engine as eng_mod, model, video_io
    monkeypatch.setattr(eng_mod, "_default", engine)
    H, W, fps = 180, 320, 30
    fr = truely_amd.synthetic.synthetic_frames(24, H, W, seed=3)
    src, dst = str(tmp_path / "in.trlv"), str(tmp_path / "out.trlv")
    video_io.write_raw(src, fr, fps)
    model.run(src, dst)
    out = engine.detect_embed(fr[::4])
    valid = out["valid"].cpu().numpy()
    rd, _f, _w, _h = video_io.open_reader(dst)
    seen_prev, drawn = False, 0
    for i in range(24):
        ok, got = rd.read()
        changed = not np.array_equal(got, fr[i])
        if i % 4 == 0 and valid[i // 4] and seen_prev:
            assert changed; drawn += 1
        else:
            assert not changed
        if i % 4 == 0 and valid[i // 4]:
            seen_prev = True
    assert drawn >= 1


def test_grouped_embedding_is_bit_identical(engine, blob):
    """trl_detect_crop + ONE trl_facenet_embed_masked over several batches == trl_detect_embed per batch, bit for bit (every output
    element is one accumulation chain, whatever tile / launch shape the larger batch selects), also through the pipelined host API."""
    from truely_amd.engine import Engine
    from truely_amd.model import analyze_video
    from truely_amd.pipeline import detect_embed_grouped
    fr = truely_amd.synthetic.synthetic_frames(12, 360, 640, seed=11)
    fr[5] = 127                                              # a faceless frame: its row must be zero in both paths
    ref = engine.detect_embed(fr)
    batches = [fr[0:4], fr[4:8], fr[8:12]]
    for G in (2, 3):
        outs = detect_embed_grouped(engine, batches, G)
        for k in ("box", "prob", "rect", "valid", "emb"):
            assert torch.equal(torch.cat([o[k] for o in outs]), ref[k]), (G, k)
        assert "faces" not in outs[0]
    a = analyze_video(fr, fps=30, engine=engine, batch=4)
    b = analyze_video(fr, fps=30, engines=[engine, Engine(blob)], batch=4, embed_group=2)
    assert a["score"] == b["score"] and torch.equal(a["emb"], b["emb"]) and torch.equal(a["sims"], b["sims"])
    assert not ref["valid"][5] and (ref["emb"][5] == 0).all()


@pytest.mark.parametrize("F,G", [(1, 1), (2, 1), (1, 3), (2, 3), (3, 2), (2, 4)])
def test_overlapped_pipeline_is_bit_identical(engine, blob, F, G):
    """pipeline.detect_embed_overlapped (what bench.py and analyze_video(engines=...) run): ONE host thread, F contexts driven through
    trl_detect_embed_begin / _end, and -- for G > 1 -- the decoupled embedder that embeds every G consecutive batches' crops from
    the ring in one call.  Batch order, the ring's wrap-around, a short last batch and faceless frames included: the bits are
    those of a plain per-batch trl_detect_embed."""
    from truely_amd.engine import Engine
    from truely_amd.pipeline import detect_embed_overlapped
    fr = truely_amd.synthetic.synthetic_frames(46, 180, 320, seed=61)
    fr[7] = 127
    fr[40:44] = 0                                               # a whole faceless batch
    batches = [fr[i:i + 4] for i in range(0, 46, 4)]            # 12 batches, the last one short (2 frames)
    ref = [engine.detect_embed(b) for b in batches]
    engs = [Engine(blob) for _ in range(F)]
    order = []
    outs = detect_embed_overlapped(engs, batches, on_result=lambda i, o: order.append(i), embed_group=G)
    assert order == list(range(len(batches)))
    for i, (o, r) in enumerate(zip(outs, ref)):
        assert "faces" not in o
        for k in ("box", "prob", "rect", "valid", "emb"):
            assert torch.equal(o[k], r[k]), (i, k)
    assert sum(int(r["valid"].sum()) for r in ref) >= 8


def test_begin_end_contract(engine):
    """trl_detect_embed_begin / _end: one call in flight per context, _end without _begin is an error, and a failed _begin leaves
    the context usable."""
    from truely_amd._lib import TrlError
    fr = truely_amd.synthetic.synthetic_frames(3, 180, 320, seed=3)
    ref = engine.detect_embed(fr)
    from truely_amd import _lib
    with pytest.raises(TrlError):
        _lib.check(engine.lib.trl_detect_embed_end(engine._h))  # nothing in flight
    engine.detect_embed_begin(fr)
    with pytest.raises(TrlError):
        engine.detect_embed_begin(fr)                            # second call on a busy context
    out = engine.detect_embed_end()
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert torch.equal(out[k], ref[k]), k
    out2 = engine.detect_embed(fr)                               # the blocking call still works afterwards
    assert torch.equal(out2["emb"], ref["emb"])


def test_large_embedder_batches_cross_kernel_families(engine, oracle):
    """ONE embedder call over more than 335 faces at 80x80 pushes block35's GEMMs past M = 16384 rows: the small-map family
    (fn_conv, 16x16x4 MFMA, grouped launches) hands over to the generic conv_tap kernels (32x32x2 MFMA).  bench.py's grouped
    embedder (768 faces per call) runs exactly that.  The embeddings must be the bits of 256-face calls and of the oracle --
    and again with the small-map family switched off altogether (trl_debug_option "no_fnconv")."""
    fr = truely_amd.synthetic.synthetic_frames(8, 360, 640, seed=11)
    c = engine.detect_crop(fr)
    v = c["valid"].bool()
    assert int(v.sum()) >= 2
    base = c["faces"][v]
    rng = torch.Generator(device="cpu").manual_seed(5)
    reps = (400 + base.shape[0] - 1) // base.shape[0]
    faces = base.repeat(reps, 1, 1, 1)[:400].clone()
    faces += (torch.rand(faces.shape, generator=rng) * (1 / 255.0)).to(faces.device)      # 400 different crops
    valid = torch.ones(400, dtype=torch.uint8)
    valid[[3, 77, 399]] = 0
    big = engine.embed_faces(faces, valid)
    small = torch.cat([engine.embed_faces(faces[i:i + 200], valid[i:i + 200]) for i in (0, 200)])
    assert torch.equal(big, small)
    assert (big[[3, 77, 399]] == 0).all()
    idx = [0, 1, 199, 200, 336, 398]
    ref = oracle.facenet(faces[idx].cpu().numpy())
    assert np.array_equal(big[idx].cpu().numpy(), ref)
    engine.option("no_fnconv", 1)                        # the small-map family switched off: every layer through conv_tap / conv_igemm
    try:
        again = engine.embed_faces(faces, valid)
    finally:
        engine.option("no_fnconv", 0)
    assert torch.equal(again, big)


def test_overlapped_pipeline_recovers_from_a_failed_call(blob):
    """A call fails in the middle of an overlapped run (a batch the library rejects: frames below the 12x12 PNet field): the
    exception propagates, every engine's queued call is finished on the way out, and the same engines work again.  (Crowded
    content is no failure any more: the noisy batch, far beyond the tiny start capacities, simply re-runs with larger lists.)"""
    from truely_amd._lib import TrlError
    from truely_amd.engine import Engine
    from truely_amd.pipeline import detect_embed_overlapped
    engs = [Engine(blob, cap_level=64, cap_frame=64) for _ in range(2)]
    good = truely_amd.synthetic.synthetic_frames(2, 97, 131, seed=1)
    noisy = np.random.default_rng(0).integers(0, 256, (2, 97, 131, 3), dtype=np.uint8)   # far more than 64 candidates per level
    bad = np.zeros((2, 8, 8, 3), np.uint8)
    for G in (1, 2):
        with pytest.raises(TrlError) as ei:
            detect_embed_overlapped(engs, [good, good, bad, good, good], embed_group=G)
        assert ei.value.status == -1
        outs = detect_embed_overlapped(engs, [good, noisy, good], embed_group=G)          # no "call in flight" left behind
        assert len(outs) == 3
        assert max(e.list_stats()["attempts"] for e in engs) >= 1
    ref = Engine(blob).detect_embed(good)
    outs = detect_embed_overlapped([Engine(blob), Engine(blob)], [good, good], embed_group=2)
    assert torch.equal(outs[1]["emb"], ref["emb"])
    ref_n = Engine(blob).detect_embed(noisy)
    outs = detect_embed_overlapped(engs, [good, noisy, good], embed_group=2)
    for k in ("box", "rect", "valid", "emb"):
        assert torch.equal(outs[1][k], ref_n[k]), k


def test_entry_points_refuse_a_context_with_a_call_in_flight(blob):
    """ABI: a context holds ONE queued call; every entry point that would reset or grow its workspaces meanwhile returns
    TRL_ERR_STATE instead of corrupting that call (trl_detect_embed_begin .. _end)."""
    from truely_amd._lib import TrlError
    from truely_amd.engine import Engine
    eng = Engine(blob)
    fr = truely_amd.synthetic.synthetic_frames(2, 97, 131, seed=1)
    ref = eng.detect_embed(fr)
    eng.detect_embed_begin(fr)
    for call in (lambda: eng.detect_embed(fr), lambda: eng.mtcnn_detect(fr), lambda: eng.facenet_embed(torch.zeros(1, 80, 80, 3)),
                 lambda: eng.embed_faces(torch.zeros(1, 80, 80, 3), torch.ones(1, dtype=torch.uint8)), lambda: eng.stage_boxes(1, 0),
                 lambda: eng.poison_workspaces(0xFF), lambda: eng.pyramid_level(fr[0], 0)):
        with pytest.raises(TrlError) as ei:
            call()
        assert ei.value.status == -5
    out = eng.detect_embed_end()
    for k in ("box", "rect", "valid", "emb"):
        assert torch.equal(out[k], ref[k]), k


def test_batches_beyond_16k_frames(blob):
    """check_call admits up to 65,535 frames per call; the exclusive scan of the per-frame candidate counts used to need (n + 1) x 4 B of
    dynamic LDS (a launch failure past 16,383 frames).  17,000 tiny frames in ONE call: the results of the first and last frames
    equal those of a small call on the same frames (batch independence), faceless frames stay faceless."""
    from truely_amd.engine import Engine
    eng = Engine(blob, cap_level=64, cap_frame=64)
    rng = np.random.default_rng(4)
    n = 17000
    fr = (rng.integers(0, 256, (n, 24, 28, 3)) // 8 + 112).astype(np.uint8)
    face = truely_amd.synthetic.synthetic_frames(4, 24, 28, seed=9)
    fr[:4] = face
    fr[-4:] = face
    out = eng.detect_embed(fr)
    small = eng.detect_embed(np.concatenate([fr[:6], fr[-6:]]))
    for k in ("box", "prob", "rect", "valid", "emb"):
        got = torch.cat([out[k][:6], out[k][-6:]])
        assert torch.equal(got, small[k]), k
    assert out["valid"].shape[0] == n


def test_run_decodes_and_writes_mjpeg_avi(engine, oracle, tmp_path, monkeypatch):
    """`run()` on a COMPRESSED clip, both ends, without OpenCV: the input is a Motion-JPEG AVI file (decoded frame by frame through
    `AviMjpegReader` inside run's own loop), the annotated output lands at a path named like the server's (`*_output.mp4`) as a
    Motion-JPEG AVI stream.  The score is the oracle's for the frames the decoder yields; the output holds every frame, at the
    input's rate and size, and reads back (it was written after annotation, so sampled frames with a face differ from the input
    only inside / around the drawn box)."""
    from truely_amd import engine as eng_mod, model, video_io
    if video_io.cv2 is not None:
        pytest.skip("OpenCV present: run() takes the cv2 path for .avi / .mp4")
    monkeypatch.setattr(eng_mod, "_default", engine)
    H, W, fps, N = 180, 320, 30, 36
    fr = truely_amd.synthetic.synthetic_frames(N, H, W, seed=3)
    src, dst = str(tmp_path / "clip.avi"), str(tmp_path / "clip_output.mp4")
    w = video_io.AviMjpegWriter(src, fps, (W, H), quality=92)
    for f in fr:
        w.write(f)
    w.release()
    rd, rfps, rw, rh = video_io.open_reader(src)
    assert isinstance(rd, video_io.AviMjpegReader) and (rfps, rw, rh, rd.n) == (fps, W, H, N)
    decoded = np.stack([rd.read()[1] for _ in range(N)])
    rd.release()
    score = model.run(src, dst)
    r = oracle.detect_embed(decoded[::4])                              # model.py:40,46: every 4th frame at 30 fps
    d = oracle.drift_score(r["emb"], r["valid"], N, fps)
    assert score == d["score"]
    out, ofps, ow, oh = video_io.open_reader(dst)
    assert isinstance(out, video_io.AviMjpegReader) and (ofps, ow, oh, out.n) == (fps, W, H, N)
    ok, first = out.read()
    assert ok and first.shape == (H, W, 3)
    assert os.path.getsize(dst) < decoded.nbytes // 3                   # bounded: a fraction of the raw frames
