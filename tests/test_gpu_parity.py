"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar: bit-exact for boxes / indices / keep masks / rects / valid; embeddings, PNet maps
and drift similarities are ALSO compared bit-for-bit (the kernels reproduce the oracle's fmaf
chains exactly); the north-star tolerance of 1e-4 is asserted as the fallback bound."""
import numpy as np
import pytest

import truely_amd
from conftest import frames_small

pytestmark = pytest.mark.gpu
TOL = 1e-4   # BASELINE.json north_star: embeddings / drift within 1e-4 fp32


def _t(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x))


def test_native_library_loaded(engine):
    """The product path is libtruely_hip.so, not an eager fallback."""
    maps = open("/proc/self/maps").read()
    assert "libtruely_hip.so" in maps


def test_facenet_embed_bit_exact(engine, oracle):
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (5, 80, 80, 3)).astype(np.float32)
    ref = oracle.facenet(x)
    got = engine.facenet_embed(_t(x)).cpu().numpy()
    assert np.abs(got - ref).max() <= TOL
    assert np.array_equal(got, ref), f"max diff {np.abs(got - ref).max()}"


def test_facenet_embed_160(engine, oracle):
    rng = np.random.default_rng(6)
    x = rng.uniform(0, 1, (2, 160, 160, 3)).astype(np.float32)
    ref = oracle.facenet(x)
    got = engine.facenet_embed(_t(x)).cpu().numpy()
    assert np.array_equal(got, ref), f"max diff {np.abs(got - ref).max()}"


def _softmax_p1(oracle, logits):
    """prob[:, 1] of a 2-way softmax exactly as the oracle / device compute it (orc softmax2_p1: shared fmaf-polynomial
    exp, f32 arithmetic, one rounding per op)."""
    out = np.empty(len(logits), np.float32)
    for i, (a0, a1) in enumerate(np.asarray(logits, np.float32)):
        m = max(a0, a1)
        e0 = np.float32(oracle.expf(float(np.float32(a0 - m))))
        e1 = np.float32(oracle.expf(float(np.float32(a1 - m))))
        out[i] = np.float32(e1 / np.float32(e0 + e1))
    return out


def test_rnet_onet_bit_exact(engine, oracle):
    """R-Net / O-Net through the layer kernels on prepared crops: regression, landmarks AND class probabilities."""
    rng = np.random.default_rng(7)
    c24 = rng.uniform(-1, 1, (37, 24, 24, 3)).astype(np.float32)
    p, r = oracle.rnet(c24)
    out = engine.rnet(_t(c24)).cpu().numpy()
    assert np.array_equal(out[:, 2:6], r)
    assert np.array_equal(_softmax_p1(oracle, out[:, :2]), p)
    c48 = rng.uniform(-1, 1, (9, 48, 48, 3)).astype(np.float32)
    p2, r2, pts = oracle.onet(c48)
    out2 = engine.onet(_t(c48)).cpu().numpy()
    assert np.array_equal(out2[:, 2:6], r2)
    assert np.array_equal(out2[:, 6:16], pts)
    assert np.array_equal(_softmax_p1(oracle, out2[:, :2]), p2)


@pytest.mark.parametrize("H,W", [(360, 640), (211, 333)])      # row pitch a multiple of 4 (aligned column sums) / odd (re-aligned dwords)
@pytest.mark.parametrize("net", [24, 48])
def test_front_kernel_on_chosen_boxes(engine, oracle, net, H, W):
    """The PRODUCTION stage-2 / stage-3 path -- k_mtcnn_front (pad, crop, area resample, conv1, PReLU, pool) + the layer tail -- on
    boxes chosen to hit every crop path: tiny (up-sampling bins), medium (the one-load small-box path, bins <= 4x4), large (the
    column-strip path), far larger than the strip, clipped by each frame edge, and degenerate 1-pixel boxes.  Class probability,
    regression and landmarks must equal the oracle's network on the oracle's crops, bit for bit."""
    fr = truely_amd.synthetic.synthetic_frames(1, H, W, seed=11)[0]
    rng = np.random.default_rng(net)
    sx, sy = W / 640.0, H / 360.0
    boxes = [[100.2 * sx, 80.7 * sy, 104.9 * sx, 85.1 * sy], [10, 10, 11, 11], [200.5 * sx, 100.5 * sy, 230.5 * sx, 130.5 * sy],
             [300 * sx, 50 * sy, 371.9 * sx, 121.9 * sy], [50.3 * sx, 40.2 * sy, 250.8 * sx, 240.7 * sy], [0.4, 0.6, W - 0.1, H - 0.1],
             [-30.5, -20.5, 80.5, 90.5], [W - 79.8, H - 109.9, W + 60.9, H + 40.3], [-100, 100 * sy, 20, 220 * sy], [W / 2, -200, W / 2 + 10, 40],
             [1, 1, 3, H - 1], [2, H - 20, W - 2, H - 2], [W - 200.5, H - 150.5, W, H], [W - 97, H - 97, W + 5, H + 5]]
    for _ in range(40):
        side = float(rng.choice([6, 15, 40, 75, 110, 160, 260, 420]))
        x, y = rng.uniform(-40, W - 10), rng.uniform(-40, H - 10)
        boxes.append([x, y, x + side * rng.uniform(0.8, 1.2), y + side * rng.uniform(0.8, 1.2)])
    boxes = np.array(boxes, np.float32)
    tb = np.trunc(boxes).astype(np.int32)                                   # pad(): trunc, clamp to [1, W] x [1, H]
    x, y = np.maximum(tb[:, 0], 1), np.maximum(tb[:, 1], 1)
    ex, ey = np.minimum(tb[:, 2], W), np.minimum(tb[:, 3], H)
    keep = (ey > y - 1) & (ex > x - 1)
    boxes, x, y, ex, ey = boxes[keep], x[keep], y[keep], ex[keep], ey[keep]
    assert len(boxes) >= 40
    crops = np.stack([oracle.area_resample_norm(fr, y[k] - 1, ey[k], x[k] - 1, ex[k], net, net) for k in range(len(boxes))])
    engine.poison_workspaces(0xFF)
    out = engine.front_net(fr, boxes, net).cpu().numpy()
    if net == 24:
        p, r = oracle.rnet(crops)
    else:
        p, r, pts = oracle.onet(crops)
        assert np.array_equal(out[:, 6:16], pts)
    assert np.array_equal(out[:, 2:6], r)
    assert np.array_equal(_softmax_p1(oracle, out[:, :2]), p)


@pytest.mark.parametrize("H,W,seed", [(120, 160, 41), (97, 131, 21)])
def test_fused_pnet_kernel_maps_bit_exact(blob, oracle, H, W, seed):
    """The PRODUCTION PNet kernel's own maps (k_pnet_fused, not the generic layer path `engine.pnet_level` runs): with
    thr0 = 0 every output cell passes `prob >= thr0`, so the candidate records are the full probability and regression
    maps of every pyramid level.  Compared bit for bit with the oracle's PNet on the oracle's pyramid, sub-threshold
    cells included."""
    from truely_amd.engine import Engine
    eng = Engine(blob, thresholds=(0.0, 0.7, 0.7), cap_level=3072, cap_frame=3072)
    assert eng.cfg.pnet_mode == 0
    fr = truely_amd.synthetic.synthetic_frames(2, H, W, seed=seed)
    eng.poison_workspaces(0xFF)
    eng.mtcnn_detect(fr)
    levels = oracle.scales(H, W)
    for f in range(2):
        for l, (sc, h, w) in enumerate(levels):
            lvl = oracle.area_resample_norm(fr[f], 0, H, 0, W, h, w)
            p_ref, r_ref = oracle.pnet_level(lvl)
            rows = eng.level_cands(f, l)
            assert len(rows) == p_ref.size, (f, l, len(rows), p_ref.shape)
            assert np.array_equal(rows["cell"], np.arange(p_ref.size))
            assert np.array_equal(rows["score"], p_ref.reshape(-1)), f"frame {f} level {l}: prob map"
            assert np.array_equal(rows["reg"], r_ref.reshape(-1, 4)), f"frame {f} level {l}: reg map"
            ys, xs = np.divmod(np.arange(p_ref.size), p_ref.shape[1])
            scf = np.float32(sc)
            q1x = np.floor((np.float32(2) * xs.astype(np.float32) + np.float32(1)) / scf)
            q2y = np.floor((np.float32(2) * ys.astype(np.float32) + np.float32(12)) / scf)
            assert np.array_equal(rows["box"][:, 0], q1x) and np.array_equal(rows["box"][:, 3], q2y)


@pytest.mark.parametrize("run", [2, 3, 5, 8, 24])
def test_pnet_halo_carry_is_bit_exact(blob, oracle, run):
    """The fused PNet kernel hands a workgroup RUNS of consecutive tiles; a tile that follows its left neighbour takes the 4 pooled
    and 2 conv2 halo columns the neighbour already computed out of LDS instead of recomputing them (conv1 80 instead of 100
    M-tiles, conv2 18 instead of 21); the tiles of a level are walked in bands of three tile rows, column by column, and a tile below another
    takes 4 pooled and 2 conv2 rows from the tile above the same way.  Large batches run with 24-tile runs; here the run length is
    forced on small frames (levels one, two and seven tile rows high: bands of 3, 2 and 1 rows; run lengths that start on any row of a band): with
    thr0 = 0 every cell of every level's probability / regression map is compared with the oracle, for run lengths that do and do
    not divide the tile rows, then the whole cascade on frames whose levels are many tiles wide (interior + edge tiles)."""
    from truely_amd.engine import Engine
    eng0 = Engine(blob, thresholds=(0.0, 0.7, 0.7), cap_level=3072, cap_frame=3072)
    eng0.pnet_run(run)
    for (H, W, seed) in [(120, 160, 41), (97, 131, 21), (70, 237, 5)]:
        fr = truely_amd.synthetic.synthetic_frames(2, H, W, seed=seed)
        eng0.poison_workspaces(0xFF)
        eng0.mtcnn_detect(fr)
        for f in range(2):
            for l, (sc, h, w) in enumerate(oracle.scales(H, W)):
                p_ref, r_ref = oracle.pnet_level(oracle.area_resample_norm(fr[f], 0, H, 0, W, h, w))
                rows = eng0.level_cands(f, l)
                assert len(rows) == p_ref.size, (H, W, f, l)
                assert np.array_equal(rows["score"], p_ref.reshape(-1)), f"{H}x{W} frame {f} level {l}: prob map"
                assert np.array_equal(rows["reg"], r_ref.reshape(-1, 4)), f"{H}x{W} frame {f} level {l}: reg map"
    eng = Engine(blob)
    eng.pnet_run(run)
    _check_cascade(eng, oracle, truely_amd.synthetic.synthetic_frames(3, 360, 640, seed=11))
    _check_cascade(eng, oracle, truely_amd.synthetic.synthetic_frames(2, 211, 333, seed=12))
    if run in (8, 24):
        _check_cascade(eng, oracle, truely_amd.synthetic.synthetic_frames(2, 720, 1280, seed=0))


@pytest.mark.parametrize("thr", [0.011, 0.3, 0.6, 0.9, 0.989, 0.995])
def test_pnet_threshold_prefilter_drops_nothing(blob, oracle, thr):
    """The fused PNet kernel evaluates the softmax only for an M-tile (32 cells) in which some cell's logit difference comes within
    reach of the first threshold (a bound 0.05 below ln(thr / (1 - thr)); off outside [0.01, 0.99]).  For thresholds across the
    range the candidates of every level are exactly the cells whose oracle probability reaches thr, with the oracle's values."""
    from truely_amd.engine import Engine
    eng = Engine(blob, thresholds=(thr, 0.7, 0.7), cap_level=3072, cap_frame=3072)
    H, W = 120, 160
    fr = truely_amd.synthetic.synthetic_frames(2, H, W, seed=41)
    eng.mtcnn_detect(fr)
    total = 0
    for f in range(2):
        for l, (sc, h, w) in enumerate(oracle.scales(H, W)):
            p_ref, r_ref = oracle.pnet_level(oracle.area_resample_norm(fr[f], 0, H, 0, W, h, w))
            keep = np.flatnonzero(p_ref.reshape(-1) >= np.float32(thr))
            rows = eng.level_cands(f, l)
            assert np.array_equal(rows["cell"], keep), f"thr {thr} frame {f} level {l}: candidate cells"
            assert np.array_equal(rows["score"], p_ref.reshape(-1)[keep])
            assert np.array_equal(rows["reg"], r_ref.reshape(-1, 4)[keep])
            total += len(keep)
    if thr <= 0.6:
        assert total > 0


def test_landmarks_export(engine, oracle):
    """`mtcnn.detect(frame, landmarks=True)`: O-Net's five points, ordered like the boxes (largest area first)."""
    from truely_amd.mtcnn import MTCNN
    m = MTCNN(engine=engine)
    fr = truely_amd.synthetic.synthetic_frames(3, 360, 640, seed=11)
    seen = 0
    for i in range(3):
        boxes, probs, pts = m.detect(fr[i], landmarks=True)
        _b, _p, tr = oracle.detect(fr[i], trace=True)
        if _b is None:
            assert boxes is None and pts is None
            continue
        b3, p3 = tr["boxes3"], tr["points3"]
        area = (b3[:, 2] - b3[:, 0]) * (b3[:, 3] - b3[:, 1])
        order = np.argsort(area, kind="stable")[::-1]            # MTCNN.detect select_largest ordering
        assert np.array_equal(boxes, b3[order, :4]) and np.array_equal(probs, b3[order, 4])
        exp = p3[order].reshape(-1, 2, 5).transpose(0, 2, 1)     # (k, 5, 2) = (x_j, y_j)
        assert pts.shape == exp.shape and np.array_equal(pts, exp)
        seen += len(order)
    assert seen >= 1
    bb, pp = m.detect(fr[0])                                     # the default call is unchanged
    assert np.array_equal(bb, m.detect(fr[0], landmarks=True)[0])


@pytest.mark.parametrize("level", [0, 2, 5])
def test_pnet_level_maps_bit_exact(engine, oracle, level):
    fr = frames_small(1, 180, 320)[0]
    sc, h, w = oracle.scales(180, 320)[level]
    lvl = oracle.area_resample_norm(fr, 0, 180, 0, 320, h, w)
    p_ref, r_ref = oracle.pnet_level(lvl)
    p, r = engine.pnet_level(fr, level)
    p, r = p.cpu().numpy(), r.cpu().numpy()
    assert p.shape == p_ref.shape
    assert np.array_equal(r, r_ref), f"reg max diff {np.abs(r - r_ref).max()}"
    assert np.array_equal(p, p_ref), f"prob max diff {np.abs(p - p_ref).max()}"


def _check_cascade(eng, oracle, frames):
    eng.poison_workspaces(0xFF)   # stale workspaces / LDS hold NaNs: nothing uninitialised may reach a result
    out = eng.detect_embed(frames)
    ref = oracle.detect_embed(frames)
    for i in range(len(frames)):
        _b, _p, tr = oracle.detect(frames[i], trace=True, max_trace=1 << 18)
        cand, keep = eng.level_counts(i)
        assert cand == tr["n_cand_scale"], f"frame {i}: PNet candidate counts"
        assert keep == tr["n_keep_scale"], f"frame {i}: per-scale NMS keep counts"
        for stage in (1, 2, 3):
            got = eng.stage_boxes(stage, i)
            exp = tr[f"boxes{stage}"]
            assert got.shape == exp.shape, f"frame {i} stage {stage}: {got.shape} vs {exp.shape}"
            assert np.array_equal(got, exp), f"frame {i} stage {stage}: max diff {np.abs(got - exp).max()}"
    assert np.array_equal(out["valid"].cpu().numpy(), ref["valid"])
    assert np.array_equal(out["rect"].cpu().numpy(), ref["rect"])
    assert np.array_equal(out["box"].cpu().numpy(), ref["box"])
    assert np.array_equal(out["prob"].cpu().numpy(), ref["prob"])
    emb = out["emb"].cpu().numpy()
    assert np.abs(emb - ref["emb"]).max() <= TOL
    assert np.array_equal(emb, ref["emb"]), f"emb max diff {np.abs(emb - ref['emb']).max()}"
    return out, ref


def test_cascade_generic_pnet_small(engine_generic, oracle):
    _check_cascade(engine_generic, oracle, frames_small(6, 180, 320))


def test_cascade_generic_pnet_360p(engine_generic, oracle):
    out, ref = _check_cascade(engine_generic, oracle, truely_amd.synthetic.synthetic_frames(4, 360, 640, seed=11))
    assert ref["valid"].sum() >= 1


def test_cascade_fused_pnet_small(engine, oracle):
    _check_cascade(engine, oracle, frames_small(6, 180, 320))


def test_cascade_fused_pnet_360p(engine, oracle):
    _check_cascade(engine, oracle, truely_amd.synthetic.synthetic_frames(4, 360, 640, seed=11))


def test_cascade_fused_pnet_odd_sizes(engine, oracle):
    """Level sizes that exercise ceil-mode pooling edges and partial tiles."""
    for (H, W, seed) in [(97, 131, 21), (200, 150, 22), (64, 333, 23)]:
        _check_cascade(engine, oracle, truely_amd.synthetic.synthetic_frames(3, H, W, seed=seed))


def test_frames_at_and_below_the_smallest_pyramid(engine, oracle):
    """min(H, W) * 12 / min_face_size < 12 (short side under 20 px at the default): detect_face() builds NO scale and reports no
    face -- the device path must do the same, not divide by a zero tile count.  At exactly 20 px there is one 13x13 level with
    a 2x2 PNet map; elongated strips give one-tile-high levels.  thr0 = 0 makes every PNet cell a candidate, so the tiny levels'
    records, NMS and stage boxes are compared, not just 'nothing found'."""
    from truely_amd.engine import Engine
    import oracle.oracle as orc_mod
    blob = engine._blob
    eng0 = Engine(blob, thresholds=(0.0, 0.7, 0.7), cap_level=3072, cap_frame=3072)
    orc0 = orc_mod.Oracle(blob)
    orc0.params.thr0 = 0.0
    for (H, W, seed) in [(12, 12, 1), (19, 19, 2), (16, 200, 3), (20, 20, 4), (21, 33, 5), (20, 700, 6), (500, 23, 7)]:
        fr = np.random.default_rng(seed).integers(0, 256, (2, H, W, 3), dtype=np.uint8)
        small = min(H, W) < 20
        assert (engine.levels(H, W) == 0) == small
        for eng, orc in ((engine, oracle), (eng0, orc0)):
            if small:
                got, ref = eng.detect_embed(fr), orc.detect_embed(fr)
                for k in ("box", "prob", "rect", "valid", "emb"):
                    assert np.array_equal(got[k].cpu().numpy(), ref[k]), (H, W, k)
                assert not got["valid"].any() and int(eng.mtcnn_detect(fr)[2].sum()) == 0
            else:
                _check_cascade(eng, orc, fr)


def test_random_shapes_match_oracle(engine, oracle):
    """Seeded sweep over frame shapes (every byte phase of the row pitch, level sizes on both sides of the tile and pooling
    edges, batches of 1-4): the whole cascade record and the embeddings equal the oracle's."""
    rng = np.random.default_rng(20261004)
    for t in range(24):
        H, W, n = int(rng.integers(20, 260)), int(rng.integers(20, 340)), int(rng.integers(1, 5))
        if t % 3 == 0:
            fr = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
            if t % 6 == 0 and H * W > 120 * 160:        # (half of the large noise frames stay raw noise: thousands of candidates)
                fr[:, :, :, :] = (fr // 8 + 112).astype(np.uint8)
        else:
            fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=1000 + t, faces=-1 if t % 3 == 2 else 1)
        try:
            _check_cascade(engine, oracle, fr)
        except AssertionError as e:
            raise AssertionError(f"case {t}: {n} x {H}x{W}: {e}") from e


@pytest.mark.parametrize("tiers", [(16, 64), (128, 256)])
def test_spill_tier_equals_oracle_at_every_stage(blob, tiers):
    """Lists longer than the LDS tier are sorted and suppressed in global memory (the spill tier).  With the tiers lowered to
    64 / 256 candidates and every threshold at 0 -- every PNet cell a candidate, every candidate through R-Net and O-Net -- the
    per-level lists, the per-frame stage-1 lists and the stage-2 / stage-3 lists of small frames all take that tier: counts,
    the boxes after every stage (order included), rectangles and embeddings equal the oracle's."""
    from truely_amd.engine import Engine
    import oracle.oracle as orc_mod
    eng = Engine(blob, thresholds=(0.0, 0.0, 0.0))
    eng.nms_tiers(*tiers)
    orc = orc_mod.Oracle(blob)
    orc.params.thr0 = orc.params.thr1 = orc.params.thr2 = 0.0
    rng = np.random.default_rng(77)
    for (H, W, n) in [(97, 131, 2), (120, 160, 3), (64, 333, 1)]:
        fr = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        fr[0] = truely_amd.synthetic.synthetic_frames(1, H, W, seed=H)[0]
        _check_cascade(eng, orc, fr)
        st = eng.list_stats()
        assert st["spill_lists"] >= 4 * n and st["spill_used"] <= st["spill_cap"], st
        assert st["max_frame_total"] > tiers[1] and st["max_level_count"] > tiers[1], st
        n2 = [len(eng.stage_boxes(2, i)) for i in range(n)]
        assert max(n2) > tiers[1], n2                     # the stage-3 list ('Min' overlap, ties by higher index) spilled as well
    # the same inputs through the default tiers: identical results (the tiers are an implementation detail)
    ref = Engine(blob, thresholds=(0.0, 0.0, 0.0))
    a, b = eng.detect_embed(fr), ref.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()), k


def test_spill_pool_grows_on_demand(blob, oracle):
    """The spill workspace is a bump pool sized up front from the list capacities; when the capacities were grown by an
    earlier, calmer batch the pool can still be too small for a crowded one: the call notices (the cursor keeps counting), grows
    the pool and re-runs."""
    from truely_amd.engine import Engine
    eng = Engine(blob, cap_level=64, cap_frame=64)
    eng.nms_tiers(16, 64)
    calm = truely_amd.synthetic.synthetic_frames(2, 97, 131, seed=1)
    noisy = np.random.default_rng(5).integers(0, 256, (2, 120, 160, 3), dtype=np.uint8)
    _check_cascade(eng, oracle, calm)
    _check_cascade(eng, oracle, noisy)
    assert eng.list_stats()["spill_lists"] > 0
    _check_cascade(eng, oracle, calm)


def _slope_variant_blob(variant):
    from truely_amd import weights
    sds = [dict(sd) for sd in weights.synthetic_state_dicts(0)]
    if variant == "generalise_prelu":                           # what bench.py --prelu general runs
        weights.generalise_prelu(sds)
        return weights.pack_state_dicts(*sds)
    for net, keys in ((sds[0], ("prelu1.weight", "prelu2.weight", "prelu3.weight")), (sds[1], ("prelu1.weight",)),
                      (sds[2], ("prelu1.weight",))):          # PNet (fused kernel) and the R-/O-Net front kernels
        for key in keys:
            w = np.array(net[key], np.float32, copy=True)
            if variant == "slopes_above_one":                   # k_pnet_fused<false, false>, front MODE 1
                w[::2] = 1.25
            elif variant == "negative_slopes":                  # all <= 1, some negative: k_pnet_fused<true, true>, front MODE 0
                w[1::3] = -0.2
            elif variant == "mixed_signs":                      # every class in one layer: k_pnet_fused<false, true>, front MODE 0
                w[0::4] = 1.5; w[1::4] = -0.35; w[2::4] = 0.0; w[3::4] = 1.0
            elif variant == "negative_deep_only":               # conv1 slopes stay in [0, 1]: k_pnet_fused<true, false> with negative conv2/3 slopes
                if key != "prelu1.weight":
                    w[::2] = -0.15
            net[key] = w
    return weights.pack_state_dicts(*sds)


@pytest.mark.parametrize("variant", ["slopes_above_one", "negative_slopes", "mixed_signs", "negative_deep_only", "generalise_prelu"])
def test_cascade_fused_pnet_prelu_variants(variant):
    """A trained checkpoint's PReLU slopes are unconstrained (server/model.py:18 loads them).  Every slope class takes the same
    pool-before-PReLU kernels: slopes above 1 select min(v, s v) through med3, a negative conv1 slope pools the window's min next
    to its max (max_i prelu(v_i) = max(m, s n)).  Same bit-exact bar as the seeded slopes: the whole cascade record, the fused
    kernel's own probability / regression maps (thr0 = 0), and the front kernels on chosen boxes."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.oracle import Oracle
    from truely_amd.engine import Engine
    blob = _slope_variant_blob(variant)
    eng, orc = Engine(blob), Oracle(blob)
    _check_cascade(eng, orc, frames_small(4, 180, 320))
    eng.pnet_run(4)                                              # ... and through the halo-carry path of the same instantiation
    _check_cascade(eng, orc, frames_small(4, 180, 320))
    _check_cascade(eng, orc, truely_amd.synthetic.synthetic_frames(2, 97, 131, seed=21))
    # the fused kernel's own maps, interior and edge tiles, sub-threshold cells included
    eng0 = Engine(blob, thresholds=(0.0, 0.7, 0.7), cap_level=3072, cap_frame=3072)
    eng0.pnet_run(3)
    H, W = 97, 131
    fr = truely_amd.synthetic.synthetic_frames(1, H, W, seed=23)
    eng0.poison_workspaces(0xFF)
    eng0.mtcnn_detect(fr)
    for l, (sc, h, w) in enumerate(orc.scales(H, W)):
        p_ref, r_ref = orc.pnet_level(orc.area_resample_norm(fr[0], 0, H, 0, W, h, w))
        rows = eng0.level_cands(0, l)
        assert len(rows) == p_ref.size
        assert np.array_equal(rows["score"], p_ref.reshape(-1)), f"level {l}: prob map"
        assert np.array_equal(rows["reg"], r_ref.reshape(-1, 4)), f"level {l}: reg map"
    # the front kernels (crop + conv1 + pool + PReLU) on boxes of every crop path
    fr1 = truely_amd.synthetic.synthetic_frames(1, 211, 333, seed=11)[0]
    boxes = np.array([[10, 10, 40, 42], [50.3, 40.2, 150.8, 140.7], [0.4, 0.6, 332.9, 210.9], [-30.5, -20.5, 80.5, 90.5],
                      [200, 100, 206, 107], [120, 60, 330, 209]], np.float32)
    tb = np.trunc(boxes).astype(np.int32)
    x, y = np.maximum(tb[:, 0], 1), np.maximum(tb[:, 1], 1)
    ex, ey = np.minimum(tb[:, 2], 333), np.minimum(tb[:, 3], 211)
    for net in (24, 48):
        crops = np.stack([orc.area_resample_norm(fr1, y[k] - 1, ey[k], x[k] - 1, ex[k], net, net) for k in range(len(boxes))])
        out = eng.front_net(fr1, boxes, net).cpu().numpy()
        ref = orc.rnet(crops) if net == 24 else orc.onet(crops)
        assert np.array_equal(out[:, 2:6], ref[1]), f"net {net}: regression"
        assert np.array_equal(_softmax_p1(orc, out[:, :2]), ref[0]), f"net {net}: probability"
        if net == 48:
            assert np.array_equal(out[:, 6:16], ref[2])


def test_cascade_fused_pnet_720p(engine, oracle):
    out, ref = _check_cascade(engine, oracle, truely_amd.synthetic.synthetic_frames(3, 720, 1280, seed=0))
    assert ref["valid"].sum() >= 1


def test_crop_resize_fixed_point(engine, oracle):
    fr = frames_small(3, 180, 320)
    rects = np.array([[10, 20, 171, 150], [0, 0, 320, 180], [100, 50, 140, 93]], np.int32)   # down, down, up-sampling
    valid = np.array([1, 1, 1], np.uint8)
    got = engine.crop_resize(fr, _t(rects), _t(valid)).cpu().numpy()
    for i, (x0, y0, x1, y1) in enumerate(rects):
        ref = oracle.resize_linear_u8(fr[i], y0, y1, x0, x1).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(got[i], ref)


def test_drift_score_matches_oracle(engine, oracle):
    rng = np.random.default_rng(9)
    n = 200
    base = rng.standard_normal(512).astype(np.float32)
    emb = np.stack([base + rng.standard_normal(512).astype(np.float32) * (0.02 if (i // 25) % 2 == 0 else 0.5) for i in range(n)])
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    ref = oracle.drift_score(emb, valid, n * 4, 30)
    got = engine.drift_score(_t(emb), _t(valid), n * 4, 30)
    assert got["score"] == ref["score"] and got["run"] == ref["run"] and got["hits"] == ref["hits"]
    assert np.array_equal(got["sims"].cpu().numpy(), ref["sims"])
    assert np.array_equal(got["flags"].cpu().numpy(), ref["flags"])


def test_drift_update_carries_the_state_machine_across_windows(engine, oracle):
    """trl_drift_update: model.py's `previous_embedding` / `consecutive_count` / `ai_detected_frames` carried on the device from
    window to window.  Windows of any sizes -- empty ones, single frames, windows that start or end inside a faceless gap, a
    window longer than the scan's LDS chunk -- give the similarities, flags, counters and score of ONE pass over the clip."""
    rng = np.random.default_rng(19)
    n = 6000
    base = rng.standard_normal(512).astype(np.float32)
    noise = np.where((np.arange(n) // 130) % 2 == 0, 0.6, 0.01).astype(np.float32)
    emb = base[None, :] + rng.standard_normal((n, 512)).astype(np.float32) * noise[:, None]
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    valid = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    valid[:3] = 0                                        # the clip starts without a face
    valid[40:75] = 0                                     # a gap that swallows whole windows
    ref = oracle.drift_score(emb, valid, n * 4, 30)
    cuts = [0, 1, 2, 2, 9, 41, 60, 61, 77, 128, 640, 640 + 4500, n]
    state = engine.drift_state()
    sims, flags = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        d = engine.drift_update(state, _t(emb[a:b]), _t(valid[a:b]), b * 4, 30)
        part = oracle.drift_score(emb[:b], valid[:b], b * 4, 30)
        assert (d["score"], d["run"], d["hits"]) == (part["score"], part["run"], part["hits"]), (a, b)
        sims.append(d["sims"].cpu().numpy()); flags.append(d["flags"].cpu().numpy())
    assert np.array_equal(np.concatenate(sims), ref["sims"]) and np.array_equal(np.concatenate(flags), ref["flags"])
    fin = engine.drift_update(state, None, None, n * 4 + 3, 30)          # the final frame count alone changes the score's total
    assert fin["score"] == oracle.drift_score(emb, valid, n * 4 + 3, 30)["score"] and fin["hits"] == ref["hits"]
    whole = engine.drift_score(_t(emb), _t(valid), n * 4, 30)
    assert whole["score"] == ref["score"] and ref["hits"] > 50


def test_drift_score_long_clip_crosses_scan_chunks(engine, oracle):
    """The run-length scan stages similarities through LDS 4096 at a time: a 9,001-frame clip (three chunks, the last one
    ragged) with runs that straddle the chunk borders and faceless stretches must score exactly like the oracle."""
    rng = np.random.default_rng(10)
    n = 9001
    base = rng.standard_normal(512).astype(np.float32)
    noise = np.where((np.arange(n) // 700) % 2 == 0, 0.6, 0.01).astype(np.float32)      # alternating drifting / steady stretches
    emb = base[None, :] + rng.standard_normal((n, 512)).astype(np.float32) * noise[:, None]
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    valid = (rng.uniform(size=n) > 0.05).astype(np.uint8)
    valid[4090:4100] = 0                                                                 # a faceless gap across the first border
    for frame_count, fps in ((n * 4, 30), (n * 3 + 1, 25)):
        ref = oracle.drift_score(emb, valid, frame_count, fps)
        got = engine.drift_score(_t(emb), _t(valid), frame_count, fps)
        assert (got["score"], got["run"], got["hits"]) == (ref["score"], ref["run"], ref["hits"])
        assert np.array_equal(got["sims"].cpu().numpy(), ref["sims"]) and np.array_equal(got["flags"].cpu().numpy(), ref["flags"])
    assert ref["hits"] > 100


@pytest.mark.parametrize("byte", [0xFF, 0x7F])
def test_results_do_not_depend_on_stale_memory(engine, oracle, byte):
    """Workspaces AND the LDS of every CU are filled with NaN (0xFF) / huge-float (0x7F) patterns before the call:
    a kernel that lets an uninitialised word reach an MFMA operand (even against a zero weight: NaN * 0 = NaN)
    or a pooling window shows up here.  Regression test for the O-Net front kernel's k = 27 zero tail."""
    for fr in (frames_small(6, 180, 320), truely_amd.synthetic.synthetic_frames(3, 720, 1280, seed=0)):
        engine.poison_workspaces(byte)
        out = engine.detect_embed(fr)
        ref = oracle.detect_embed(fr)
        for k in ("box", "prob", "rect", "valid", "emb"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), k


@pytest.mark.parametrize("offset", [4, 8, 12])
def test_frame_buffer_at_dword_alignment_only(engine, oracle, offset):
    """The C ABI asks for a 4-byte aligned frame pointer, not 16: a batch living at +4/+8/+12 bytes inside a larger
    device allocation, with 0xFF guard bytes on both sides, must give the same bits (dword-aligned 16-byte loads,
    clamped reads at the end of the buffer)."""
    import torch
    fr = truely_amd.synthetic.synthetic_frames(3, 97, 131, seed=21)        # odd row pitch: frames start at every byte phase
    nbytes = fr.size
    big = torch.full((nbytes + 64,), 0xFF, dtype=torch.uint8, device="cuda")
    view = big[offset:offset + nbytes]
    view.copy_(torch.from_numpy(fr.reshape(-1)))
    dev = view.view(fr.shape)
    assert dev.data_ptr() % 16 == offset % 16 and dev.is_contiguous()
    engine.poison_workspaces(0xFF)
    out = engine.detect_embed(dev)
    ref = oracle.detect_embed(fr)
    for k in ("box", "prob", "rect", "valid", "emb"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    assert bool((big[:offset] == 0xFF).all()) and bool((big[offset + nbytes:] == 0xFF).all())


@pytest.mark.parametrize("H,W", [(180, 320), (97, 131), (720, 1280), (1080, 1920), (2160, 3840)])
def test_pyramid_levels_bit_exact(engine, oracle, H, W):
    """Every level of the production pyramid (what the fused PNet kernel reads) against imresample + normalise of the
    oracle, pixel for pixel.  4K exercises the multiply-high row decode beyond its exact range (fix-up step), table
    driven bin edges and bins too large for the reciprocal division."""
    fr = truely_amd.synthetic.synthetic_frames(1, H, W, seed=5)[0]
    scales = oracle.scales(H, W)
    levels = range(len(scales)) if H <= 720 else (0, 1, 2, 3, len(scales) // 2, len(scales) - 1)
    for level in levels:
        _sc, h, w = scales[level]
        ref = oracle.area_resample_norm(fr, 0, H, 0, W, h, w)
        engine.poison_workspaces(0xFF)
        got = engine.pyramid_level(fr, level).cpu().numpy()
        assert got.shape == ref.shape, (level, got.shape, ref.shape)
        bad = np.argwhere(got != ref)
        assert bad.size == 0, f"level {level} ({h}x{w}): {len(bad)} pixels differ, first at {bad[:3].tolist()}"


@pytest.mark.parametrize("kind", ["noise", "edges", "saturated"])
def test_cascade_stress_inputs(blob, oracle, kind):
    """Inputs chosen to stress the cascade plumbing rather than to look like faces: uniform noise (hundreds of
    candidates per level, boxes hanging over every border -> crop clamps, NMS ties), hard-edged blocks (identical
    scores -> tie-breaks), saturated frames (0 / 255 -> integer sums at their extremes)."""
    from truely_amd.engine import Engine
    rng = np.random.default_rng(77)
    if kind == "noise":
        fr = rng.integers(0, 256, (3, 150, 210, 3), dtype=np.uint8)
    elif kind == "edges":
        fr = np.zeros((3, 144, 192, 3), np.uint8)
        for i in range(3):
            for by in range(0, 144, 24):
                for bx in range(0, 192, 24):
                    fr[i, by:by + 24, bx:bx + 24] = 255 if ((by // 24 + bx // 24 + i) % 2) else 30
    else:
        fr = np.stack([np.full((120, 160, 3), 255, np.uint8), np.zeros((120, 160, 3), np.uint8),
                       np.concatenate([np.full((60, 160, 3), 255, np.uint8), np.zeros((60, 160, 3), np.uint8)])])
    eng = Engine(blob, cap_level=3072, cap_frame=3072)
    _check_cascade(eng, oracle, fr)


def test_batch_capacity_overflow_reruns_the_call(blob, oracle):
    """The R-/O-Net launches are sized by an optimistic capacity kept on the host while the real candidate totals stay on the
    device (no mid-call synchronisation).  When a total exceeds its capacity the call must notice and re-run itself with a larger
    one -- never return results computed from a truncated batch."""
    from truely_amd.engine import Engine
    eng = Engine(blob)
    fr = truely_amd.synthetic.synthetic_frames(4, 360, 640, seed=11)
    ref = oracle.detect_embed(fr)
    eng.batch_capacity(0.25, 0.25)                       # one candidate of head-room for four frames: both stages overflow
    out = eng.detect_embed(fr)
    assert eng.batch_capacity() >= 2                     # the call needed more than one attempt ...
    for k in ("box", "prob", "rect", "valid", "emb"):    # ... and still returned the full result
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    out2 = eng.detect_embed(fr)
    assert eng.batch_capacity() == 1                     # capacities were raised: the next call fits at once
    assert np.array_equal(out2["emb"].cpu().numpy(), ref["emb"])
    eng.batch_capacity(0.25, 1000.0)                     # only the R-Net batch overflows
    b, p, c = eng.mtcnn_detect(fr)
    assert eng.batch_capacity() >= 2
    rb, rp = oracle.detect(fr[0])
    k0 = int(c[0])
    assert (rb is None and k0 == 0) or np.array_equal(b[0, :k0].cpu().numpy(), rb)


def test_multi_chunk_candidate_batches(blob, oracle):
    """R-/O-Net candidate batches larger than one launch set are processed in chunks (device-side `total - chunk_base` clamps).
    With the chunk sizes shrunk to 16 / 16 candidates (trl_debug_option) a 360p clip needs many chunks, some of them partially
    filled, some empty -- results must not change."""
    from truely_amd.engine import Engine
    eng = Engine(blob)
    eng.option("rnet_chunk", 16)
    eng.option("onet_chunk", 16)
    fr = truely_amd.synthetic.synthetic_frames(3, 360, 640, seed=11)
    _check_cascade(eng, oracle, fr)
    assert sum(eng.stage_boxes(1, i).shape[0] for i in range(3)) > 48    # several 16-candidate chunks, the last one ragged
