"""CPU tests of the host side: weight packing, the C-ABI library's symbol table, video containers,
sharding helpers, and model.run's error convention (server/model.py:20-34).  No GPU compute."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

import truely_amd
from truely_amd import _lib, video_io, weights
from truely_amd.distributed import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported_by_library():
    """Every function declared in include/truely_hip.h is exported by libtruely_hip.so (no compute)."""
    hdr = open(os.path.join(ROOT, "include", "truely_hip.h")).read()
    declared = set(re.findall(r"\b(trl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (trl_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported


def test_library_loads_and_reports_abi():
    lib = _lib.load()
    assert lib.trl_abi_version() == 7
    cfg = _lib.TrlConfig()
    assert lib.trl_default_config(ctypes.byref(cfg)) == 0
    assert (cfg.min_face_size, round(cfg.thr0, 3), round(cfg.thr1, 3), round(cfg.thr2, 3), cfg.factor) == (20, 0.6, 0.7, 0.7, 0.709)
    # bad config is rejected with a message, not a crash
    cfg.cap_level = 7
    h = ctypes.c_void_p()
    assert lib.trl_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"trl_config" in lib.trl_last_error()


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from truely_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(b"")


def test_weight_pack_roundtrip_and_layout(state_dicts, blob):
    t = weights.unpack_tensors(blob)
    pn, rn, on, fn = state_dicts
    # conv OIHW -> [ (ky*KW+kx)*Cin + c ][ Cout ]
    w = pn["conv2.weight"]
    k = (1 * 3 + 2) * 10 + 7
    assert t["pnet.conv2.w"].shape == (90, 16) and t["pnet.conv2.w"][k, 5] == w[5, 7, 1, 2]
    # dense4: torch flatten order after permute(0,3,2,1) is (w, h, c); packed order is (h, w, c)
    d = rn["dense4.weight"]
    hh, ww, cc = 2, 1, 33
    assert t["rnet.dense4.w"][(hh * 3 + ww) * 64 + cc, 9] == d[9, (ww * 3 + hh) * 64 + cc]
    # BN fold: alpha = g / sqrt(var + 1e-3), beta = b - mean * alpha
    g, b = fn["conv2d_1a.bn.weight"], fn["conv2d_1a.bn.bias"]
    m, v = fn["conv2d_1a.bn.running_mean"], fn["conv2d_1a.bn.running_var"]
    a = (g * (np.float32(1) / np.sqrt(v + np.float32(1e-3)))).astype(np.float32)
    assert np.array_equal(t["facenet.conv2d_1a.scale"], a)
    assert np.array_equal(t["facenet.conv2d_1a.shift"], (b - m * a).astype(np.float32))
    assert t["facenet.last_linear.w"].shape == (1792, 512)
    assert weights.pack_tensors(t) == blob
    assert len(weights.facenet_basic_convs()) == 111 and len(weights.facenet_proj_convs()) == 21   # 132 convs


def test_synthetic_frames_are_deterministic():
    a = truely_amd.synthetic.synthetic_frames(2, 90, 160, seed=5)
    b = truely_amd.synthetic.synthetic_frames(2, 90, 160, seed=5)
    assert a.dtype == np.uint8 and a.shape == (2, 90, 160, 3) and np.array_equal(a, b)
    assert not np.array_equal(a, truely_amd.synthetic.synthetic_frames(2, 90, 160, seed=6))


def test_raw_video_container_roundtrip(tmp_path):
    fr = truely_amd.synthetic.synthetic_frames(5, 48, 64, seed=1)
    p = str(tmp_path / "clip.trlv")
    video_io.write_raw(p, fr, 29.97)
    rd, fps, w, h = video_io.open_reader(p)
    assert (fps, w, h) == (29, 64, 48)       # int(cap.get(CAP_PROP_FPS)) truncates (model.py:28)
    got = []
    while True:
        ok, f = rd.read()
        if not ok:
            break
        got.append(f)
    assert np.array_equal(np.stack(got), fr)


def test_draw_box_stays_inside_frame():
    img = np.zeros((40, 50, 3), np.uint8)
    video_io.draw_box(img, -5, 3, 60, 38, (0, 255, 0), 2)
    assert img[:, :, 1].max() == 255 and img.shape == (40, 50, 3)


def test_run_error_convention(tmp_path, capsys):
    """Missing / empty / unreadable inputs return 0 and print, like server/model.py:20-26."""
    from truely_amd import model
    assert model.run(str(tmp_path / "nope.mp4"), str(tmp_path / "o.mp4")) == 0
    empty = tmp_path / "empty.mp4"; empty.write_bytes(b"")
    assert model.run(str(empty), str(tmp_path / "o.mp4")) == 0
    junk = tmp_path / "junk.mp4"; junk.write_bytes(b"not a video at all" * 10)
    assert model.run(str(junk), str(tmp_path / "o.mp4")) == 0
    out = capsys.readouterr().out
    assert "doesn't exist or is empty" in out and "couldn't open" in out


def test_shard_bounds_partition():
    for n in (0, 1, 7, 240, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_flop_model_matches_survey():
    import bench
    assert abs(bench.pnet_macs(720, 1280) / 1e6 - 1171.4) < 0.1
    assert abs(bench.pnet_macs(360, 640) / 1e6 - 281.2) < 0.1


def test_committed_bench_line_keeps_the_contract():
    """The newest committed bench line (profiles/roundN_bench.json, written by the driver's command on the GPU box) carries every
    field the measurement contract names, with consistent values: BASELINE's metric and workload, whole-job throughput =
    frames x steps / time, the roofline object of the dominant kernel (achieved = algorithmic FLOPs / measured launch duration,
    frac = achieved / peak) and the CPU baseline of the oracle port on a bounded sample."""
    import glob
    import json
    import bench
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench.json")))
    assert paths
    d = json.loads(open(paths[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("frames/sec") and d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["n_gpus"] == 1
    cfg = d["config"]
    assert "configs[1]" in cfg["workload"] and "model" not in cfg and (cfg["frames_per_gpu"], cfg["height"], cfg["width"]) == (256, 720, 1280)
    assert abs(d["value"] - cfg["frames_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.PEAK_F32_MFMA_TFLOPS
    flops = 2.0 * bench.pnet_macs(720, 1280) * 256
    assert abs(r["flop_per_step"] - flops) / flops < 1e-6
    assert abs(r["achieved"] - flops / (r["kernel_ms_per_step"] / 1e3) / 1e12) < 0.05 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_ms_per_step"] < d["ms_per_step"] and 0.3 < r["frac"] < 1.0 and r["traffic"] > 2.0e9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and 0 < c["value"] < d["value"]


def test_bench_multi_gpu_launch_is_correct_by_construction():
    """The 8-GPU run is the driver's to launch (no node here): what can be checked without hardware.  `--gpus 8` without a
    launcher starts torch.distributed.run with 8 ranks on this node, rendezvous on 127.0.0.1, dmabuf IPC for RCCL, and passes the
    bench flags through; each rank takes LOCAL_RANK as its device -- set before ANY HIP call (engines, process group, tensors) --
    and RCCL refuses to stack ranks on fewer GPUs, while gloo rehearsals may."""
    import bench
    args = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    cmd, env = bench.rank_command(args, ["--gpus", "8", "--steps", "20", "--warmup", "5"], have_gpus=8, environ={"PATH": "/usr/bin"})
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 1024
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/usr/bin"
    with pytest.raises(SystemExit):
        bench.rank_command(args, [], have_gpus=1)                        # RCCL needs one GPU per rank
    assert [bench.rank_device(r, 8, "nccl", 8) for r in range(8)] == list(range(8))
    with pytest.raises(SystemExit):
        bench.rank_device(3, 8, "nccl", 4)
    assert [bench.rank_device(r, 4, "gloo", 1) for r in range(4)] == [0, 0, 0, 0]
    # set_device(LOCAL_RANK) comes before anything that initialises HIP in a rank
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    at = body.index("torch.cuda.set_device(local)")
    for later in ("init_process_group(", "Engine(blob", ".to(dev)", "torch.cuda.Stream(", "pin_memory()"):
        assert at < body.index(later), later
    assert body.index("spawn_ranks(args)") < body.index("import torch")  # the parent never touches the GPU


def test_analysis_service_keeps_event_loop_free():
    """requests queue on one worker, in order, while the event loop keeps running."""
    import asyncio
    import time
    from truely_amd.service import AnalysisService
    order = []

    def fake_run(a, b):
        time.sleep(0.05)
        order.append(a)
        if a == "boom":
            raise RuntimeError("decode failed")
        return len(a)

    svc = AnalysisService(run_fn=fake_run)

    async def main():
        ticks = 0

        async def ticker():
            nonlocal ticks
            for _ in range(20):
                await asyncio.sleep(0.005)
                ticks += 1

        t = asyncio.create_task(ticker())
        res = await asyncio.gather(svc.analyze("a", "o"), svc.analyze("bbb", "o"), svc.analyze("cc", "o"))
        with pytest.raises(RuntimeError):
            await svc.analyze("boom", "o")
        await t
        return res, ticks

    res, ticks = asyncio.run(main())
    svc.close()
    assert res == [1, 3, 2] and order[:3] == ["a", "bbb", "cc"]      # FIFO, one at a time
    assert ticks == 20 and svc.completed == 3                       # the loop was never blocked


def test_rectangle_pixel_fixture():
    """cv2.rectangle(img, (3, 2), (8, 6), c, 2) as restated in annotate.py: thickness 2 covers offsets -1..+1 around each edge
    (3-pixel lines), clipped to the frame.  Hand-derived mask."""
    from truely_amd import annotate
    img = np.zeros((10, 12, 3), np.uint8)
    annotate.rectangle(img, (3, 2), (8, 6), (0, 255, 0), 2)
    exp = np.zeros((10, 12), bool)
    exp[1:8, 2:10] = True            # outer bound: rows 2-1 .. 6+1, columns 3-1 .. 8+1
    exp[4:5, 5:7] = False            # inner hole: rows 2+2 .. 6-2, columns 3+2 .. 8-2
    assert np.array_equal(img[..., 1] == 255, exp) and not img[..., 0].any() and not img[..., 2].any()
    img = np.zeros((6, 6, 3), np.uint8)
    annotate.rectangle(img, (-5, -5), (20, 3), (9, 9, 9), 1)          # thickness 1, mostly outside: only the bottom edge is visible
    assert (img[3] == 9).all() and not img[:3].any() and not img[4:].any()


def test_put_text_hershey_style():
    """Anti-aliased stroke text: ink stays inside the text box cv2.getTextSize would report, sits on the baseline at `org`, scales
    with fontScale and blends (intermediate values exist)."""
    from truely_amd import annotate
    img = np.zeros((60, 420, 3), np.uint8)
    annotate.put_text(img, "AI Detected - Frame 120", (10, 40), 1, (0, 0, 255), 2)
    ink = img[..., 2] > 0
    w, h = annotate.text_size("AI Detected - Frame 120", 1)
    ys, xs = np.nonzero(ink)
    assert xs.min() >= 10 - 2 and xs.max() <= 10 + w + 2 and ys.max() <= 40 + 2 and ys.min() >= 40 - h - 2
    assert ys.max() >= 39                                              # strokes reach the baseline
    assert ((img[..., 2] > 0) & (img[..., 2] < 255)).sum() > 50        # anti-aliased edges
    assert not img[..., 0].any() and not img[..., 1].any()
    small = np.zeros((60, 420, 3), np.uint8)
    annotate.put_text(small, "Real Frame", (10, 40), 0.5, (0, 255, 0), 2)
    ys2, xs2 = np.nonzero(small[..., 1] > 0)
    assert (ys2.max() - ys2.min()) < 0.7 * (ys.max() - ys.min())      # half the scale
    ref = np.zeros((60, 420, 3), np.uint8)
    annotate.put_text(ref, "Real Frame", (10, 40), 0.5, (0, 255, 0), 2)
    assert np.array_equal(small, ref)                                  # deterministic


def test_async_writer_order_skip_and_errors(tmp_path):
    from truely_amd import video_io
    fr = np.random.default_rng(0).integers(0, 255, (7, 16, 20, 3), dtype=np.uint8)
    path = str(tmp_path / "o.trlv")
    w = video_io.AsyncWriter(video_io.RawWriter(path, 30, (20, 16)), annotate=False, depth=2)
    for i in range(7):
        w.put(fr[i].copy(), (i, (2, 2, 10, 10), True))                # notes ignored: annotate=False
    w.close()
    rd, fps, W, H = video_io.open_reader(path)
    assert (rd.n, fps, W, H) == (7, 30, 20, 16)
    for i in range(7):
        assert np.array_equal(rd.read()[1], fr[i])
    w = video_io.AsyncWriter(None)                                     # skipped stage: nothing written, nothing raised
    w.put(fr[0]); w.close()
    assert w.frames == 1

    class Bad:
        def write(self, f):
            raise IOError("disk full")

        def release(self):
            pass
    w = video_io.AsyncWriter(Bad())
    for i in range(5):
        w.put(fr[i])                                                   # must not dead-lock on a failed stage
    with pytest.raises(IOError):
        w.close()


def test_nv12_container_roundtrip(tmp_path):
    from truely_amd import video_io
    from truely_amd.ingest import bgr_to_nv12
    fr = np.random.default_rng(1).integers(0, 255, (3, 8, 12, 3), dtype=np.uint8)
    nv = bgr_to_nv12(fr)
    assert nv.shape == (3, 8 * 12 * 3 // 2)
    path = str(tmp_path / "c.trlv")
    video_io.write_raw(path, nv, 25.0, pixfmt="nv12", size=(12, 8))
    rd, fps, W, H = video_io.open_reader(path)
    assert (rd.pixfmt, fps, W, H, rd.n) == ("nv12", 25, 12, 8, 3)
    for i in range(3):
        assert np.array_equal(rd.read()[1], nv[i])
    gray = np.full((1, 4, 4, 3), 128, np.uint8)                        # BT.601 limited range: mid grey -> Y = 126, U = V = 128
    g = bgr_to_nv12(gray)[0]
    assert (g[:16] == 126).all() and (g[16:] == 128).all()


def _box(typ, payload):
    return struct.pack(">I4s", 8 + len(payload), typ) + payload


def test_mp4_probe_plain_file(tmp_path):
    """A minimal non-fragmented ISO-BMFF file built here: size, timescale -> fps, sample table -> frame count and byte ranges."""
    import struct as st
    from truely_amd import mp4probe
    sps, pps = bytes([0x67, 100, 0, 31, 0xAC]), bytes([0x68, 0xEE, 0x3C, 0x80])
    avcc = _box(b"avcC", bytes([1, 100, 0, 31, 0xFF, 0xE1]) + st.pack(">H", len(sps)) + sps + bytes([1]) + st.pack(">H", len(pps)) + pps)
    entry_body = bytes(6) + st.pack(">H", 1) + bytes(16) + st.pack(">HH", 320, 180) + bytes(50) + avcc
    stsd = _box(b"stsd", st.pack(">II", 0, 1) + st.pack(">I4s", 8 + len(entry_body), b"avc1") + entry_body)
    sizes = [100, 40, 60, 30, 50]
    stts = _box(b"stts", st.pack(">III", 0, 1, 5) + st.pack(">I", 512))
    stsc = _box(b"stsc", st.pack(">II", 0, 1) + st.pack(">III", 1, 5, 1))
    stsz = _box(b"stsz", st.pack(">III", 0, 0, 5) + st.pack(">5I", *sizes))
    stco = _box(b"stco", st.pack(">II", 0, 1) + st.pack(">I", 4096))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    hdlr = _box(b"hdlr", st.pack(">II4s", 0, 0, b"vide") + bytes(13))
    mdhd = _box(b"mdhd", st.pack(">IIIII", 0, 0, 0, 12800, 2560) + bytes(4))
    mdia = _box(b"mdia", mdhd + hdlr + _box(b"minf", stbl))
    tkhd = _box(b"tkhd", st.pack(">IIII", 0, 0, 0, 7) + bytes(68))
    moov = _box(b"moov", _box(b"trak", tkhd + mdia))
    path = tmp_path / "clip.mp4"
    path.write_bytes(_box(b"ftyp", b"isom" + bytes(4)) + moov)
    i = mp4probe.probe(str(path))
    assert (i.width, i.height, i.frame_count, i.codec, i.profile_idc, i.nal_length_size) == (320, 180, 5, "avc1", 100, 4)
    assert abs(i.fps - 25.0) < 1e-9 and i.sps == [sps] and i.pps == [pps] and not i.fragmented
    assert i.sample_ranges == [(4096, 100), (4196, 40), (4236, 60), (4296, 30), (4326, 50)]
    assert "H.264 High@L3.1 320x180, 25.000 fps, 5 frames" in video_io.describe(str(path))
    assert mp4probe.probe(__file__) is None and video_io.describe(__file__) == "unknown container"


def test_mp4_probe_reference_sample():
    """The reference's own sample clip (BASELINE configs[0]; fragmented mp4 from yt-dlp): the container facts SURVEY section 6
    lists -- 640x360, 30 fps, 960 frames -- read by this build's demultiplexer.  (Decoding it needs OpenCV: DESIGN section 8.)"""
    import glob
    from truely_amd import mp4probe
    files = glob.glob("/root/reference/test/*.mp4")
    if not files:
        pytest.skip("the reference checkout is not present on this box")
    i = mp4probe.probe(files[0])
    assert (i.width, i.height, i.frame_count, i.fragmented, i.codec) == (640, 360, 960, True, "avc1")
    assert abs(i.fps - 30.0) < 1e-9 and i.profile_idc == 77 and len(i.sps) == 1 and len(i.pps) == 1
    n, idr = 0, 0
    for au in mp4probe.samples(files[0], i):
        q = 0
        while q + 4 <= len(au):
            ln = int.from_bytes(au[q:q + 4], "big")
            idr += (au[q + 4] & 31) == 5
            q += 4 + ln
        assert q == len(au)                                   # every access unit is a whole number of NAL units
        n += 1
    assert n == 960 and idr >= 1
    assert max(1, int(int(i.fps) / 7)) == 4                   # model.py:40 -> 240 sampled frames


def test_y4m_reader_repacks_planes_to_nv12(tmp_path):
    """YUV4MPEG2 4:2:0 in, flat NV12 out (what the device ingest consumes); header variants; unsupported streams are 'cannot open'."""
    rng = np.random.default_rng(3)
    H, W, n = 6, 8, 5
    nv = rng.integers(0, 256, (n, H * W * 3 // 2), dtype=np.uint8)
    p = str(tmp_path / "a.y4m")
    video_io.write_y4m(p, nv, 25, (W, H))
    rd, fps, w, h = video_io.open_reader(p)
    assert (fps, w, h, rd.n, rd.pixfmt) == (25, W, H, n, "nv12")
    for i in range(n):
        ok, fr = rd.read()
        assert ok and np.array_equal(fr, nv[i])
    assert rd.read()[0] is False
    rd.release()
    # planar layout on disk: Y, then U, then V
    raw = open(p, "rb").read()
    body = raw[raw.index(b"\n") + 1:]
    assert body[:6] == b"FRAME\n" and body[6:6 + H * W] == nv[0][:H * W].tobytes()
    assert body[6 + H * W:6 + H * W + (H // 2) * (W // 2)] == nv[0][H * W::2].tobytes()
    # NTSC-style rational rate, frame parameters, no C tag (defaults to 4:2:0)
    q = str(tmp_path / "b.y4m")
    with open(q, "wb") as f:
        f.write(b"YUV4MPEG2 W8 H6 F30000:1001 Ip A1:1\n")
        f.write(b"FRAME\n" + bytes(H * W * 3 // 2))
    rd, fps, w, h = video_io.open_reader(q)
    assert fps == 29 and rd.read()[0]                     # int(fps) like model.py:28
    for bad in (b"YUV4MPEG2 W8 H6 F30:1 Ip C444\n", b"YUV4MPEG2 W8 H6 F30:1 It C420jpeg\n", b"YUV4MPEG2 W6 H6 F30:1 Ip C420jpeg\n",
                b"YUV4MPEG2 W8 H6 F30:1 Ip C420p10\n"):
        r = str(tmp_path / "bad.y4m")
        open(r, "wb").write(bad + b"FRAME\n" + bytes(100))
        assert video_io.open_reader(r) is None


def _mp4_with(stsz_payload=None, stco_payload=None, stsc_payload=None, moof=None, trex=True):
    """A minimal H.264 mp4 whose sample tables (or fragments) are supplied by the caller: the probe's hostile-input tests."""
    import struct as st
    sps, pps = bytes([0x67, 100, 0, 31, 0xAC]), bytes([0x68, 0xEE, 0x3C, 0x80])
    avcc = _box(b"avcC", bytes([1, 100, 0, 31, 0xFF, 0xE1]) + st.pack(">H", len(sps)) + sps + bytes([1]) + st.pack(">H", len(pps)) + pps)
    entry_body = bytes(6) + st.pack(">H", 1) + bytes(16) + st.pack(">HH", 320, 180) + bytes(50) + avcc
    stsd = _box(b"stsd", st.pack(">II", 0, 1) + st.pack(">I4s", 8 + len(entry_body), b"avc1") + entry_body)
    stsc = _box(b"stsc", stsc_payload if stsc_payload is not None else st.pack(">II", 0, 1) + st.pack(">III", 1, 5, 1))
    stsz = _box(b"stsz", stsz_payload if stsz_payload is not None else st.pack(">III", 0, 0, 0))
    stco = _box(b"stco", stco_payload if stco_payload is not None else st.pack(">II", 0, 1) + st.pack(">I", 4096))
    stbl = _box(b"stbl", stsd + stsc + stsz + stco)
    hdlr = _box(b"hdlr", st.pack(">II4s", 0, 0, b"vide") + bytes(13))
    mdhd = _box(b"mdhd", st.pack(">IIIII", 0, 0, 0, 12800, 2560) + bytes(4))
    mdia = _box(b"mdia", mdhd + hdlr + _box(b"minf", stbl))
    tkhd = _box(b"tkhd", st.pack(">IIII", 0, 0, 0, 7) + bytes(68))
    mvex = _box(b"mvex", _box(b"trex", st.pack(">IIIIII", 0, 7, 1, 512, 0, 0))) if trex else b""
    moov = _box(b"moov", _box(b"trak", tkhd + mdia) + mvex)
    return _box(b"ftyp", b"isom" + bytes(4)) + moov + (moof or b"")


def test_mp4_probe_rejects_hostile_counts(tmp_path):
    """The 32-bit counts of stsz / stco / stsc / trun come straight from an upload (model.run describes every clip the decoder
    rejects: video_io.describe -> mp4probe.probe).  A count the box cannot hold must be refused at once -- not turned into a
    34 GB list or a 4e9-iteration loop inside the single-worker analysis service."""
    import struct as st
    import time
    from truely_amd import mp4probe
    big = 0xFFFFFFFF
    traf = lambda trun: _box(b"moof", _box(b"traf", _box(b"tfhd", st.pack(">II", 0x020000, 7)) + trun))   # noqa: E731
    cases = {
        "stsz_uniform": _mp4_with(stsz_payload=st.pack(">III", 0, 1000, big)),                  # uniform size, 4e9 samples
        "stsz_table": _mp4_with(stsz_payload=st.pack(">III", 0, 0, big) + bytes(16)),           # table count past the box
        "stco": _mp4_with(stsz_payload=st.pack(">III", 0, 0, 2) + st.pack(">2I", 10, 10), stco_payload=st.pack(">II", 0, big)),
        "stsc": _mp4_with(stsz_payload=st.pack(">III", 0, 0, 2) + st.pack(">2I", 10, 10), stsc_payload=st.pack(">II", 0, big)),
        "trun_no_fields": _mp4_with(moof=traf(_box(b"trun", st.pack(">II", 0, big)))),           # flags = 0: nothing per sample
        "trun_sizes": _mp4_with(moof=traf(_box(b"trun", st.pack(">II", 0x200, big) + bytes(64)))),
        "trun_default_size": _mp4_with(moof=_box(b"moof", _box(b"traf", _box(b"tfhd", st.pack(">III", 0x020010, 7, 4096)) +
                                                          _box(b"trun", st.pack(">II", 0, 9_000_000))))),
    }
    for name, blob in cases.items():
        p = tmp_path / f"{name}.mp4"
        p.write_bytes(blob)
        t0 = time.time()
        with pytest.raises(mp4probe.Mp4Error):
            mp4probe.probe(str(p))
        assert video_io.describe(str(p)) == "unknown container", name       # what model.run prints for it
        assert time.time() - t0 < 1.0, name
    # sane neighbours still parse: a uniform-size table, and a fragment run that relies on the default sample size
    ok = tmp_path / "uniform.mp4"
    ok.write_bytes(_mp4_with(stsz_payload=st.pack(">III", 0, 100, 5)) + bytes(5000))
    i = mp4probe.probe(str(ok))
    assert i.frame_count == 5 and i.sample_ranges == [(4096 + 100 * k, 100) for k in range(5)]
    frag = tmp_path / "frag.mp4"
    frag.write_bytes(_mp4_with(moof=_box(b"moof", _box(b"traf", _box(b"tfhd", st.pack(">III", 0x020010, 7, 64)) +
                                                   _box(b"trun", st.pack(">II", 0, 3))))) + bytes(256))
    j = mp4probe.probe(str(frag))
    assert j.fragmented and j.frame_count == 3 and [s for _, s in j.sample_ranges] == [64, 64, 64]


def test_mjpeg_avi_sink_and_source(tmp_path):
    """Without OpenCV the annotated output is Motion-JPEG in an AVI container (bounded size, a standard file), whatever the
    output path is called; the same container is the one compressed INPUT this build decodes itself.  RIFF structure, header
    fields, index, lossy-but-close frames, odd-sized chunks padded, and the `.trlv` escape hatch of the tests."""
    import struct as st
    fr = truely_amd.synthetic.synthetic_frames(6, 90, 160, seed=4)
    p = str(tmp_path / "video_output.mp4")                             # the name server.py gives it (H.264 needs OpenCV)
    w = video_io.open_writer(p, 30, (160, 90))
    assert isinstance(w, video_io.AviMjpegWriter)
    for f in fr:
        w.write(f)
    w.release()
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and st.unpack_from("<I", raw, 4)[0] == len(raw) - 8
    assert len(raw) < fr.nbytes // 4                                   # bounded: far below the raw frames
    movi = raw.index(b"movi")
    assert raw[movi - 8:movi - 4] == b"LIST"
    (movi_bytes,) = st.unpack_from("<I", raw, movi - 4)
    idx = movi + movi_bytes - 4 + 4                                    # list payload starts at 'movi'
    assert raw[idx:idx + 4] == b"idx1" and st.unpack_from("<I", raw, idx + 4)[0] == 16 * 6
    for k in range(6):                                                  # every index entry points at a JPEG (SOI marker)
        cid, _fl, off, size = st.unpack_from("<4sIII", raw, idx + 8 + 16 * k)
        assert cid == b"00dc" and raw[movi + off:movi + off + 4] == b"00dc" and raw[movi + off + 8:movi + off + 10] == b"\xff\xd8"
    rd, fps, W, H = video_io.open_reader(p)
    assert isinstance(rd, video_io.AviMjpegReader) and (fps, W, H, rd.n) == (30, 160, 90, 6)
    for k in range(6):
        ok, g = rd.read()
        assert ok and g.shape == (90, 160, 3)
        mse = ((g.astype(np.float64) - fr[k]) ** 2).mean()
        assert 10 * np.log10(255.0 ** 2 / mse) > 30.0                  # BGR order kept (a swapped channel would be ~10 dB)
    assert rd.read() == (False, None)
    rd.release()
    assert isinstance(video_io.open_writer(str(tmp_path / "o.trlv"), 30, (160, 90)), video_io.RawWriter)
    bad = tmp_path / "bad.avi"
    bad.write_bytes(raw[:200])                                          # truncated: no frames -> "cannot open"
    assert video_io.open_reader(str(bad)) is None or video_io.open_reader(str(bad))[0].n == 0


def test_run_cleans_up_when_analysis_fails(tmp_path, monkeypatch):
    """An exception inside run()'s loop (an allocation failure, a HIP error, a damaged clip ...) must not leave the reader or the
    writer thread blocked on its queue or the files open: the long-lived service would leak them per failed request.  The real
    run() loop and reader thread, with the device pieces (contexts, streams, pinned ring) replaced by host stand-ins."""
    import threading
    from truely_amd import engine as eng_mod, model, pipeline

    class Boom(RuntimeError):
        pass

    class FakeOverlapped:
        def __init__(self, engines, **kw):
            self.pushed = 0

        def push(self, batch):
            self.pushed += 1
            if self.pushed == 2:
                raise Boom("libtruely_hip status -2: out of memory")

        def finish(self):
            return []

        def abandon(self):
            pass

    class FakeCtx:
        def __init__(self, eng):
            self.engines, self.streams, self.lock = [eng, eng], [None, None], threading.Lock()

        def buffers(self, rows, row_bytes, yuv, H, W):
            bufs = [np.zeros((rows, row_bytes), np.uint8) for _ in range(4)]
            return bufs, bufs, [np.zeros((rows, row_bytes), np.uint8) for _ in range(2)]

    class FakeEngine:
        device = None

        def drift_state(self):
            return None

    fr = np.random.default_rng(2).integers(0, 255, (200, 24, 32, 3), dtype=np.uint8)
    src, dst = str(tmp_path / "in.trlv"), str(tmp_path / "out.trlv")
    video_io.write_raw(src, fr, 7.0)                                    # fps 7 -> every frame sampled: 32 per window
    released = []
    real_open = video_io.open_reader

    def spy_open(path):
        r = real_open(path)
        rel = r[0].release
        r[0].release = lambda: (released.append(path), rel())
        return r

    fake = FakeEngine()
    fake._run_ctx = FakeCtx(fake)
    monkeypatch.setattr(video_io, "open_reader", spy_open)
    monkeypatch.setattr(eng_mod, "_default", fake)
    monkeypatch.setattr(model, "default_engine", lambda: fake)
    monkeypatch.setattr(pipeline, "Overlapped", FakeOverlapped)
    before = {t.ident for t in threading.enumerate()}
    with pytest.raises(Boom):
        model.run(src, dst)
    assert released == [src]                                            # the reader was closed
    left = [t for t in threading.enumerate() if t.ident not in before and t.name in ("truely-writer", "truely-reader") and t.is_alive()]
    assert not left, f"threads survived the failed request: {left}"
    rd, _f, _w, _h = real_open(dst)                                     # the sink was closed properly (header patched)
    assert rd.n >= 0
    rd.release()
    # ... and a clip that ends in the middle of a frame is analysed up to its last whole frame, not an error
    with open(src, "r+b") as f:
        f.truncate(32 + 24 * 32 * 3 * 50 + 100)
    monkeypatch.setattr(pipeline, "Overlapped", lambda engines, **kw: type("Ok", (), {"push": lambda self, b: None, "finish": lambda self: [], "abandon": lambda self: None})())
    fake.drift_update = lambda *a, **k: {"score": 0}
    assert model.run(src, dst) == 0 and released == [src, src]


def test_opencv_branches_with_a_recording_stand_in(tmp_path, monkeypatch):
    """The branches taken when `cv2` IS importable (it is not in this image, so nothing else executes them): a stand-in object
    that records its calls is put in cv2's place.  The reader hands back the capture with the reference's int() truncation of
    the frame rate (server/model.py:28: 29.97 -> 29), the writer is opened as model.py:35-36 opens it (fourcc "H264", fps,
    (w, h)), the annotations are model.py:67-74's calls argument for argument, and run()'s reader thread walks a capture object
    through cap.read() (the sequential path every compressed source takes)."""
    from truely_amd import annotate, model
    calls = []
    frames = np.random.default_rng(0).integers(0, 256, (9, 24, 32, 3), dtype=np.uint8)

    class Cap:
        def __init__(self, path):
            calls.append(("VideoCapture", path)); self.i = 0; self.open = path.endswith(".mp4")
        def isOpened(self):
            return self.open
        def get(self, prop):
            return {5: 29.97, 3: 32.0, 4: 24.0}[prop]
        def read(self):
            if self.i >= len(frames):
                return False, None
            self.i += 1
            return True, frames[self.i - 1].copy()
        def release(self):
            calls.append(("release",))

    class FakeCv2:
        CAP_PROP_FPS, CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT = 5, 3, 4
        FONT_HERSHEY_SIMPLEX, LINE_AA = 0, 16
        VideoCapture = Cap
        @staticmethod
        def VideoWriter_fourcc(*c):
            return "".join(c)
        @staticmethod
        def VideoWriter(path, fourcc, fps, size):
            calls.append(("VideoWriter", path, fourcc, fps, size)); return "writer"
        @staticmethod
        def rectangle(frame, p1, p2, color, thickness):
            calls.append(("rectangle", p1, p2, color, thickness))
        @staticmethod
        def putText(frame, text, org, font, scale, color, thickness, line):
            calls.append(("putText", text, org, font, scale, color, thickness, line))

    monkeypatch.setattr(video_io, "cv2", FakeCv2)
    monkeypatch.setattr(annotate, "cv2", FakeCv2)
    clip = tmp_path / "clip.mp4"; clip.write_bytes(b"\x00\x00\x00\x18ftypmp42 not really")
    cap, fps, w, h = video_io.open_reader(str(clip))
    assert (fps, w, h) == (29, 32, 24) and isinstance(cap, Cap)               # int(29.97) = 29 -> step 4 (model.py:28,40)
    bad = tmp_path / "clip.mov"; bad.write_bytes(b"junk junk junk")
    assert video_io.open_reader(str(bad)) is None                               # cap.isOpened() false -> "couldn't open" (model.py:24-26)
    assert video_io.open_writer(str(tmp_path / "out.mp4"), 29, (32, 24)) == "writer"
    assert ("VideoWriter", str(tmp_path / "out.mp4"), "H264", 29, (32, 24)) in calls
    calls.clear()
    img = np.zeros((24, 32, 3), np.uint8)
    annotate.annotate(img, 12, (3, 4, 20, 21), True)
    annotate.annotate(img, 16, (3, 4, 20, 21), False)
    assert calls == [("rectangle", (3, 4), (20, 21), (0, 0, 255), 2),
                     ("putText", "AI Detected - Frame 12", (10, 30), 0, 1, (0, 0, 255), 2, 16),
                     ("rectangle", (3, 4), (20, 21), (0, 255, 0), 2),
                     ("putText", "Real Frame", (3, -6), 0, 0.5, (0, 255, 0), 2, 16)]
    assert not img.any()                                                       # OpenCV draws when it is there, not this module
    # run()'s reader thread over such a capture: sampled frames into the slot, every frame kept for the writer
    slots = [np.zeros((2, 24 * 32 * 3), np.uint8) for _ in range(2)]
    rd = model._WindowReader(cap, 4, 2, slots, False, True, (24, 32, 3))
    assert not rd.random
    for k in range(2):
        rd.free.put((k, None))
    rd.start()
    got = []
    while True:
        item = rd.full.get(timeout=30)
        if item is None:
            break
        assert not isinstance(item, BaseException), item
        slot, nrows, first, nfr, host = item
        got.append((nrows, first, nfr, slots[slot][:nrows].copy(), host))
        rd.free.put((slot, None))
    rd.join(10)
    assert [(g[0], g[1], g[2]) for g in got] == [(2, 0, 8), (1, 8, 1)] and rd.frame_count == 9
    assert np.array_equal(got[0][3][1], frames[4].reshape(-1)) and np.array_equal(got[1][3][0], frames[8].reshape(-1))
    assert all(np.array_equal(got[0][4][k], frames[k]) for k in range(8))


def test_window_reader_reads_what_is_needed(tmp_path):
    """model._WindowReader (run()'s reader thread) on the containers with fixed-size frames: with the output stage skipped only the
    SAMPLED frames are read (positioned reads into the slot), with a 4:2:0 source and the output on every frame lands in the slot,
    with a BGR source and the output on every frame becomes a host array and the sampled ones are copied to the slot; windows are
    ragged at the end of the clip; slots are handed back and reused; YUV4MPEG2 frames arrive planar (I420), as stored."""
    from truely_amd import model
    from truely_amd.ingest import bgr_to_nv12
    H, W, step, win = 24, 32, 4, 3
    fr = np.random.default_rng(4).integers(0, 256, (29, H, W, 3), dtype=np.uint8)      # 29 frames: windows of 12, 12, 5
    nv = bgr_to_nv12(fr)
    a, b, c = str(tmp_path / "a.trlv"), str(tmp_path / "b.trlv"), str(tmp_path / "c.y4m")
    video_io.write_raw(a, fr, 30.0)
    video_io.write_raw(b, nv, 30.0, pixfmt="nv12", size=(W, H))
    video_io.write_y4m(c, nv, 30, (W, H))

    def drain(path, all_rows, keep_host, rows):
        cap = video_io.open_reader(path)[0]
        slots = [np.full((rows, cap.frame_bytes), 0xEE, np.uint8) for _ in range(2)]
        rd = model._WindowReader(cap, step, win, slots, all_rows, keep_host, (H, W, 3))
        assert rd.random
        for k in range(2):
            rd.free.put((k, None))
        rd.start()
        got = []
        while True:
            item = rd.full.get(timeout=30)
            if item is None:
                break
            assert not isinstance(item, BaseException), item
            slot, nrows, first, nfr, host = item
            got.append((nrows, first, nfr, slots[slot][:nrows].copy(), host))
            rd.free.put((slot, None))                 # two slots, three windows: the first one is reused
        rd.join(10)
        cap.release()
        assert rd.frame_count == 29 and [(g[1], g[2]) for g in got] == [(0, 12), (12, 12), (24, 5)]
        return got

    got = drain(a, False, False, win)                  # BGR, output skipped: sampled frames only
    assert [g[0] for g in got] == [3, 3, 2]
    for nrows, first, nfr, rows, host in got:
        assert host is None
        for r in range(nrows):
            assert np.array_equal(rows[r], fr[first + r * step].reshape(-1))
    got = drain(a, False, True, win)                   # BGR, output written: every frame on the host, sampled ones in the slot
    for nrows, first, nfr, rows, host in got:
        assert len(host) == nfr and all(np.array_equal(host[k], fr[first + k]) for k in range(nfr))
        assert all(np.array_equal(rows[r], fr[first + r * step].reshape(-1)) for r in range(nrows))
    got = drain(b, True, False, win * step)            # NV12, output written: every frame in the slot (converted on the device later)
    assert [g[0] for g in got] == [12, 12, 5]
    assert all(np.array_equal(g[3][k], nv[g[1] + k]) for g in got for k in range(g[0]))
    got = drain(c, False, False, win)                  # YUV4MPEG2: planar frames as stored (no repacking on the host)
    ys = H * W
    for nrows, first, nfr, rows, host in got:
        for r in range(nrows):
            f = nv[first + r * step]
            assert np.array_equal(rows[r][:ys], f[:ys]) and np.array_equal(rows[r][ys:ys + ys // 4], f[ys::2]) and np.array_equal(rows[r][ys + ys // 4:], f[ys + 1::2])
    # a consumer that goes away: the reader ends instead of blocking on its queue
    cap = video_io.open_reader(a)[0]
    rd = model._WindowReader(cap, step, win, [np.zeros((win, cap.frame_bytes), np.uint8)], False, False, (H, W, 3))
    rd.free.put((0, None))
    rd.start()
    assert rd.full.get(timeout=30)[1] == 3
    rd.stop = True
    rd.free.put(None)
    rd.join(10)
    assert not rd.is_alive()
    cap.release()


def test_analysis_service_spreads_requests_over_gpus():
    """f3 widened: AnalysisService(gpus=[...]) keeps one worker (and engine) per GPU and hands each request to the least-loaded
    one; a GPU's requests stay in arrival order; failures free their slot."""
    import asyncio
    import threading
    import time
    from truely_amd.service import AnalysisService
    seen, lock = [], threading.Lock()
    gate = threading.Event()

    def fake_run(a, b, device=None):
        if a.startswith("slow"):
            gate.wait(5)
        with lock:
            seen.append((a, device))
        if a == "boom":
            raise RuntimeError("bad clip")
        return 10 * device + len(a)

    svc = AnalysisService(run_fn=fake_run, gpus=[0, 1, 2])
    f1 = svc.submit("slow1", "o")                                       # GPU 0 (all idle: lowest ordinal)
    f2 = svc.submit("slow22", "o")                                      # GPU 1
    time.sleep(0.05)
    assert [l for _g, l, _c in svc.loads()] == [1, 1, 0]
    q0 = svc.submit("q0", "o")                                          # the idle GPU 2
    assert q0.result(5) == 22                                           # ... finished while 0 and 1 are still busy
    q1 = svc.submit("q1", "o")                                          # GPU 2 again: the only idle one
    assert q1.result(5) == 22
    q2, q3 = svc.submit("q2", "o"), svc.submit("q3", "o")               # q2 -> GPU 2 (load 0); q3 -> GPU 2 if q2 is done, else GPU 0
    quick = [q0, q1, q2, q3]
    assert q2.result(5) == 22
    gate.set()
    assert f1.result(5) == 5 and f2.result(5) == 16 and q3.result(5) in (2, 22)
    with pytest.raises(RuntimeError):
        svc.submit("boom", "o").result(5)
    assert [l for _g, l, _c in svc.loads()] == [0, 0, 0] and (svc.completed, svc.failed) == (6, 1)

    async def main():
        return await asyncio.gather(*[svc.analyze(f"r{i}", "o") for i in range(6)])

    res = asyncio.run(main())
    svc.close()
    assert sorted(r // 10 for r in res) == [0, 0, 1, 1, 2, 2]            # two each: the six requests were spread evenly
    order0 = [a for a, d in seen if d == 0 and a.startswith("r")]
    assert order0 == sorted(order0)                                     # arrival order within a GPU
    with pytest.raises(ValueError):
        AnalysisService(run_fn=fake_run, gpus=[])
