"""End-to-end wall time of the drop-in `run(video_path_one, video_path_two)` on BASELINE configs[0]'s shape -- the reference's only
timing hook (server/model.py:15,78-80 prints "Total Execution Time" around the whole call: model construction, decode, analysis and
re-encoding of every frame).  The reference's sample clip (test/*.mp4: 640x360, 30 fps, 960 frames, H.264) cannot be decoded without
OpenCV, so the clip is 960 seeded synthetic frames of that shape in the raw TRLV container (BGR and NV12 variants).

    python tools/run_wall_time.py [out.json]

Reported per variant: the FIRST call of the process (engine construction + weight upload + workspace growth, which the reference
pays on every call: model.py:18-19) and a warm call; with the annotated MJPEG/AVI output written and with the output stage skipped."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import truely_amd  # noqa: E402
from truely_amd import model, video_io  # noqa: E402
from truely_amd.ingest import bgr_to_nv12  # noqa: E402


def main():
    H, W, fps, N = 360, 640, 30, 960
    uniq = truely_amd.synthetic.synthetic_frames(48, H, W, seed=21)           # 48 distinct frames, each held for 20 (0.67 s)
    res = {"clip": f"{N} frames {W}x{H} @ {fps} fps (BASELINE configs[0] shape, seeded synthetic frames), {N // 4} analysed",
           "reference_hook": "server/model.py:15,78-80 (Total Execution Time of run())", "runs": []}
    with tempfile.TemporaryDirectory() as td:
        src_bgr, src_nv = os.path.join(td, "c0.trlv"), os.path.join(td, "c0_nv12.trlv")
        wr = video_io.RawWriter(src_bgr, fps, (W, H))
        wn = video_io.RawWriter(src_nv, fps, (W, H), "nv12")
        nv = bgr_to_nv12(uniq)
        for i in range(N):
            wr.write(uniq[i // 20]); wn.write(nv[i // 20])
        wr.release(); wn.release()
        for label, src, env in (("first call: BGR clip, annotated output written", src_bgr, {}),
                                ("warm: BGR clip, annotated output written", src_bgr, {}),
                                ("warm: BGR clip, output stage skipped", src_bgr, {"TRUELY_WRITE_OUTPUT": "0"}),
                                ("warm: NV12 clip (device ingest), annotated output written", src_nv, {}),
                                ("warm: NV12 clip (device ingest), output stage skipped", src_nv, {"TRUELY_WRITE_OUTPUT": "0"})):
            for k in ("TRUELY_WRITE_OUTPUT",):
                os.environ.pop(k, None)
            os.environ.update(env)
            dst = os.path.join(td, "out.avi")
            t0 = time.perf_counter()
            score = model.run(src, dst)
            dt = time.perf_counter() - t0
            size = os.path.getsize(dst) if os.path.exists(dst) and not env else None
            res["runs"].append({"what": label, "seconds": round(dt, 3), "decoded_frames_per_s": round(N / dt, 1),
                                "analysed_frames_per_s": round(N / 4 / dt, 1), "score": int(score), "output_bytes": size})
            if os.path.exists(dst):
                os.remove(dst)
    line = json.dumps(res)
    print(line)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(line + "\n")


if __name__ == "__main__":
    main()
