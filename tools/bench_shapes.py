"""Supplementary timings at BASELINE.json's other frame shapes (fp32 path, synthetic frames); NOT the headline metric.
python tools/bench_shapes.py > gpurun_out/shapes.json"""
import sys; sys.path.insert(0, "/root/repo")
import json, statistics, time
import torch, truely_amd
from truely_amd.engine import Engine

res = []
for (name, n, H, W, faces) in [("360p clip frames (configs[0] shape)", 256, 360, 640, 1), ("720p 1-face (configs[1], headline)", 256, 720, 1280, 1),
                               ("1080p multi-face (configs[2] shape, fp32)", 128, 1080, 1920, -1), ("4K (configs[4] shape, fp32)", 32, 2160, 3840, 2)]:
    base = truely_amd.synthetic.synthetic_frames(min(n, 16), H, W, seed=0, faces=faces)
    fr = torch.from_numpy(base).cuda().repeat((n + len(base) - 1) // len(base), 1, 1, 1)[:n].contiguous()
    eng = Engine(truely_amd.weights.synthetic_blob(0), cap_level=3072, cap_frame=3072)
    for _ in range(2):
        out = eng.detect_embed(fr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = eng.detect_embed(fr); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    tm = eng.timings()
    res.append({"shape": name, "frames": n, "H": H, "W": W, "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(n / dt, 1),
                "pnet_ms": round(tm["pnet_ms"], 3), "pyramid_ms": round(tm["pyramid_ms"], 3), "valid_faces": int(out["valid"].sum().item())})
    del eng, fr
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
