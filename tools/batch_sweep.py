"""Batch sweep ("sweep batch to trace the ... roofline knee"): frames/s of trl_detect_embed and the PNet kernel's
TFLOP/s against the frames per call, one batch in flight, for a BASELINE config's frame shape.

    python tools/batch_sweep.py CONFIG OUT.json [batches...]      (CONFIG = 1, 2 or 4 as in bench.py)

The kernel is a persistent launch over (frame, level, tile) work items: below ~2 tiles per resident workgroup the launch cannot
fill the 512 workgroup slots and the step is dominated by fixed costs; the knee is where tiles >> workgroups."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import truely_amd
from truely_amd.engine import Engine

cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out_path = sys.argv[2] if len(sys.argv) > 2 else None
cfg = bench.CONFIGS[cfg_id]
batches = [int(b) for b in sys.argv[3:]] or {1: [4, 8, 16, 32, 64, 128, 256, 512], 2: [2, 4, 8, 16, 32, 64, 128, 256],
                                              4: [1, 2, 4, 8, 16, 32, 64]}[cfg_id]
H, W = cfg["H"], cfg["W"]
base = bench.make_clip(cfg, min(max(batches), cfg["unique"]), seed=0)
ekw = dict(min_face_size=cfg["min_face"], embed_precision=cfg["embed"])
if H > 1080:
    ekw.update(cap_level=3072, cap_frame=3072)
eng = Engine(truely_amd.weights.synthetic_blob(0), **ekw)
macs1 = bench.pnet_macs(H, W, cfg["min_face"])
rows = []
for n in batches:
    fr = np.stack([base[i % len(base)] if i < len(base) else np.roll(base[i % len(base)], 11 * (i // len(base)), axis=1) for i in range(n)])
    x = torch.from_numpy(fr).cuda()
    for _ in range(3):
        eng.detect_embed(x)
    steps = max(4, min(40, 2048 // n))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pnet = 0.0
    for _ in range(steps):
        eng.detect_embed(x)
        pnet += eng.timings()["pnet_ms"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pm = pnet / steps
    rows.append({"frames_per_call": n, "ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(n / dt, 1), "pnet_ms": round(pm, 3),
                 "pnet_tflops": round(2.0 * macs1 * n / (pm * 1e-3) / 1e12, 2),
                 "pnet_frac_of_f32_mfma_peak": round(2.0 * macs1 * n / (pm * 1e-3) / 1e12 / bench.PEAK_F32_MFMA_TFLOPS, 4)})
    print(rows[-1], flush=True)
    del x
res = {"config": cfg_id, "workload": cfg["name"], "in_flight": 1, "levels": eng.levels(H, W), "rows": rows}
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
