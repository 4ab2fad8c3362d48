"""Run the same batch repeatedly (fresh engines and repeated calls) and report any run-to-run difference.
python tools/determinism_check.py [trials]"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, truely_amd
from truely_amd.engine import Engine

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blob = truely_amd.weights.synthetic_blob(0)
sets = [truely_amd.synthetic.synthetic_frames(6, 180, 320, seed=3), truely_amd.synthetic.synthetic_frames(8, 360, 640, seed=11),
        truely_amd.synthetic.synthetic_frames(16, 720, 1280, seed=0)]
ref = None
bad = 0
for t in range(trials):
    eng = Engine(blob) if t % 4 == 0 else eng
    cur = []
    for fr in sets:
        out = eng.detect_embed(fr)
        cur.append({k: out[k].cpu().numpy().copy() for k in ("box", "prob", "rect", "valid", "emb")})
    if ref is None:
        ref = cur
        continue
    for si, (a, b) in enumerate(zip(ref, cur)):
        for k in a:
            if not np.array_equal(a[k], b[k]):
                bad += 1
                d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
                print(f"trial {t} set {si} key {k}: max diff {d.max():.3e} at frames {sorted(set(np.argwhere(d > 0)[:, 0].tolist()))[:8]}")
print("trials", trials, "mismatching (set,key) pairs:", bad)
