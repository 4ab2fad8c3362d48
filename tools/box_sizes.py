"""Size distribution of the R-Net / O-Net candidate boxes of the bench clip (what the front kernels' crop paths see).
python tools/box_sizes.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, truely_amd
from truely_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fr = truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=0)
eng = Engine(truely_amd.weights.synthetic_blob(0))
eng.detect_embed(fr)
for stage, S in ((1, 24), (2, 48)):
    sz = []
    for i in range(n):
        b = eng.stage_boxes(stage, i)
        sz.append(np.maximum(b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]))
    sz = np.concatenate(sz)
    edges = [0, S, 2 * S, 3 * S, 6 * S, 12 * S, 1e9]
    h, _ = np.histogram(sz, edges)
    print(f"stage {stage} -> net input {S}: {len(sz) / n:.1f} boxes/frame; side <= S, 2S, 3S (one-load path), 6S, 12S, more:", (h / len(sz)).round(3).tolist(),
          "median", float(np.median(sz)))
