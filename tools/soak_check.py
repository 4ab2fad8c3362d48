"""Soak: the overlapped pipeline (F cascades in flight, grouped embedder, crop ring reuse) for many steps on the same batch; EVERY
step's outputs must equal the first step's, bit for bit.  python tools/soak_check.py [steps] [in_flight] [embed_group] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, truely_amd
from truely_amd.engine import Engine
from truely_amd.pipeline import detect_embed_overlapped

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
blob = truely_amd.weights.synthetic_blob(0)
fr = torch.from_numpy(truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=0)).cuda()
ref = Engine(blob).detect_embed(fr)
engs = [Engine(blob) for _ in range(F)]
bad = []

def check(i, out):
    for k in ("box", "prob", "rect", "valid", "emb"):
        if not torch.equal(out[k], ref[k]):
            bad.append((i, k))
    if i % 100 == 0:
        print(f"step {i}: {'ok' if not bad else bad[:3]}", flush=True)

detect_embed_overlapped(engs, lambda i, j: fr, on_result=check, embed_group=G, n_batches=steps)
print(f"{steps} steps, F={F}, G={G}: {len(bad)} mismatching (step, key) pairs")
sys.exit(1 if bad else 0)
