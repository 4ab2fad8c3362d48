"""Bit-exactness under poisoned workspaces: results must not depend on stale activation memory.
python tools/poison_check.py"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, truely_amd
from truely_amd.engine import Engine
blob = truely_amd.weights.synthetic_blob(0)
sets = [truely_amd.synthetic.synthetic_frames(6, 180, 320, seed=3), truely_amd.synthetic.synthetic_frames(5, 97, 131, seed=21),
        truely_amd.synthetic.synthetic_frames(8, 360, 640, seed=11), truely_amd.synthetic.synthetic_frames(16, 720, 1280, seed=0)]
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
eng = {"default": lambda: Engine(blob), "native": lambda: Engine(blob, embed_mode=2), "bf16": lambda: Engine(blob, embed_precision="bf16"), "fp16": lambda: Engine(blob, embed_precision="fp16"),
       "generic_pnet": lambda: Engine(blob, pnet_mode=1)}[mode]()
ref = []
for fr in sets:
    out = eng.detect_embed(fr)
    ref.append({k: out[k].cpu().numpy().copy() for k in ("box", "prob", "rect", "valid", "emb")})
bad = 0
for byte in (0xFF, 0x7F, 0x00, 0xFF):
    for si, fr in enumerate(sets):
        eng.poison_workspaces(byte)
        out = eng.detect_embed(fr)
        for k in ref[si]:
            b = out[k].cpu().numpy()
            if not np.array_equal(ref[si][k], b, equal_nan=False):
                bad += 1
                d = np.abs(ref[si][k].astype(np.float64) - b.astype(np.float64))
                print(f"poison 0x{byte:02X} set {si} key {k}: max diff {np.nanmax(d):.3e} nan={int(np.isnan(b.astype(np.float64)).sum())}")
print(mode, "mismatches:", bad)
