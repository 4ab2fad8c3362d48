"""One-off fuzz beyond the test suite: random frame shapes / contents through the whole cascade against the oracle.
python tools/fuzz_shapes.py [cases] [seed] [pnet_run] [slopes]     (GPU box; prints the first mismatch and exits 1)
pnet_run > 0 forces the fused PNet kernel's tile runs (its halo-carry path) on these small frames; slopes = "general" uses the
generalised PReLU slopes (above 1 and negative: the med3 / min+max pooling instantiations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, truely_amd
from truely_amd.engine import Engine
from oracle.oracle import Oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
run = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 4 and sys.argv[4] == "general":
    blob = truely_amd.weights.pack_state_dicts(*truely_amd.weights.generalise_prelu(list(truely_amd.weights.synthetic_state_dicts(0))))
else:
    blob = truely_amd.weights.synthetic_blob(0)
eng, orc = Engine(blob, cap_level=3072, cap_frame=3072), Oracle(blob)
eng.pnet_run(run)
rng = np.random.default_rng(seed)
for t in range(cases):
    H, W, n = int(rng.integers(20, 420)), int(rng.integers(20, 560)), int(rng.integers(1, 4))
    if t % 2 == 0:
        W = max(20, W & ~3)                      # row pitch a multiple of 4: the crop's same-phase path
    kind = t % 4
    if kind == 0:
        fr = (rng.integers(0, 256, (n, H, W, 3)) // 8 + 112).astype(np.uint8)
    elif kind == 1:
        fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=5000 + t, faces=-1)
    elif kind == 2:
        fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=6000 + t, faces=1)
    else:                                        # smooth gradients + blocks: big low-frequency candidates
        yy, xx = np.mgrid[0:H, 0:W]
        base = ((np.sin(yy / rng.uniform(8, 40))[..., None] + np.cos(xx / rng.uniform(8, 40))[..., None]) * 60 + 128)
        fr = np.clip(base + rng.normal(0, 6, (n, H, W, 3)), 0, 255).astype(np.uint8)
    out = eng.detect_embed(fr)
    ref = orc.detect_embed(fr)
    bad = [k for k in ("box", "prob", "rect", "valid", "emb") if not np.array_equal(out[k].cpu().numpy(), ref[k])]
    for i in range(n):
        _b, _p, tr = orc.detect(fr[i], trace=True)
        for stage in (1, 2, 3):
            if not np.array_equal(eng.stage_boxes(stage, i), tr[f"boxes{stage}"]):
                bad.append(f"frame {i} stage {stage}")
    if bad:
        print(f"MISMATCH case {t}: {n} x {H}x{W} kind {kind}: {bad}")
        sys.exit(1)
    if t % 20 == 0:
        print(f"case {t}: {n} x {H}x{W} kind {kind} ok, faces {int(ref['valid'].sum())}", flush=True)
print(f"{cases} cases identical")
