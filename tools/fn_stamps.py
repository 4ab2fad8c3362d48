"""Where do the waves of one small-map conv launch spend their cycles?  python tools/fn_stamps.py <launch index in one FaceNet pass>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, truely_amd
from truely_amd.engine import Engine
from truely_amd import _lib
eng = Engine(truely_amd.weights.synthetic_blob(0))
lib = _lib.load()
x = torch.rand(256, 80, 80, 3, device="cuda")
for _ in range(3): eng.facenet_embed(x)
torch.cuda.synchronize()
for k in [int(v) for v in sys.argv[1:]]:
    lib.trl_debug_fn_arm(C.c_int(k))
    eng.facenet_embed(x)
    buf = np.zeros((1 << 16, 8), np.uint64); n = C.c_int()
    lib.trl_debug_fn_read(buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(n))
    r = buf[:n.value].astype(np.int64)
    r = r[r[:, 0] > 0]
    t0 = r[:, 0].min()
    seg = np.stack([r[:, 0] - t0, r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2], r[:, 4] - r[:, 3]], 1)
    wall = (r[:, 5].max() - r[:, 5].min()) / 100.0
    print(f"launch {k}: {len(r)} waves; cycles (mean / max): start skew {seg[:,0].mean():.0f}/{seg[:,0].max()}  prologue+issue {seg[:,1].mean():.0f}/{seg[:,1].max()}  "
          f"first data {seg[:,2].mean():.0f}/{seg[:,2].max()}  K loop {seg[:,3].mean():.0f}/{seg[:,3].max()}  epilogue {seg[:,4].mean():.0f}/{seg[:,4].max()}  "
          f"| first start -> last end {(r[:,4].max() - t0)} cycles; end-stamp wall spread {wall:.2f} us")
