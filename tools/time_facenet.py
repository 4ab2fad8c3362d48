"""FaceNet alone: ms per 256 faces (80x80) through trl_facenet_embed.  python tools/time_facenet.py [iters] [faces]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, truely_amd
from truely_amd.engine import Engine
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eng = Engine(truely_amd.weights.synthetic_blob(0))
x = torch.rand(nf, 80, 80, 3, device="cuda")
for _ in range(3): eng.facenet_embed(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it): eng.facenet_embed(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
print("facenet %dx80x80: %.3f ms (%.3f ms per 256 faces)" % (nf, ms, ms * 256 / nf))
