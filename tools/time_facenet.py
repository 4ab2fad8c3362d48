import sys; sys.path.insert(0,"/root/repo")
import torch, truely_amd
from truely_amd.engine import Engine
eng = Engine(truely_amd.weights.synthetic_blob(0))
x = torch.rand(256, 80, 80, 3, device="cuda")
for _ in range(3): eng.facenet_embed(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.facenet_embed(x)
e1.record(); torch.cuda.synchronize()
print("facenet 256x80x80: %.3f ms" % (e0.elapsed_time(e1) / 20))
