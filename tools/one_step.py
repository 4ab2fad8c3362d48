import sys; sys.path.insert(0,"/root/repo")
import torch, truely_amd
from truely_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fr = torch.from_numpy(truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=0)).cuda()
eng = Engine(truely_amd.weights.synthetic_blob(0))
for i in range(2):
    out = eng.detect_embed(fr)
print(eng.timings())
