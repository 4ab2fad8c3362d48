import sys; sys.path.insert(0,"/root/repo")
import torch, truely_amd, time
from truely_amd.engine import Engine
fr = torch.from_numpy(truely_amd.synthetic.synthetic_frames(64, 720, 1280, seed=0)).cuda()
fr = fr.repeat(4,1,1,1).contiguous()
eng = Engine(truely_amd.weights.synthetic_blob(0))
for i in range(3):
    out = eng.detect_embed(fr); print(eng.timings())
