import sys; sys.path.insert(0,"/root/repo")
import torch, truely_amd, time, statistics
from truely_amd.engine import Engine
fr = torch.from_numpy(truely_amd.synthetic.synthetic_frames(64, 720, 1280, seed=0)).cuda()
fr = fr.repeat(4,1,1,1).contiguous()
eng = Engine(truely_amd.weights.synthetic_blob(0))
rows = []
for i in range(12):
    out = eng.detect_embed(fr); rows.append(eng.timings())
rows = rows[4:]
print({k: round(statistics.median(r[k] for r in rows), 3) for k in rows[0]}, "min pnet %.3f call %.3f" % (min(r["pnet_ms"] for r in rows), min(r["call_ms"] for r in rows)))
