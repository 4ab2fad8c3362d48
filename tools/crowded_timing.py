"""Wall time of detect_embed on content that overflows every start capacity (the spill tier, grow-and-re-run):
    python tools/crowded_timing.py [--pathological]
Prints one JSON line per case: attempts of the first call, ms of the first call (with the re-runs) and of the second (capacities
settled), list statistics.  --pathological adds the 4K frame at thr0 = 0.3 (736 k candidates at the finest level)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import truely_amd  # noqa: E402
from truely_amd.engine import Engine  # noqa: E402


def case(name, eng, fr):
    fr = torch.from_numpy(fr).cuda()
    torch.cuda.synchronize()
    t0 = time.time(); eng.detect_embed(fr); torch.cuda.synchronize(); t1 = time.time()
    st1 = eng.list_stats()
    eng.detect_embed(fr); torch.cuda.synchronize(); t2 = time.time()
    st2 = eng.list_stats()
    print(json.dumps({"case": name, "frames": int(fr.shape[0]), "first_call_ms": round(1e3 * (t1 - t0), 1), "first_call_attempts": st1["attempts"],
                      "second_call_ms": round(1e3 * (t2 - t1), 1), "second_call_attempts": st2["attempts"], "stage_totals": eng.stage_totals(),
                      **{k: st2[k] for k in ("spill_lists", "spill_used", "cap_frame", "slots_per_frame", "max_level_count", "max_frame_total")}}), flush=True)


def main():
    blob = truely_amd.weights.synthetic_blob(0)
    rng = np.random.default_rng(1)
    calm = truely_amd.synthetic.synthetic_frames(4, 720, 1280, seed=0)
    case("720p synthetic x4 (no overflow)", Engine(blob), calm)
    case("720p uniform noise x1", Engine(blob), rng.integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8))
    case("720p uniform noise x16", Engine(blob), rng.integers(0, 256, (16, 720, 1280, 3), dtype=np.uint8))
    f4k = truely_amd.synthetic.synthetic_frames(1, 2160, 3840, seed=32, faces=1)
    case("4K synthetic, thr0 0.6", Engine(blob), f4k)
    case("4K synthetic, thr0 0.55", Engine(blob, thresholds=(0.55, 0.7, 0.7)), f4k)
    if "--pathological" in sys.argv:
        case("4K synthetic, thr0 0.5", Engine(blob, thresholds=(0.5, 0.7, 0.7)), f4k)
        case("4K synthetic, thr0 0.3", Engine(blob, thresholds=(0.3, 0.7, 0.7)), f4k)


if __name__ == "__main__":
    main()
