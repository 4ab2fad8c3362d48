"""Dev tool (GPU box): class-logit offsets that give the `--prelu general` synthetic weights the SAME candidate mix as the seeded
unit-slope weights on the bench clip, so that the two bench lines differ only in which kernel instantiations run.

Changing PReLU slopes changes what the random nets compute, and with it how many boxes reach R-Net / O-Net (measured: 58.5 k / 17.3 k
per 256 frames against 29.8 k / 8.2 k) -- twice the candidate work, which says nothing about the kernels.  The face-logit bias of each
net is a free parameter of the synthetic weights (weights.synthetic_state_dicts calibrates it the same way for the seeded slopes):
bisect it per net until the stage totals match.  Prints the offsets to paste into weights.GENERAL_PRELU_MATCH.

    python tools/calibrate_general_prelu.py [frames]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import truely_amd  # noqa: E402
from truely_amd.engine import Engine  # noqa: E402
from truely_amd import weights  # noqa: E402

CLS = {"pnet": "conv4_1", "rnet": "dense5_1", "onet": "dense6_1"}


def totals(sds, frames):
    eng = Engine(weights.pack_state_dicts(*sds), cap_level=3072, cap_frame=3072)
    out = eng.mtcnn_detect(frames)
    t2, t3 = eng.stage_totals()
    n3 = int(out[2].sum().item())
    eng.close()
    return t2, t3, n3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    frames = truely_amd.synthetic.synthetic_frames(n, 720, 1280, seed=0)
    target = totals(weights.synthetic_state_dicts(0), frames)
    print("unit slopes: T2, T3, faces =", target)
    off = {"pnet": 0.0, "rnet": 0.0, "onet": 0.0}

    def build():
        sds = weights.generalise_prelu(list(weights.synthetic_state_dicts(0)), match=off)
        return sds

    print("general, unmatched:", totals(build(), frames))
    for k, net in enumerate(("pnet", "rnet", "onet")):
        lo, hi = -6.0, 6.0                       # totals[k] grows with the net's face-logit offset
        for _ in range(22):
            off[net] = 0.5 * (lo + hi)
            try:
                got = totals(build(), frames)[k]
            except Exception:                    # capacity overflow = far too many candidates
                got = 1 << 30
            if got > target[k]:
                hi = off[net]
            else:
                lo = off[net]
        off[net] = lo
        print(net, "offset", off[net], "->", totals(build(), frames))
    print("GENERAL_PRELU_MATCH =", {k: round(v, 6) for k, v in off.items()})


if __name__ == "__main__":
    main()
