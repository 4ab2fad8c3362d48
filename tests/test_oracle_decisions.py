"""C oracle vs the torch-CPU restatement at the DECISION level on every golden clip and the odd-size stress inputs, plus
the near-threshold audit.  The oracle is "parity unpinned" against the real facenet-pytorch (it
cannot be installed here); this is the strongest link available in-container: torch's own conv / pool / interpolate
kernels, with their accumulation order, select exactly the same candidates at every stage."""
import glob
import json
import os

import numpy as np
import pytest

import truely_amd
from decision_audit import audit_frame, merge_audits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# name, n, H, W, seed, faces (-1 = seeded 3..5 faces per frame)
STRESS = [("stress_200x150", 3, 200, 150, 22, 1), ("stress_64x333", 3, 64, 333, 23, 1), ("stress_multiface_360p", 2, 360, 640, 34, -1)]


def _cases():
    out = []
    for path in sorted(glob.glob(os.path.join(GOLD, "clip_*.npz"))):
        z = np.load(path)
        out.append((os.path.basename(path)[:-4], int(z["n"]), int(z["H"]), int(z["W"]), int(z["seed"]),
                    int(z["faces_per_frame"]) if "faces_per_frame" in z else 1))
    return out + STRESS


@pytest.fixture(scope="module")
def tref(state_dicts):
    from oracle.torch_ref import TorchRef
    return TorchRef(*state_dicts)


@pytest.mark.parametrize("name,n,H,W,seed,faces", _cases(), ids=[c[0] for c in _cases()])
def test_decisions_identical_and_audit(oracle, tref, name, n, H, W, seed, faces):
    fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed, faces=faces)
    audits, decided = [], 0
    for f in fr:
        a, nd = audit_frame(oracle, tref, f)
        audits.append(a)
        decided += nd["cand"]
    tot = merge_audits(audits)
    assert decided >= 0 and tot["pnet_prob_vs_thr0"]["n"] > 0      # clip_odd: no cell passes thr0, on either side
    # the committed audit table (profiles/round2_near_threshold_audit.json, tools/near_threshold_audit.py) is reproducible
    path = os.path.join(os.path.dirname(GOLD), "..", "profiles", "round2_near_threshold_audit.json")
    if os.path.exists(path):
        rec = json.load(open(path))["clips"].get(name)
        if rec is not None:
            for k, v in tot.items():
                assert rec["audit"][k]["n"] == v["n"] and rec["audit"][k]["within_1e-5"] == v["within_1e-5"], k
