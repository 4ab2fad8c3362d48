"""Consumes REAL facenet-pytorch goldens when someone has produced them (tests/golden/dump_reference_goldens.py, not
runnable offline).  Until `tests/golden/ref_*.npz` + `reference_weights.trlw` exist every test here is skipped and parity
stays "unpinned" (DESIGN.md section 2).  Bars (BASELINE.json north_star): decisions -- valid mask, integer crop rectangle,
per-level candidate / keep counts -- bit-exact; boxes within 1e-3 px; embeddings and drift similarities within 1e-4."""
import glob
import os

import numpy as np
import pytest

import truely_amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = sorted(glob.glob(os.path.join(GOLD, "ref_*.npz")))
WEIGHTS = os.path.join(GOLD, "reference_weights.trlw")
needs_ref = pytest.mark.skipif(not REF or not os.path.exists(WEIGHTS), reason="no real reference goldens committed (parity unpinned)")


def _check(z, r, counts):
    n = int(z["n"])
    assert np.array_equal(r["valid"], z["valid"]) and np.array_equal(r["rect"], z["rect"])
    assert np.abs(r["box"] - z["box"]).max() <= 1e-3 and np.abs(r["prob"] - z["prob"]).max() <= 1e-5
    assert np.abs(r["emb"] - z["emb"]).max() <= 1e-4
    for i in range(n):
        cand, keep = counts(i)
        assert cand == z[f"f{i}_cand"].tolist() and keep == z[f"f{i}_keep"].tolist()


@needs_ref
@pytest.mark.parametrize("path", REF or [None])
def test_oracle_against_reference_goldens(path):
    from oracle.oracle import Oracle
    z = np.load(path)
    orc = Oracle(open(WEIGHTS, "rb").read())
    fr = truely_amd.synthetic.synthetic_frames(int(z["n"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), faces=int(z["faces_per_frame"]))
    assert int(fr.astype(np.uint64).sum()) == int(z["frames_crc"])
    r = orc.detect_embed(fr, want_faces=True)
    assert np.array_equal(r["faces"][z["valid"].astype(bool)], z["faces"][z["valid"].astype(bool)])   # cv2.resize, byte for byte
    traces = [orc.detect(f, trace=True)[2] for f in fr]
    _check(z, r, lambda i: (traces[i]["n_cand_scale"], traces[i]["n_keep_scale"]))
    d = orc.drift_score(r["emb"], r["valid"], int(z["n"]) * 4, 30)
    assert d["score"] == int(z["score"]) and np.abs(d["sims"] - z["sims"]).max() <= 1e-4


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("path", REF or [None])
def test_hip_path_against_reference_goldens(path):
    from truely_amd.engine import Engine
    z = np.load(path)
    eng = Engine(open(WEIGHTS, "rb").read())
    fr = truely_amd.synthetic.synthetic_frames(int(z["n"]), int(z["H"]), int(z["W"]), seed=int(z["seed"]), faces=int(z["faces_per_frame"]))
    out = {k: v.cpu().numpy() for k, v in eng.detect_embed(fr).items()}
    _check(z, out, lambda i: eng.level_counts(i))
    d = eng.drift_score(eng.detect_embed(fr)["emb"], eng.detect_embed(fr)["valid"], int(z["n"]) * 4, 30)
    assert d["score"] == int(z["score"]) and np.abs(d["sims"].cpu().numpy() - z["sims"]).max() <= 1e-4
