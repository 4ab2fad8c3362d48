"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot be imported here:
facenet_pytorch / cv2 / torchvision are absent, SURVEY.md section 8c -> PARITY UNPINNED).

A fixture is data only: seeded inputs are regenerated from (seed, shape) by
truely_amd.synthetic, expected outputs are stored.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import truely_amd  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

CASES = [  # name, n, H, W, seed, faces per frame (-1 = seeded 3..5)
    ("clip_180p", 6, 180, 320, 3, 1),
    ("clip_360p", 3, 360, 640, 11, 1),
    ("clip_odd", 3, 97, 131, 21, 1),
    ("clip_720p", 2, 720, 1280, 0, 1),
    ("clip_multiface_270p", 2, 270, 480, 33, -1),
]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    o = Oracle(truely_amd.weights.synthetic_blob(0))
    only = sys.argv[1:]
    for name, n, H, W, seed, faces in CASES:
        if only and name not in only:
            continue
        fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed, faces=faces)
        r = o.detect_embed(fr, want_faces=True)
        d = o.drift_score(r["emb"], r["valid"], n * 4, 30)
        stages = {}
        for i in range(n):
            _b, _p, tr = o.detect(fr[i], trace=True)
            stages[f"f{i}_cand"] = np.array(tr["n_cand_scale"], np.int32)
            stages[f"f{i}_keep"] = np.array(tr["n_keep_scale"], np.int32)
            for s in (1, 2, 3):
                stages[f"f{i}_boxes{s}"] = tr[f"boxes{s}"]
        np.savez_compressed(os.path.join(here, name + ".npz"), n=n, H=H, W=W, seed=seed, faces_per_frame=faces,
                            frames_crc=np.uint64(int(fr.astype(np.uint64).sum())),
                            box=r["box"], prob=r["prob"], rect=r["rect"], valid=r["valid"], emb=r["emb"],
                            faces=r["faces"], sims=d["sims"], flags=d["flags"], score=d["score"], run=d["run"], hits=d["hits"],
                            **stages)
        print(name, "valid", r["valid"].tolist(), "score", d["score"])
    if only:
        return
    # primitive vectors
    rng = np.random.default_rng(123)
    x = rng.uniform(0, 1, (2, 80, 80, 3)).astype(np.float32)
    c24 = rng.uniform(-1, 1, (4, 24, 24, 3)).astype(np.float32)
    c48 = rng.uniform(-1, 1, (3, 48, 48, 3)).astype(np.float32)
    pr, rr = o.rnet(c24)
    po, ro, pt = o.onet(c48)
    np.savez_compressed(os.path.join(here, "primitives.npz"), facenet_in=x, facenet_out=o.facenet(x),
                        rnet_in=c24, rnet_prob=pr, rnet_reg=rr, onet_in=c48, onet_prob=po, onet_reg=ro, onet_pts=pt)


if __name__ == "__main__":
    main()
