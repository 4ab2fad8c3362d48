"""OPTIONAL -- dumps REAL reference goldens.  Not runnable in the build container, never shipped as a dependency.

Every fixture in this directory was produced by the CPU oracle (make_golden.py), because the reference's arithmetic lives
in pip packages that are not installed here and cannot be fetched: parity is UNPINNED against the
real facenet-pytorch.  This script is the one route to pinning it.  Run it on any machine that has

    facenet-pytorch==2.6.0  torchvision  opencv-python  (requirements.txt:1,6,7,11 of the reference)

and it will, for each seeded synthetic clip the fixtures use,
  1. build `MTCNN()` and `InceptionResnetV1(pretrained="vggface2").eval()` exactly as server/model.py:18-19 does,
  2. pack their real weights with `truely_amd.weights.pack_state_dicts` into `reference_weights.trlw`,
  3. drive the library calls of server/model.py:47-66 on every frame (detect -> boxes[0] -> int cast + clamp -> crop ->
     cv2.resize 80x80 -> to_tensor -> embed -> cosine with the previous embedded frame -> run-length counters -> score),
  4. write `ref_<clip>.npz` with the SAME schema as the oracle-made fixtures (box, prob, rect, valid, emb, faces, sims,
     flags, score, run, hits, per-frame per-level candidate / keep counts) plus `source = "facenet-pytorch <version>"`.

tests/test_reference_goldens.py picks those files up automatically (oracle on CPU, HIP path on the GPU) and is skipped
while none exist.  Commit the .npz files and the .trlw blob (data, not source) to turn "parity unpinned" into a pinned claim.

    python tests/golden/dump_reference_goldens.py [--out tests/golden] [--pretrained vggface2]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [  # name, n, H, W, seed, faces   (same seeded inputs as make_golden.py, plus a multi-face clip)
    ("clip_180p", 6, 180, 320, 3, 1),
    ("clip_360p", 3, 360, 640, 11, 1),
    ("clip_odd", 3, 97, 131, 21, 1),
    ("clip_720p", 2, 720, 1280, 0, 1),
    ("clip_multiface_270p", 2, 270, 480, 33, -1),
]
THRESHOLD_SIMILARITY, THRESHOLD_FRAMES = 0.99, 15      # server/model.py:16-17


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--pretrained", default="vggface2")
    args = ap.parse_args()
    try:
        import cv2
        import torch
        import facenet_pytorch
        from facenet_pytorch import MTCNN, InceptionResnetV1
        from facenet_pytorch.models.utils import detect_face as df
        from torchvision.transforms.functional import to_tensor
    except Exception as e:  # noqa: BLE001
        sys.exit(f"dump_reference_goldens.py needs facenet-pytorch, torchvision and opencv-python ({e}); "
                 "it cannot run in the offline build container -- see the module docstring")
    import truely_amd

    mtcnn = MTCNN()                                                   # server/model.py:18
    net = InceptionResnetV1(pretrained=args.pretrained).eval()        # server/model.py:19
    sd = lambda m: {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}   # noqa: E731
    blob = truely_amd.weights.pack_state_dicts(sd(mtcnn.pnet), sd(mtcnn.rnet), sd(mtcnn.onet), sd(net))
    with open(os.path.join(args.out, "reference_weights.trlw"), "wb") as f:
        f.write(blob)

    # per-level candidate / keep counts: observe the library's own NMS calls (0.5 = one call per pyramid level)
    calls = []
    real_nms = df.batched_nms

    def spy(boxes, scores, idxs, thr):
        pick = real_nms(boxes, scores, idxs, thr)
        calls.append((float(thr), int(len(boxes)), int(len(pick))))
        return pick

    df.batched_nms = spy
    version = getattr(facenet_pytorch, "__version__", "2.6.0")
    for name, n, H, W, seed, faces in CASES:
        fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed, faces=faces)
        box = np.zeros((n, 4), np.float32); prob = np.zeros((n,), np.float32); rect = np.zeros((n, 4), np.int32)
        valid = np.zeros((n,), np.uint8); emb = np.zeros((n, 512), np.float32); face80 = np.zeros((n, 80, 80, 3), np.uint8)
        sims = np.full((n,), 2.0, np.float32); flags = np.zeros((n,), np.uint8)
        stages = {}
        prev, run, hits = None, 0, 0
        for i in range(n):
            frame = fr[i]
            calls.clear()
            boxes, probs = mtcnn.detect(frame)                        # server/model.py:47
            lv = [c for c in calls if c[0] == 0.5]
            stages[f"f{i}_cand"] = np.array([c[1] for c in lv], np.int32)
            stages[f"f{i}_keep"] = np.array([c[2] for c in lv], np.int32)
            if boxes is None:
                continue
            box[i], prob[i] = boxes[0], probs[0]
            x0, y0, x1, y1 = boxes[0].astype(int)                     # server/model.py:49-53
            x0, y0, x1, y1 = max(0, x0), max(0, y0), min(frame.shape[1], x1), min(frame.shape[0], y1)
            rect[i] = (x0, y0, x1, y1)
            if x1 <= x0 or y1 <= y0:
                continue
            face = cv2.resize(frame[y0:y1, x0:x1], (80, 80))          # server/model.py:55-57
            face80[i] = face
            with torch.no_grad():
                e = net(to_tensor(face).unsqueeze(0)).numpy().reshape(-1)   # server/model.py:58-59
            emb[i], valid[i] = e, 1
            if prev is not None:                                      # server/model.py:60-66
                s = float(np.dot(e, prev) / (np.linalg.norm(e) * np.linalg.norm(prev)))
                sims[i] = s
                run = run + 1 if s < THRESHOLD_SIMILARITY else 0
                if run > THRESHOLD_FRAMES:
                    hits += 1; flags[i] = 1
            prev = e
        frame_count, fps = n * 4, 30                                  # fixtures treat the clip as every 4th frame of a 30 fps video
        total = -(-frame_count // max(1, int(fps / 7)))
        pct = 100.0 * hits / total if total else 0.0                  # server/model.py:86-95
        conf = min(pct * (run / THRESHOLD_FRAMES), 100.0)
        score = int(max(0, min(100, min(pct + conf * (0.5 if frame_count > fps * 30 else 0.3), 100.0))))
        np.savez_compressed(os.path.join(args.out, f"ref_{name}.npz"), n=n, H=H, W=W, seed=seed, faces_per_frame=faces,
                            frames_crc=np.uint64(int(fr.astype(np.uint64).sum())), source=f"facenet-pytorch {version}",
                            box=box, prob=prob, rect=rect, valid=valid, emb=emb, faces=face80, sims=sims, flags=flags,
                            score=score, run=run, hits=hits, **stages)
        print(name, "valid", valid.tolist(), "score", score)
    df.batched_nms = real_nms


if __name__ == "__main__":
    main()
