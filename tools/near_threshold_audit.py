"""Near-threshold audit of the MTCNN cascade: for every golden clip and stress input, how many
compared quantities (PNet / R-Net / O-Net face probabilities, NMS overlaps) come within 1e-5 / 1e-6 of their decision
threshold under torch's accumulation order, and confirmation that the C oracle and the torch restatement decide
identically.  Writes profiles/round2_near_threshold_audit.json and prints the markdown table DESIGN.md quotes.
CPU only:  python tools/near_threshold_audit.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import truely_amd  # noqa: E402
from decision_audit import audit_frame, merge_audits  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from oracle.torch_ref import TorchRef  # noqa: E402
from test_oracle_decisions import _cases  # noqa: E402

KEYS = ["pnet_prob_vs_thr0", "stage1a_iou_vs_0.5", "stage1b_iou_vs_0.7", "rnet_prob_vs_thr1", "stage2_iou_vs_0.7",
        "onet_prob_vs_thr2", "stage3_iom_vs_0.7"]


def main():
    sds = truely_amd.weights.synthetic_state_dicts(0)
    orc, ref = Oracle(truely_amd.weights.pack_state_dicts(*sds)), TorchRef(*sds)
    out = {"note": "torch-CPU accumulation order (oracle/torch_ref.py); decisions asserted identical to the C oracle", "clips": {}}
    print("| clip | frames | " + " | ".join(k.replace("_vs_", " vs ") for k in KEYS) + " | decisions identical |")
    print("|---|---|" + "---|" * (len(KEYS) + 1))
    for name, n, H, W, seed, faces in _cases():
        fr = truely_amd.synthetic.synthetic_frames(n, H, W, seed=seed, faces=faces)
        audits, decs = [], []
        for f in fr:
            a, nd = audit_frame(orc, ref, f)
            audits.append(a); decs.append(nd)
        tot = merge_audits(audits)
        out["clips"][name] = {"frames": n, "H": H, "W": W, "seed": seed, "faces": faces, "audit": tot, "decisions": decs}
        cells = []
        for k in KEYS:
            v = tot.get(k)
            cells.append("-" if v is None else f"{v['n']} cmp, {v['within_1e-5']} / {v['within_1e-6']} near, min {v['min_margin']:.1e}")
        print(f"| {name} | {n} | " + " | ".join(cells) + " | yes |")
    with open(os.path.join(ROOT, "profiles", "round2_near_threshold_audit.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
