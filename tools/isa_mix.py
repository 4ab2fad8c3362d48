#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (dev tool).

  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only -o /tmp/k.s file.hip
  python tools/isa_mix.py /tmp/k.s k_pnet_fused

f32 MFMA and f32 VALU share the FP32 datapath on gfx950, so the VALU count of the blocks
inside the MFMA loops is paid in matrix throughput.
"""
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], {"lbl": "entry", "valu": 0, "mfma": 0, "ds": 0, "salu": 0, "vmem": 0, "br": []}
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append(cur)
            cur = {"lbl": m.group(1), "valu": 0, "mfma": 0, "ds": 0, "salu": 0, "vmem": 0, "br": []}
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                cur["br"].append(t[1])
    blocks.append(cur)
    tot = {k: 0 for k in ("valu", "mfma", "ds", "salu", "vmem")}
    for b in blocks:
        if b["valu"] + b["mfma"] + b["ds"] + b["vmem"] < 8:
            for k in tot:
                tot[k] += b[k]
            continue
        print("%-12s valu=%4d mfma=%4d ds=%4d salu=%4d vmem=%3d -> %s" % (b["lbl"], b["valu"], b["mfma"], b["ds"], b["salu"], b["vmem"], " ".join(b["br"])))
        for k in tot:
            tot[k] += b[k]
    print("static total", tot)


if __name__ == "__main__":
    main()
