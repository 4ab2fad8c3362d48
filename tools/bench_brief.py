#!/usr/bin/env python3
"""Key figures of bench.py JSON lines (stdin or files): frames/s, ms/step, dominant-kernel time and roofline fraction."""
import json
import sys

for src in (sys.argv[1:] or ["-"]):
    for line in (sys.stdin if src == "-" else open(src)):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d.get("roofline", {})
        print("%-28s %9.1f %s  %7.3f ms/step  kernel %.3f ms frac %.4f  alone %s ms frac %s" % (
            src if src != "-" else "", d["value"], d["unit"], d["ms_per_step"], r.get("kernel_ms_per_step", 0), r.get("frac", 0),
            r.get("kernel_ms_alone"), r.get("frac_alone")))
