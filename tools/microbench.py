#!/usr/bin/env python3
"""Random-access ceilings of one MI355X for tables of the count path's sizes
(SURVEY 7.1 step 7): uniformly random 8-byte loads, 4-byte no-return atomic
adds, and load-then-atomic, 16 independent operations in flight per lane.
These are the empirical denominators quoted next to roofline.frac in DESIGN.md.
Usage: python tools/microbench.py [lgH ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))
import pyqm2  # noqa: E402

names = {0: "load8", 1: "atomic4", 2: "load8+atomic4"}
n_ops = 1 << 30
out = []
for lg in [int(a) for a in sys.argv[1:]] or [21, 28, 32]:
    for mode in (0, 1, 2):
        ms = pyqm2.tools.microbench(1 << lg, n_ops, mode, repeats=3)
        r = {"slots": "2^%d" % lg, "op": names[mode], "ms": round(ms, 3),
             "Gops_per_s": round(n_ops / ms / 1e6, 2)}
        print(json.dumps(r), flush=True)
        out.append(r)
