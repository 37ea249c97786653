#!/usr/bin/env python3
"""Golden md5s for a dictionary whose table GROWS while it is built (QuicKmer.c:891-895, Resize_hash_table :738-779),
made by RUNNING THE REAL REFERENCE (oracle/_ref/quicKmer2 search -e 0) on the regenerated "mini" reference with a
table that starts too small: -s 64K doubles once, -s 16K three times.  Writes tests/golden/growth_manifest.json.

Below 128 KiB the reference's realloc of its occurrence array stays on the heap and the new half is never cleared
(:749), so what it reads there is whatever the heap held — undefined; the runs recorded here are the ones where that
memory was fresh (zeros), which is also what every table of a real size gets (mremap).  `-s 8K` (four doublings) is
NOT recorded: there the reference's own output depends on heap garbage."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "quicKmer2")
sys.path.insert(0, os.path.dirname(HERE))
from synth import mini_inputs  # noqa: E402


def md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


out = {"generator": "tests/golden/make_growth_golden.py", "reference_binary": "oracle/_ref/quicKmer2 search -k 30 -e 0 -w 1000 -c ctrl.bed", "cases": {}}
for size in ("64K", "16K"):
    with tempfile.TemporaryDirectory() as td:
        fa, ctrl, rfa, rfq = mini_inputs(td)
        txt = subprocess.run([REF_BIN, "search", "-k", "30", "-s", size, "-e", "0", "-w", "1000", "-c", ctrl, fa],
                             stdout=subprocess.PIPE, check=True).stdout.decode()
        out["cases"][size] = {"resizes": txt.count("Resize "), "qm_md5": md5(fa + ".qm"), "qgc_md5": md5(fa + ".qgc"),
                              "bed_md5": md5(fa + ".bed"), "qm_bytes": os.path.getsize(fa + ".qm")}
json.dump(out, open(os.path.join(HERE, "growth_manifest.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
