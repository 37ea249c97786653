#!/usr/bin/env python3
"""Stand-in for the reference's GC-smoothing helper, for tests only: same interface
(`smooth_GC_mrsfast.py <sample>.txt` on PATH, 401 little-endian float32 on stdout,
QuicKmer.c:632-640), deterministic curve.  It is NOT the reference's LOWESS script (that one
needs numpy < 2 and matplotlib); `est` treats the curve as an opaque input, so any fixed
curve pins the C side.  correction[i] = mean depth / depth of GC bin i, clipped to [1/4, 4];
1 for bins without control k-mers."""
import struct
import sys

rows = []
with open(sys.argv[1]) as f:
    for line in f:
        p = line.split("\t")
        if len(p) >= 3:
            rows.append((float(p[1]), int(p[2])))
rows = (rows + [(0.0, 0)] * 401)[:401]
total = sum(c for _, c in rows)
mean = sum(d * c for d, c in rows) / total if total else 0.0
out = []
for d, c in rows:
    v = mean / d if c and d > 0 else 1.0
    out.append(min(4.0, max(0.25, v)))
sys.stdout.buffer.write(struct.pack("<401f", *out))
