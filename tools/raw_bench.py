#!/usr/bin/env python3
"""Time of the device line discipline on a FASTQ already in HBM: 10 M x 150 bp
records (3.07 GB of text), chr22-like dictionary; prints the three-kernel time per launch, its
bandwidth (1 B read by raw_tile + 1 B read and 1 B written by raw_apply per input byte) and the
count pipeline's time on the neutralised copy next to the stripped-input time."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pyqm2  # noqa: E402
from bench import genome_layout  # noqa: E402

n, H, k, n_reads = 50_818_468, 1 << 28, 30, 10_000_000
genome = torch.empty(n, dtype=torch.uint8, device="cuda")
n_blocks, dups, covered = genome_layout(n)
pyqm2.tools.genome_generate(genome.data_ptr(), n, seed=22, n_blocks=n_blocks, dups=dups)
keys = torch.empty(H, dtype=torch.int64, device="cuda")
scratch = torch.empty(H, dtype=torch.int64, device="cuda")
pyqm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
del scratch
reads = torch.empty(n_reads * 151, dtype=torch.uint8, device="cuda")
pyqm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, 150, 150, 22, sample_from=covered)
seq = reads.view(n_reads, 151)
fq = torch.empty((n_reads, 307), dtype=torch.uint8, device="cuda")       # "@r\n" seq "\n" "+\n" qual "\n"
fq[:, 0:3] = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device="cuda")
fq[:, 3:154] = seq
fq[:, 154:156] = torch.tensor(list(b"+\n"), dtype=torch.uint8, device="cuda")
fq[:, 156:306] = ord("I")
fq[:, 306] = 10
fq = fq.reshape(-1)
torch.cuda.synchronize()
out = {}
for label, run in (("stripped", lambda c: c.submit_device(reads.data_ptr(), reads.numel())),
                   ("raw_fastq", lambda c: c.submit_raw_device(fq.data_ptr(), fq.numel(), True, skip=1))):
    with pyqm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), flags=pyqm2.FLAG_TIMING) as c:
        for _ in range(2):
            run(c)
        c.sync()
        p0 = c.profile()
        for _ in range(5):
            run(c)
        _, st = c.finish(want_depth=False)
        p1 = c.profile()
        ms = {kk: (p1["class_ms"][kk] - p0["class_ms"][kk]) / 5 for kk in p1["class_ms"]}
        out[label] = {"ms_per_launch": {kk: round(v, 3) for kk, v in ms.items() if v > 0}, "n_kmers": st["n_kmers"], "n_hits": st["n_hits"]}
lines_ms = out["raw_fastq"]["ms_per_launch"]["lines"]
out["line_discipline"] = {"bytes": int(fq.numel()), "ms": lines_ms, "GBs_moved": round(3 * fq.numel() / lines_ms / 1e6, 1),
                          "frac_of_8TBs": round(3 * fq.numel() / lines_ms / 1e6 / 8000, 3)}
assert out["raw_fastq"]["n_kmers"] == out["stripped"]["n_kmers"] and out["raw_fastq"]["n_hits"] == out["stripped"]["n_hits"]
print(json.dumps(out))
