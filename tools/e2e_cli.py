#!/usr/bin/env python3
"""End-to-end timing of the `quicKmer2 count` CLI at config-C2 scale on the GPU box:
synthetic chr22-like dictionary written as a real .qm (chain = occupied slots in ascending
order: a valid cycle, enough for timing), 10 M x 150 bp reads as FASTQ and as FASTA on local
disk, then the CLI with wall-clock per phase from its own output.  This is the
PCIe- and host-inclusive number (never bench.py's `value`)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pyqm2  # noqa: E402
from bench import genome_layout  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/qm2_e2e"
os.makedirs(work, exist_ok=True)
scale = sys.argv[3] if len(sys.argv) > 3 else "c2"
n, H, k = (3_100_000_000, 1 << 32, 30) if scale == "c3" else (50_818_468, 1 << 28, 30)
if scale == "c3":                        # 51 GB .qm + reads + 4.5 GB .bin must fit on the scratch disk
    import shutil
    free = shutil.disk_usage(work).free
    print("scratch disk free: %.0f GB" % (free / 1e9), flush=True)
    if free < 70e9:
        sys.exit("not enough scratch space for the hg38-scale run")
genome = torch.empty(n, dtype=torch.uint8, device="cuda")
n_blocks, dups, covered = genome_layout(n)
pyqm2.tools.genome_generate(genome.data_ptr(), n, seed=22, n_blocks=n_blocks, dups=dups)
keys = torch.empty(H, dtype=torch.int64, device="cuda")
scratch = torch.empty(H, dtype=torch.int64, device="cuda")
nd, nu = pyqm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
del scratch
hk = keys.cpu().numpy().view(np.uint64)
occ = np.flatnonzero(hk).astype(np.uint32)
chain = np.zeros(H, dtype=np.uint32)
chain[occ] = np.roll(occ, -1)
t = time.time()
with open(os.path.join(work, "ref.fa.qm"), "wb") as f:
    f.write(b"QM11" + bytes([k, 0, 100, 100]) + int(H).to_bytes(8, "little") + int(occ[0]).to_bytes(8, "little"))
    hk.tofile(f)
    chain.tofile(f)
print("wrote .qm %.1f GB in %.1f s" % (os.path.getsize(os.path.join(work, "ref.fa.qm")) / 1e9, time.time() - t), flush=True)
del hk, chain
reads = torch.empty(n_reads * 151, dtype=torch.uint8, device="cuda")
pyqm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, 150, 150, 22, sample_from=covered)
torch.cuda.synchronize()
lines = reads.cpu().numpy()
del reads, genome, keys
torch.cuda.empty_cache()
t = time.time()
arr = lines.reshape(n_reads, 151)
fq = np.empty((n_reads, 4 + 151 + 2 + 151), dtype=np.uint8)      # "@r\n" seq "\n" "+\n" qual "\n"
fq[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
fq[:, 3:154] = arr
fq[:, 154:156] = np.frombuffer(b"+\n", dtype=np.uint8)
fq[:, 156:306] = ord("I")
fq[:, 306] = 10
fq = fq[:, :307]
fq.tofile(os.path.join(work, "reads.fq"))
fa = np.empty((n_reads if scale != "c3" else 1, 2 + 151), dtype=np.uint8)      # (c3: scratch space is tight, FASTQ only)
fa[:, 0:2] = np.frombuffer(b">\n", dtype=np.uint8)
fa[:, 2:] = arr[:len(fa)]
fa.tofile(os.path.join(work, "reads.fa"))
print("wrote reads (fq %.1f GB, fa %.1f GB) in %.1f s" % (fq.nbytes / 1e9, fa.nbytes / 1e9, time.time() - t), flush=True)
del fq, fa, arr, lines
cli = os.environ.get("QM2_E2E_CLI") or os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")
import re
runs = [(f, t, e) for f in ("reads.fq", "reads.fa")
        for t, e in (("4", {}), ("8", {}), ("12", {}), ("8", {"QM2_READER_RAW": "0"}), ("4", {"QM2_READER_RAW": "0", "QM2_READER_STREAM": "1"}))]
reps = 4
if scale == "c3":
    runs, reps = [("reads.fq", "8", {})], 2
for name, thr, extra in runs:
    best, walls = None, []
    for rep in range(reps):
        t = time.time()
        r = subprocess.run([cli, "count", "-t", thr, os.path.join(work, "ref.fa"), os.path.join(work, name), os.path.join(work, "out")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, QM2_TIMING="1", **extra))
        walls.append(time.time() - t)
        out = r.stdout.decode(errors="replace")
        m = re.search(r"QM2 phase count ([0-9.]+) s", out)
        sec = float(m.group(1)) if m else float("nan")
        best = sec if best is None else min(best, sec)
    print("== %s -t %s %s rc=%d  count phase best: %.3f s = %.2f G bases/s; wall best %.2f s (incl. .qm load, upload, epilogue)" %
          (name, thr, "host strip, read(2) stream" if "QM2_READER_STREAM" in extra else "host strip, pread blocks" if extra else "device line discipline", r.returncode, best, n_reads * 150 / best / 1e9, min(walls)), flush=True)
print(out)
