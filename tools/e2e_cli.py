#!/usr/bin/env python3
"""End-to-end timing of the `quicKmer2 count` CLI on the GPU box: a synthetic dictionary written as a
real .qm (chain = occupied slots in ascending order: a valid cycle, enough for timing), N x 150 bp
reads as FASTQ (and, at chr22 scale, FASTA) on local disk or /dev/shm, then the CLI with wall-clock
per phase from its own output (QM2_TIMING=1).  This is the PCIe- and host-inclusive number (never
bench.py's `value`).

  tools/e2e_cli.py [n_reads] [workdir] [c2|c3] [reads_dir]

Checks on the way: the reference's progress lines (`Read %liG kmers`, QuicKmer.c:446) number
n_kmers >> 30, and no allocation happens inside the count phase (qm2_prepare)."""
import os
import re
import hashlib
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
scale = sys.argv[3] if len(sys.argv) > 3 else "c2"
reads_dir = sys.argv[4] if len(sys.argv) > 4 else work
os.makedirs(work, exist_ok=True)
os.makedirs(reads_dir, exist_ok=True)
n, H, k = (3_100_000_000, 1 << 32, 30) if scale == "c3" else (50_818_468, 1 << 28, 30)
import shutil
# QM2_E2E_FILES_NEAR=1: this script (which WRITES the test files, i.e. decides on which NUMA node their page-cache pages lie) runs on the
# CPUs of the GPU's NUMA node, where the CLI's reader threads run too (qm2_bind_near_device); 0 / unset: wherever the scheduler puts it
# "far": on the CPUs of the OTHER node(s) — the case the reader's look at the file's pages (qm2_reader.c, file_pages_cpus) is for
if os.environ.get("QM2_E2E_FILES_NEAR") == "1":
    print("files written from the GPU's NUMA node: %d CPUs" % pyqm2.bind_near_device(0), flush=True)
elif os.environ.get("QM2_E2E_FILES_NEAR") == "far":
    all_cpus = os.sched_getaffinity(0)
    pyqm2.bind_near_device(0)
    far = all_cpus - os.sched_getaffinity(0)
    if far:
        os.sched_setaffinity(0, far)
        print("files written from the node(s) the GPU does NOT hang off: %d CPUs" % len(far), flush=True)
    else:
        os.sched_setaffinity(0, all_cpus)
        print("one NUMA node only: files written from wherever", flush=True)
print("scratch free: %.0f GB in %s, %.0f GB in %s" % (shutil.disk_usage(work).free / 1e9, work,
                                                       shutil.disk_usage(reads_dir).free / 1e9, reads_dir), flush=True)
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
if os.environ.get("QM2_E2E_QGC", "1") != "0":
    # a .qgc like `search` writes (one u16 per dictionary k-mer in chain order: GC bin 0..400, bit 15 = control region;
    # QuicKmer.c:1023-1046): the epilogue's GC curve (.txt, :501-542) then has its real amount of work
    rng = np.random.default_rng(7)
    n_dict = len(occ)
    g = rng.integers(120, 260, n_dict, dtype=np.uint16)
    g |= (rng.integers(0, 10, n_dict, dtype=np.uint16) < 3).astype(np.uint16) << 15
    g.tofile(os.path.join(work, "ref.fa.qgc"))
    print("wrote .qgc %.1f GB (30 %% control k-mers)" % (g.nbytes / 1e9), flush=True)
    del g
del hk, chain, keys
torch.cuda.empty_cache()
# reads: made and framed on the device a piece at a time ("@r\n" seq "\n" "+\n" qual "\n" = 307 bytes a record)
t = time.time()
piece = 10_000_000
want_fa = scale != "c3"
head = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device="cuda")
plus = torch.tensor(list(b"+\n"), dtype=torch.uint8, device="cuda")
with open(os.path.join(reads_dir, "reads.fq"), "wb") as fq, open(os.path.join(reads_dir, "reads.fa"), "wb") as fa:
    for first in range(0, n_reads, piece):
        m = min(piece, n_reads - first)
        reads = torch.empty(m * 151, dtype=torch.uint8, device="cuda")
        pyqm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, m, first, 150, 150, 22, sample_from=covered)
        torch.cuda.synchronize()
        rec = torch.empty((m, 307), dtype=torch.uint8, device="cuda")
        rec[:, 0:3] = head
        rec[:, 3:154] = reads.view(m, 151)
        rec[:, 154:156] = plus
        rec[:, 156:306] = ord("I")
        rec[:, 306] = 10
        rec.cpu().numpy().tofile(fq)
        if want_fa:
            rec = torch.empty((m, 153), dtype=torch.uint8, device="cuda")
            rec[:, 0] = ord(">")
            rec[:, 1] = 10
            rec[:, 2:] = reads.view(m, 151)
            rec.cpu().numpy().tofile(fa)
        del reads, rec
print("wrote %d reads (fq %.1f GB) in %.1f s" % (n_reads, os.path.getsize(os.path.join(reads_dir, "reads.fq")) / 1e9, time.time() - t), flush=True)
del genome
torch.cuda.empty_cache()
# This process has just freed > 100 GB of device memory.  QM2_E2E_SETTLE=<seconds>: wait before the first CLI run, to tell the
# driver's clearing of that memory (which the next process's hipMalloc waits for) from the CLI's own load phase.
if os.environ.get("QM2_E2E_SETTLE"):
    time.sleep(float(os.environ["QM2_E2E_SETTLE"]))
# what a pipe carries on this host when nothing is done with the bytes: the ceiling of every `... | quicKmer2 count ref /dev/fd/0 out` below
for name in ("reads.fq",):
    t = time.time()
    subprocess.run("cat %s | cat > /dev/null" % os.path.join(reads_dir, name), shell=True, check=True)
    sec = time.time() - t
    print("pipe ceiling: `cat %s | cat > /dev/null` %.2f s = %.2f GB/s" % (name, sec, os.path.getsize(os.path.join(reads_dir, name)) / sec / 1e9), flush=True)
cli = os.environ.get("QM2_E2E_CLI") or os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")
runs = [(f, t, e) for f in ("reads.fq", "reads.fa")
        for t, e in (("8", {}), ("12", {}), ("16", {}), ("12", {"QM2_READER": "raw"}), ("8", {"QM2_READER": "lines"}), ("4", {"QM2_READER": "lines", "QM2_READER_STREAM": "1"}))]
reps = 4
if scale == "c3":
    runs, reps = [("reads.fq", "12", {}), ("reads.fq", "16", {}), ("reads.fq", "12", {"QM2_READER": "raw"})], 2
# QM2_E2E_RUNS="fq:12,fq:24,fq:12:0+0,fa:8::lines,fa:12:s0,fq:12::raw": file : threads [: devices joined by '+' (s0+1 = --shards 0,1)]
# [: packed | raw | lines (QM2_READER; "host" = lines)] [: pipe (the input through `cat file |` and /dev/fd/0: README.md:86-94's usage)]
if os.environ.get("QM2_E2E_RUNS"):
    runs = []
    for item in os.environ["QM2_E2E_RUNS"].split(","):
        f = item.split(":") + ["", "", "", ""]
        extra = {"QM2_READER": {"host": "lines"}.get(f[3], f[3])} if f[3] else {}
        if f[2]:
            extra["_devices"] = f[2].replace("+", ",")
        if f[4] == "pipe":
            extra["_pipe"] = "1"
        runs.append(("reads." + f[0], f[1], extra))
reps = int(os.environ.get("QM2_E2E_REPS", reps))
ok = True
out_sums = {}
for name, thr, extra in runs:
    best, walls = None, []
    extra = dict(extra)
    devs = extra.pop("_devices", None)
    pipe = extra.pop("_pipe", None)
    for rep in range(reps):
        t = time.time()
        cat = subprocess.Popen(["cat", os.path.join(reads_dir, name)], stdout=subprocess.PIPE) if pipe else None
        r = subprocess.run([cli, "count"] + (["-t", thr] if thr else []) + ((["--shards", devs[1:]] if devs.startswith("s") else ["--devices", devs]) if devs else []) +
                           [os.path.join(work, "ref.fa"), "/dev/fd/0" if pipe else os.path.join(reads_dir, name), os.path.join(work, "out")],
                           stdin=cat.stdout if pipe else None,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, QM2_TIMING="1", **extra))
        if cat:
            cat.stdout.close()
            cat.wait()
        walls.append(time.time() - t)
        out = r.stdout.decode(errors="replace")
        m = re.search(r"QM2 phase count ([0-9.]+) s", out)
        sec = float(m.group(1)) if m else float("nan")
        best = sec if best is None else min(best, sec)
        ld = re.search(r"QM2 phase load-dictionary ([0-9.]+) s", out)
        ep = re.search(r"QM2 phase epilogue ([0-9.]+) s", out)
        print("   rep %d: wall %.2f s = load %s + count %.3f + epilogue %s" % (rep, walls[-1], ld.group(1) if ld else "?", sec, ep.group(1) if ep else "?"), flush=True)
    # the outputs of every arrangement of the same input have to be the same bytes (.bin: the depth per dictionary k-mer, .txt: the GC curve)
    sums = []
    for suffix in (".bin", ".txt"):
        h = hashlib.blake2b(digest_size=16)
        try:
            with open(os.path.join(work, "out" + suffix), "rb") as fh:
                for piece in iter(lambda: fh.read(1 << 26), b""):
                    h.update(piece)
            sums.append(h.hexdigest())
        except OSError:
            sums.append("missing")
    out_sums.setdefault(name, []).append(tuple(sums))
    nk = re.search(r"total (\d+) kmers", out)
    prog = re.findall(r"^Read (\d+)G kmers$", out, flags=re.M)
    al = re.search(r"allocations inside it: (\d+)", out)
    good = r.returncode == 0 and nk and [int(x) for x in prog] == list(range(1, (int(nk.group(1)) >> 30) + 1)) and ((al and int(al.group(1)) == 0) or devs)
    load = re.search(r"QM2 phase load-dictionary.*", out)
    ok = ok and bool(good)
    print("== %s -t %s %s%s rc=%d  count phase best: %.3f s = %.2f G bases/s; wall best %.2f s (incl. .qm load, upload, epilogue); "
          "progress lines %d, allocations in the count phase %s%s | %s" %
          (name + (" through a pipe" if pipe else ""), thr, ("--devices %s " % devs) if devs else "", "host strip, read(2) stream" if "QM2_READER_STREAM" in extra else
           {"lines": "host strip to ASCII lines, pread blocks", "raw": "raw text, device line discipline", "packed": "packed by the reader threads"}[extra.get("QM2_READER", "packed")],
           r.returncode, best, n_reads * 150 / best / 1e9, min(walls), len(prog), al.group(1) if al else "?", "" if good else "  <-- CHECK FAILED",
           load.group(0) if load else ""), flush=True)
for name, got in out_sums.items():
    same = len(set(got)) == 1 and "missing" not in got[0]
    ok = ok and same
    print("== %s: .bin and .txt of the %d arrangements above are the same bytes: %s (blake2b-128 %s / %s)" % (name, len(got), same, got[0][0], got[0][1]), flush=True)
print(out)
for d_ in {work, reads_dir}:
    shutil.rmtree(d_, ignore_errors=True)
sys.exit(0 if ok else 1)
