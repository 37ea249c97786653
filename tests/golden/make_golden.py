#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REAL REFERENCE.

This script is the provenance of everything under tests/golden/.  It needs
oracle/_ref/quicKmer2 (built by `make -C oracle ref` from
/root/reference/QuicKmer.c where it lies) and is therefore only runnable in
the build container; the vectors it writes are plain data (inputs and the
reference's outputs) and are what travels.

Layout written:
  kat/                5 kb reference, `search -k 30 -s 16K -e 0 -c ctrl.bed`
    ref.fa ref.fa.qm ref.fa.qgc ref.fa.bed ctrl.bed
    <case>.fa|.fq     reads for each known-answer case
    <case>.bin .txt   the reference's `count` outputs
  kat_k25/ kat_k31/ kat_k12/   same reference at other k (k != 30 quirk)
  mini/               60 kb reference, -s 128K; reads regenerated from a seed
    ref.fa.qm ref.fa.qgc reads.bin reads.txt (+ md5 of regenerated inputs)
  manifest.json       per case: k-mers reported by the reference, md5s, sizes
  c1_manifest.json    md5s of the full config-C1 run (inputs are regenerated
                      from numpy seed 20261003; too big to commit)
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "quicKmer2")
sys.path.insert(0, os.path.dirname(HERE))
from synth import c1_inputs, mini_inputs, kat_reference, kat_cases  # noqa: E402


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def run(args, cwd=None, stdin=None):
    p = subprocess.run(args, cwd=cwd, stdin=stdin, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, check=True)
    return p.stdout.decode(errors="replace")


def ref_search(fa, k, size, ctrl=None, extra=()):
    args = [REF_BIN, "search", "-k", str(k), "-s", size, "-e", "0", "-w", "1000"]
    if ctrl:
        args += ["-c", ctrl]
    args += list(extra) + [fa]
    return run(args)


def ref_count(fa, reads, out_prefix, threads=None):
    args = [REF_BIN, "count"]
    if threads is not None:
        args += ["-t", str(threads)]
    args += [fa, reads, out_prefix]
    out = run(args)
    m = re.search(r"total (\d+) kmers", out)
    mean = re.search(r"Mean sequencing depth: ([0-9.naninf-]+)", out)
    return int(m.group(1)), (mean.group(1) if mean else None), out


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("need oracle/_ref/quicKmer2 (make -C oracle ref)")
    manifest = {"generator": "tests/golden/make_golden.py",
                "reference_binary": "gcc QuicKmer.c -O3 -g -pthread -std=c99 -lm (reference makefile:4)",
                "sets": {}}

    ref = kat_reference()
    ctrl = b"chrK\t200\t2400\nchrK\t3000\t4800\n"
    for setname, k, size in (("kat", 30, "16K"), ("kat_k25", 25, "16K"),
                             ("kat_k31", 31, "16K"), ("kat_k12", 12, "16K")):
        d = os.path.join(HERE, setname)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        fa = os.path.join(d, "ref.fa")
        with open(fa, "wb") as f:
            f.write(b">chrK\n")
            s = ref.tobytes()
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + b"\n")
        with open(os.path.join(d, "ctrl.bed"), "wb") as f:
            f.write(ctrl)
        ref_search(fa, k, size, ctrl=os.path.join(d, "ctrl.bed"))
        entry = {"k": k, "qm_md5": md5(fa + ".qm"), "qgc_md5": md5(fa + ".qgc"), "cases": {}}
        cases = kat_cases(ref)
        if setname != "kat":   # the other-k sets only need a few cases
            cases = {n: cases[n] for n in ("upper", "reads_fa", "polyA", "long70000", "short_reads")}
        for name, (kind, data) in cases.items():
            rp = os.path.join(d, name + "." + kind)
            with open(rp, "wb") as f:
                f.write(data)
            nk, mean, _ = ref_count(fa, rp, os.path.join(d, name))
            # threaded run must agree (reference property)
            with tempfile.TemporaryDirectory() as td:
                nk2, _, _ = ref_count(fa, rp, os.path.join(td, "t"), threads=2)
                assert nk2 == nk and md5(os.path.join(td, "t.bin")) == md5(os.path.join(d, name + ".bin"))
            entry["cases"][name] = {
                "reads": name + "." + kind, "n_kmers": nk, "mean_depth": mean,
                "bin_md5": md5(os.path.join(d, name + ".bin")),
                "txt_md5": md5(os.path.join(d, name + ".txt")),
            }
            if len(data) > 300_000:      # regenerable big inputs are not committed
                entry["cases"][name]["reads_md5"] = md5(rp)
                entry["cases"][name]["reads_regenerated"] = True
                os.remove(rp)
        manifest["sets"][setname] = entry

    # ---- mini: a scaled-down C1 (N gap, duplicated segment, 2 contigs) -------
    d = os.path.join(HERE, "mini")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    with tempfile.TemporaryDirectory() as td:
        fa, ctrlp, rfa, rfq = mini_inputs(td)
        ref_search(fa, 30, "128K", ctrl=ctrlp)
        nk, mean, _ = ref_count(fa, rfa, os.path.join(d, "reads"))
        nkq, _, _ = ref_count(fa, rfq, os.path.join(td, "q"), threads=3)
        assert nkq == nk and md5(os.path.join(td, "q.bin")) == md5(os.path.join(d, "reads.bin"))
        shutil.copy(fa + ".qm", os.path.join(d, "ref.fa.qm"))
        shutil.copy(fa + ".qgc", os.path.join(d, "ref.fa.qgc"))
        manifest["sets"]["mini"] = {
            "k": 30, "n_kmers": nk, "mean_depth": mean,
            "ref_md5": md5(fa), "reads_fa_md5": md5(rfa), "reads_fq_md5": md5(rfq),
            "qm_md5": md5(fa + ".qm"), "qgc_md5": md5(fa + ".qgc"), "bed_md5": md5(fa + ".bed"),
            "bin_md5": md5(os.path.join(d, "reads.bin")), "txt_md5": md5(os.path.join(d, "reads.txt")),
        }

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)

    # ---- full C1 -> md5s only ---------------------------
    with tempfile.TemporaryDirectory() as td:
        fa, ctrlp, rfa, rfq = c1_inputs(td)
        out = ref_search(fa, 30, "2M", ctrl=ctrlp)
        nk, mean, _ = ref_count(fa, rfa, os.path.join(td, "s"))
        nk4, _, _ = ref_count(fa, rfq, os.path.join(td, "t4"), threads=4)
        assert nk4 == nk and md5(os.path.join(td, "s.bin")) == md5(os.path.join(td, "t4.bin"))
        c1 = {"n_kmers": nk, "mean_depth": mean,
              "dict_kmers": os.path.getsize(os.path.join(td, "s.bin")) // 2,
              "ref_md5": md5(fa), "reads_fa_md5": md5(rfa), "reads_fq_md5": md5(rfq),
              "qm_md5": md5(fa + ".qm"), "qgc_md5": md5(fa + ".qgc"), "bed_md5": md5(fa + ".bed"),
              "bin_md5": md5(os.path.join(td, "s.bin")), "txt_md5": md5(os.path.join(td, "s.txt")),
              "search_stdout_tail": [l for l in out.splitlines()[-6:] if not l.startswith("ADDR")]}
    with open(os.path.join(HERE, "c1_manifest.json"), "w") as f:
        json.dump(c1, f, indent=1, sort_keys=True)
    print(json.dumps(c1, indent=1))


if __name__ == "__main__":
    main()
