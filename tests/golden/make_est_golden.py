#!/usr/bin/env python3
"""Golden vectors for `est`: runs the REAL reference binary
(oracle/_ref/quicKmer2, built from /root/reference by oracle/Makefile) as
`quicKmer2 est ref.fa sample out.bed` on the committed KAT files, with the test
stand-in for the smoothing helper on PATH (tests/est_helper/), and stores the beds
and the stdout under tests/golden/est/.  Run from the repo root, in the build
container (the reference does not travel)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "quicKmer2")
KAT = os.path.join(ROOT, "tests", "golden", "kat")
OUT = os.path.join(ROOT, "tests", "golden", "est")
SAMPLES = ["reads_fa", "reads_fq", "polyA", "long70000", "wrap70000", "short_reads", "n_upper"]


def main():
    if not os.path.exists(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    env = dict(os.environ, PATH=os.path.join(ROOT, "tests", "est_helper") + os.pathsep + os.environ["PATH"])
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("ref.fa.qgc", "ref.fa.bed"):
            shutil.copy(os.path.join(KAT, f), tmp)
        for s in SAMPLES:
            for ext in (".bin", ".txt"):
                shutil.copy(os.path.join(KAT, s + ext), tmp)
            r = subprocess.run([REF, "est", "ref.fa", s, s + ".bed"], cwd=tmp, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, check=True)
            shutil.copy(os.path.join(tmp, s + ".bed"), os.path.join(OUT, s + ".bed"))
            with open(os.path.join(OUT, s + ".stdout"), "wb") as f:
                f.write(r.stdout)
            print(s, os.path.getsize(os.path.join(OUT, s + ".bed")), "bytes", r.stdout.decode().strip())
    # full config C1 (regenerable: only md5s are kept), plus a bed cut short so that k-mers remain
    # behind the last window in a later 1 MiB block (the reference then repeats that window)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import json
    import synth
    import make_golden as mg
    with tempfile.TemporaryDirectory() as tmp:
        fa, ctrl, rfa, rfq = synth.c1_inputs(tmp)
        mg.ref_search(fa, 30, "2M", ctrl)
        mg.ref_count(fa, rfa, os.path.join(tmp, "s"))
        man = {}
        r = subprocess.run([REF, "est", fa, os.path.join(tmp, "s"), os.path.join(tmp, "s.bed")], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
        man["c1_bed_md5"] = mg.md5(os.path.join(tmp, "s.bed"))
        man["c1_stdout"] = r.stdout.decode()
        man["c1_windows"] = sum(1 for _ in open(os.path.join(tmp, "s.bed")))
        lines = open(fa + ".bed").read().splitlines(True)
        open(fa + ".bed", "w").write("".join(lines[:100]))
        subprocess.run([REF, "est", fa, os.path.join(tmp, "s"), os.path.join(tmp, "t.bed")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
        man["c1_short_bed_md5"] = mg.md5(os.path.join(tmp, "t.bed"))
        man["c1_short_bed_lines"] = sum(1 for _ in open(os.path.join(tmp, "t.bed")))
        json.dump(man, open(os.path.join(OUT, "c1_manifest.json"), "w"), indent=1)
        print(man)


if __name__ == "__main__":
    sys.exit(main())
