"""`quicKmer2 est` (quick-mer2_amd/csrc/qm2_est.c) against beds written by the REAL
reference's `est` (tests/golden/est/, made by tests/golden/make_est_golden.py).  The smoothing
helper both sides call is the deterministic stand-in in tests/est_helper/ — `est` treats its 401
floats as an opaque input.  Host-only: no GPU, no device code involved."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

import oracle_lib as ol
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")
MKDICT = os.path.join(ROOT, "quick-mer2_amd", "qm2-mkdict")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "quicKmer2")
HELPER_DIR = os.path.join(ROOT, "tests", "est_helper")


def env_with_helper(extra_first=None):
    path = ([extra_first] if extra_first else []) + [HELPER_DIR, os.environ["PATH"]]
    return dict(os.environ, PATH=os.pathsep.join(path))


def md5_file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.fixture(scope="module")
def built():
    if not (os.path.exists(CLI) and os.path.exists(MKDICT)):
        import __graft_entry__ as g
        g.build()
    return CLI


def run_est(args, cwd, env):
    return subprocess.run([CLI, "est"] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.mark.parametrize("sample", ["reads_fa", "reads_fq", "polyA", "long70000", "wrap70000", "short_reads", "n_upper"])
def test_kat_beds_byte_identical(built, sample, tmp_path):
    for f in ("ref.fa.qgc", "ref.fa.bed", sample + ".bin", sample + ".txt"):
        shutil.copy(os.path.join(GOLD, "kat", f), tmp_path)
    r = run_est(["ref.fa", sample, "out.bed"], str(tmp_path), env_with_helper())
    assert r.returncode == 0, r.stdout
    assert open(tmp_path / "out.bed", "rb").read() == open(os.path.join(GOLD, "est", sample + ".bed"), "rb").read()
    assert r.stdout == open(os.path.join(GOLD, "est", sample + ".stdout"), "rb").read()


def test_c1_bed_and_the_repeated_last_window(built, tmp_path):
    """Full config C1 (989 windows over 989 471 k-mers = two 1 MiB blocks), dictionary by qm2-mkdict,
    counts by the oracle; then the bed cut to 100 windows: k-mers remain behind the last window
    in the second block, where the reference prints that window once more (QuicKmer.c:672-673
    leaves only the inner loop)."""
    man = json.load(open(os.path.join(GOLD, "est", "c1_manifest.json")))
    fa, ctrl, rfa, rfq = synth.c1_inputs(str(tmp_path))
    subprocess.run([MKDICT, "-k", "30", "-s", "2M", "-w", "1000", "-c", ctrl, fa], check=True, stdout=subprocess.DEVNULL)
    rc, st, _ = ol.count_main(fa, rfa, str(tmp_path / "s"))
    assert rc == 0
    r = run_est([fa, str(tmp_path / "s"), str(tmp_path / "s.bed")], str(tmp_path), env_with_helper())
    assert r.returncode == 0, r.stdout
    assert r.stdout.decode() == man["c1_stdout"]
    assert md5_file(tmp_path / "s.bed") == man["c1_bed_md5"]
    assert sum(1 for _ in open(tmp_path / "s.bed")) == man["c1_windows"]
    if os.path.exists(REF_BIN):                     # live, where the reference build is present
        subprocess.run([REF_BIN, "est", fa, str(tmp_path / "s"), str(tmp_path / "ref.bed")], env=env_with_helper(),
                       check=True, stdout=subprocess.DEVNULL)
        assert open(tmp_path / "ref.bed", "rb").read() == open(tmp_path / "s.bed", "rb").read()
    lines = open(fa + ".bed").read().splitlines(True)
    open(fa + ".bed", "w").write("".join(lines[:100]))
    r = run_est([fa, str(tmp_path / "s"), str(tmp_path / "t.bed")], str(tmp_path), env_with_helper())
    assert r.returncode == 0, r.stdout
    assert md5_file(tmp_path / "t.bed") == man["c1_short_bed_md5"]
    assert sum(1 for _ in open(tmp_path / "t.bed")) == man["c1_short_bed_lines"] == 101


def test_failures_are_loud(built, tmp_path):
    for f in ("ref.fa.qgc", "ref.fa.bed", "reads_fa.bin", "reads_fa.txt"):
        shutil.copy(os.path.join(GOLD, "kat", f), tmp_path)
    cwd = str(tmp_path)
    # no helper on PATH: the reference would write a bed from an unread buffer; here it is an error
    bare = dict(os.environ, PATH="/usr/bin:/bin")
    r = run_est(["ref.fa", "reads_fa", "out.bed"], cwd, bare)
    assert r.returncode == 1 and b"GC correction helper failed" in r.stdout
    assert not os.path.exists(tmp_path / "out.bed")
    # a helper that prints too little
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "smooth_GC_mrsfast.py").write_text("#!/bin/sh\nprintf 'abcd'\n")
    os.chmod(bad / "smooth_GC_mrsfast.py", 0o755)
    r = run_est(["ref.fa", "reads_fa", "out.bed"], cwd, env_with_helper(str(bad)))
    assert r.returncode == 1 and b"got 1" in r.stdout
    # missing inputs: the reference's messages (QuicKmer.c:573, :581, :589)
    for victim, msg in (("ref.fa.qgc", b"GC control file missing."), ("ref.fa.bed", b"Window segmentation file missing."),
                        ("reads_fa.bin", b"Sample file missing (Use prefix).")):
        os.rename(tmp_path / victim, tmp_path / (victim + ".away"))
        r = run_est(["ref.fa", "reads_fa", "out.bed"], cwd, env_with_helper())
        assert r.returncode == 1 and msg in r.stdout, victim
        os.rename(tmp_path / (victim + ".away"), tmp_path / victim)
    os.remove(tmp_path / "reads_fa.txt")
    r = run_est(["ref.fa", "reads_fa", "out.bed"], cwd, env_with_helper())
    assert r.returncode == 1 and b".txt" in r.stdout
    r = subprocess.run([CLI, "est"], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"quicKmer2 est ref.fa sample_prefix output.bed" in r.stdout
