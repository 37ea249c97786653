"""tools/qm2_mkdict.c (the reference's `search -e 0` re-stated) must write
.qm / .qgc / .bed byte-identical to what the REAL reference wrote (md5s and files under
tests/golden/, produced by tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
MAN = json.load(open(os.path.join(GOLD, "manifest.json")))
EXE = os.path.join(ROOT, "quick-mer2_amd", "qm2-mkdict")


def md5_file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


@pytest.fixture(scope="module", autouse=True)
def built():
    src = os.path.join(ROOT, "tools", "qm2_mkdict.c")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-o", EXE, src, "-lm"], check=True)


@pytest.mark.parametrize("setname,k", [("kat", 30), ("kat_k25", 25), ("kat_k31", 31), ("kat_k12", 12)])
def test_kat_dictionaries_byte_identical(setname, k, tmp_path):
    d = os.path.join(GOLD, setname)
    shutil.copy(os.path.join(d, "ref.fa"), tmp_path / "ref.fa")
    shutil.copy(os.path.join(d, "ctrl.bed"), tmp_path / "ctrl.bed")
    subprocess.run([EXE, "-k", str(k), "-s", "16K", "-e", "0", "-w", "1000", "-c", str(tmp_path / "ctrl.bed"),
                    str(tmp_path / "ref.fa")], check=True, stdout=subprocess.DEVNULL)
    for ext in (".qm", ".qgc", ".bed"):
        assert open(str(tmp_path / "ref.fa") + ext, "rb").read() == open(os.path.join(d, "ref.fa" + ext), "rb").read(), ext


def test_mini_dictionary_byte_identical(tmp_path):
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    subprocess.run([EXE, "-k", "30", "-s", "128K", "-e", "0", "-w", "1000", "-c", ctrl, fa], check=True,
                   stdout=subprocess.DEVNULL)
    assert md5_file(fa + ".qm") == m["qm_md5"]
    assert md5_file(fa + ".qgc") == m["qgc_md5"]
    assert md5_file(fa + ".bed") == m["bed_md5"]
    assert open(fa + ".qm", "rb").read() == open(os.path.join(GOLD, "mini", "ref.fa.qm"), "rb").read()


def test_c1_dictionary_byte_identical(tmp_path):
    c1 = json.load(open(os.path.join(GOLD, "c1_manifest.json")))
    fa, ctrl, rfa, rfq = synth.c1_inputs(str(tmp_path))
    out = subprocess.run([EXE, "-k", "30", "-s", "2M", "-e", "0", "-w", "1000", "-c", ctrl, fa], check=True,
                         stdout=subprocess.PIPE).stdout.decode()
    assert "Total output %d k-mers" % c1["dict_kmers"] in out
    assert md5_file(fa + ".qm") == c1["qm_md5"]
    assert md5_file(fa + ".qgc") == c1["qgc_md5"]
    assert md5_file(fa + ".bed") == c1["bed_md5"]


def test_without_control_bed_and_refusals(tmp_path):
    d = os.path.join(GOLD, "kat")
    shutil.copy(os.path.join(d, "ref.fa"), tmp_path / "ref.fa")
    subprocess.run([EXE, "-s", "16K", "-e", "0", str(tmp_path / "ref.fa")], check=True, stdout=subprocess.DEVNULL)
    assert open(str(tmp_path / "ref.fa.qm"), "rb").read() == open(os.path.join(d, "ref.fa.qm"), "rb").read()
    assert not os.path.exists(str(tmp_path / "ref.fa.qgc"))           # no -c: no .qgc (QuicKmer.c:1276-1279)
    r = subprocess.run([EXE, "-s", "16K", "-e", "2", str(tmp_path / "ref.fa")], stdout=subprocess.PIPE)
    assert r.returncode != 0 and b"only -e 0" in r.stdout


@pytest.mark.parametrize("size", ["64K", "16K"])
def test_table_grows_like_the_reference(size, tmp_path):
    """A table that passes 80 % fill is doubled and re-placed in place (QuicKmer.c:891-895, Resize_hash_table
    :738-779): -s 64K on the mini reference doubles once, -s 16K three times; .qm / .qgc / .bed must be the bytes
    the real reference wrote (tests/golden/growth_manifest.json, made by make_growth_golden.py) — the order of the
    re-placement decides where colliding keys end up, so this pins it."""
    g = json.load(open(os.path.join(GOLD, "growth_manifest.json")))["cases"][size]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    out = subprocess.run([EXE, "-k", "30", "-s", size, "-e", "0", "-w", "1000", "-c", ctrl, fa], check=True,
                         stdout=subprocess.PIPE).stdout.decode()
    assert out.count("Resize ") == g["resizes"] and out.count("Allocation success") == g["resizes"]
    assert os.path.getsize(fa + ".qm") == g["qm_bytes"]
    assert md5_file(fa + ".qm") == g["qm_md5"]
    assert md5_file(fa + ".qgc") == g["qgc_md5"]
    assert md5_file(fa + ".bed") == g["bed_md5"]
