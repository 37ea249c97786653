"""The oracle (oracle/qm2_oracle.c) against the vectors the REAL reference
produced (tests/golden/, made by tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAN = json.load(open(os.path.join(GOLD, "manifest.json")))


def md5_file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def _cases():
    out = []
    for s, e in MAN["sets"].items():
        for name in e.get("cases", {}):
            out.append((s, name))
    return out


_REGEN = {}


def reads_path(setname, name, tmp):
    c = MAN["sets"][setname]["cases"][name]
    p = os.path.join(GOLD, setname, c["reads"])
    if os.path.exists(p):
        return p
    if not _REGEN:
        _REGEN.update(synth.kat_cases(synth.kat_reference()))
    kind, data = _REGEN[name]
    assert hashlib.md5(data).hexdigest() == c["reads_md5"]
    p = os.path.join(tmp, c["reads"])
    open(p, "wb").write(data)
    return p


@pytest.mark.parametrize("setname,name", _cases())
@pytest.mark.parametrize("threads", [0, 3])
def test_kat_against_reference_vectors(setname, name, threads, tmp_path):
    c = MAN["sets"][setname]["cases"][name]
    if threads and c["n_kmers"] > 100000 and setname != "kat":
        pytest.skip("threaded variant covered on the k=30 set")
    rp = reads_path(setname, name, str(tmp_path))
    out = str(tmp_path / "o")
    rc, st, _ = ol.count_main(os.path.join(GOLD, setname, "ref.fa"), rp, out, threads)
    assert rc == 0
    assert st["n_kmers"] == c["n_kmers"]          # "total %lu kmers", QuicKmer.c:481
    assert md5_file(out + ".bin") == c["bin_md5"]
    assert md5_file(out + ".txt") == c["txt_md5"]
    # and byte-for-byte where the vector itself is committed
    exp = os.path.join(GOLD, setname, name + ".bin")
    assert open(out + ".bin", "rb").read() == open(exp, "rb").read()


def test_mini_against_reference_vectors(tmp_path):
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    assert md5_file(fa) == m["ref_md5"] and md5_file(rfa) == m["reads_fa_md5"]
    assert md5_file(rfq) == m["reads_fq_md5"]
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    for reads, thr in ((rfa, 0), (rfq, 2), (rfq, 0)):
        out = str(tmp_path / ("o%d" % thr))
        rc, st, _ = ol.count_main(ref_prefix, reads, out, thr)
        assert rc == 0 and st["n_kmers"] == m["n_kmers"]
        assert md5_file(out + ".bin") == m["bin_md5"]
        assert md5_file(out + ".txt") == m["txt_md5"]
        assert open(out + ".bin", "rb").read() == open(os.path.join(GOLD, "mini", "reads.bin"), "rb").read()


def test_known_answers():
    # wrap mod 65536: 70 000 copies of one dictionary 30-mer -> 4464
    qm = ol.load_qm(os.path.join(GOLD, "kat", "ref.fa.qm"))
    ref = synth.kat_reference()
    one = ref[1000:1030].tobytes()
    depth, st = ol.count_lines(qm["keys"], 30, (one + b"\n") * 70000)
    assert st["n_kmers"] == 70000 and st["n_hits"] == 70000
    assert int(depth.sum()) == 4464 and int(depth.max()) == 4464 and int((depth != 0).sum()) == 1
    # DJB of 0 and of a known value, by the published recurrence
    L = ol.lib()
    h = 5381
    for _ in range(8):
        h = (h * 33) & (2**64 - 1)
    assert L.qm2o_djb(0) == h
    x = 0x0123456789ABCDEF
    h = 5381
    for b in x.to_bytes(8, "little"):
        h = (h * 33 + b) & (2**64 - 1)
    assert L.qm2o_djb(x) == h


def test_every_dictionary_key_is_found_once():
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    occ = np.nonzero(keys)[0]
    # feed each key back as a 30-base line
    def dec(v):
        s = bytearray(30)
        for i in range(30):
            s[29 - i] = b"ACTG"[(int(v) >> (2 * i)) & 3]
        return bytes(s)
    lines = b"".join(dec(keys[i]) + b"\n" for i in occ[:5000])
    depth, st = ol.count_lines(keys, 30, lines)
    assert st["n_hits"] == 5000
    assert np.array_equal(np.nonzero(depth)[0], np.sort(occ[:5000]))
    assert depth.max() == 1


def test_strip_reads_pipe_loses_first_line(tmp_path):
    p = tmp_path / "r.fa"
    p.write_bytes(b"ACGTACGT\n>h\nTTTT\n")
    assert ol.strip_reads(str(p), seekable=True) == b"ACGTACGT\nTTTT\n"
    assert ol.strip_reads(str(p), seekable=False) == b"TTTT\n"      # QuicKmer.c:396 on a pipe
    q = tmp_path / "r.fq"
    q.write_bytes(b"@a\nACGT\n+\n@III\n@b\nGGCC\n+\n>III\n")
    assert ol.strip_reads(str(q), seekable=False) == b"ACGT\nGGCC\n"


@pytest.mark.skipif(not os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "quicKmer2")),
                    reason="real reference binary not built")
def test_c1_full_against_live_reference(tmp_path):
    """Config C1 end to end: oracle vs md5s recorded from (and re-run on) the real binary."""
    import subprocess
    c1 = json.load(open(os.path.join(GOLD, "c1_manifest.json")))
    fa, ctrl, rfa, rfq = synth.c1_inputs(str(tmp_path))
    assert md5_file(fa) == c1["ref_md5"] and md5_file(rfa) == c1["reads_fa_md5"]
    ref = os.path.join(ol.ORACLE_DIR, "_ref", "quicKmer2")
    subprocess.run([ref, "search", "-k", "30", "-s", "2M", "-e", "0", "-w", "1000", "-c", ctrl, fa],
                   check=True, stdout=subprocess.DEVNULL)
    assert md5_file(fa + ".qm") == c1["qm_md5"]
    for reads, thr in ((rfa, 0), (rfq, 4)):
        out = str(tmp_path / ("c%d" % thr))
        rc, st, _ = ol.count_main(fa, reads, out, thr)
        assert rc == 0 and st["n_kmers"] == c1["n_kmers"]
        assert md5_file(out + ".bin") == c1["bin_md5"]
        assert md5_file(out + ".txt") == c1["txt_md5"]


def test_threaded_drain_padding_lands_on_zero_slot():
    """The threaded model differs from the serial one only by the zero padding of the last
    batch, which bumps the (off-chain) empty slot k-mer 0 resolves to (QuicKmer.c:460-466)."""
    qm = ol.load_qm(os.path.join(GOLD, "kat", "ref.fa.qm"))
    ref = synth.kat_reference().tobytes()
    lines = (ref[100:300] + b"\n") * 37
    d0, s0 = ol.count_lines(qm["keys"], 30, lines, threads=0)
    d3, s3 = ol.count_lines(qm["keys"], 30, lines, threads=3)
    assert s0["n_kmers"] == s3["n_kmers"]
    assert not np.array_equal(d0, d3)
    assert np.array_equal(ol.remove_drain_padding(d3, qm["keys"], s3["n_kmers"]), d0)
    assert qm["keys"][ol.zero_slot(qm["keys"])] == 0
