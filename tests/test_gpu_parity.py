"""GPU parity tests: the HIP path (through the C-ABI, libqm2_hip.so) against the
CPU oracle on the same inputs and against the vectors the real reference
produced.  Bit-exact everywhere (integer / byte work).  Run with `-m gpu`."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
MAN = json.load(open(os.path.join(GOLD, "manifest.json")))
CLI = os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")


@pytest.fixture(scope="module")
def qm2():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))
    import pyqm2
    pyqm2.lib()
    return pyqm2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def md5_file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def kat_qm(setname="kat"):
    return ol.load_qm(os.path.join(GOLD, setname, "ref.fa.qm"))


def check_against_oracle(qm2, keys, k, lines, partition_available=True, **kw):
    """Both pipelines (direct kernel; partitioned: radix split + LDS probe/count) vs the oracle, and both forms the text
    crosses the link in: packed by qm2_submit on the way into the pinned buffers (its default since ABI 6: the kernels'
    PK instantiations read 2-bit codes + reset flags) and as the caller's ASCII (QM2_FLAG_SUBMIT_ASCII)."""
    exp_depth, exp_st = ol.count_lines(keys, k, lines)
    for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION):
        for form in (0, qm2.FLAG_SUBMIT_ASCII):
            with qm2.Counter(keys, k=k, flags=flags | form | qm2.FLAG_TIMING, **kw) as c:
                c.submit(lines)
                depth, st = c.finish()
                prof = c.profile()
            assert st == exp_st, (flags, form)
            assert np.array_equal(depth, exp_depth), (flags, form)
            if len(lines) > 64:
                if flags == qm2.FLAG_PARTITION and partition_available:
                    assert prof["partitioned_batches"] > 0 and prof["direct_batches"] == 0
                    assert prof["class_launches"]["bucket_count"] > 0
                elif flags == qm2.FLAG_DIRECT or not partition_available:
                    assert prof["partitioned_batches"] == 0 and prof["class_launches"]["count_direct"] > 0
    return st


# ---- 1. the reference's own vectors, end to end through the CLI ---------------

def _cases():
    return [(s, n) for s, e in MAN["sets"].items() for n in e.get("cases", {})]


_REGEN = {}


def _reads_path(setname, name, tmp):
    c = MAN["sets"][setname]["cases"][name]
    p = os.path.join(GOLD, setname, c["reads"])
    if os.path.exists(p):
        return p
    if not _REGEN:
        _REGEN.update(synth.kat_cases(synth.kat_reference()))
    p = os.path.join(tmp, c["reads"])
    open(p, "wb").write(_REGEN[name][1])
    assert md5_file(p) == c["reads_md5"]
    return p


@pytest.mark.parametrize("setname,name", _cases())
def test_cli_matches_reference_vectors(setname, name, tmp_path):
    c = MAN["sets"][setname]["cases"][name]
    rp = _reads_path(setname, name, str(tmp_path))
    out = str(tmp_path / "o")
    r = subprocess.run([CLI, "count", "-t", "2", os.path.join(GOLD, setname, "ref.fa"), rp, out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    txt = r.stdout.decode(errors="replace")
    assert r.returncode == 0, txt
    assert "total %d kmers" % c["n_kmers"] in txt          # QuicKmer.c:481
    assert open(out + ".bin", "rb").read() == open(os.path.join(GOLD, setname, name + ".bin"), "rb").read()
    assert md5_file(out + ".txt") == c["txt_md5"]
    if c["mean_depth"] is not None:
        assert "Mean sequencing depth: %s" % c["mean_depth"] in txt
    assert txt.rstrip().endswith("Exit quicK-mer2 count")


@pytest.mark.parametrize("mode", ["packed", "raw", "lines"])
def test_cli_raw_and_host_line_discipline(mode, tmp_path):
    """The three ways the CLI's input reaches the device (QM2_READER): packed (the default: the reader threads strip and
    pack, 0.375 bytes per base over the link), raw (regular files: the host only preads, the device does the line
    discipline), lines (stripped to ASCII on the host).  Every k = 30 reference vector must come
    out byte-identical all three ways; an input the device line discipline refuses must silently go round again through
    the host path."""
    refused = []
    raw = "1" if mode == "raw" else "0"
    env = dict(os.environ, QM2_READER=mode, QM2_TIMING="1")
    for name, c in MAN["sets"]["kat"]["cases"].items():
        rp = _reads_path("kat", name, str(tmp_path))
        out = str(tmp_path / ("raw_" + name))
        r = subprocess.run([CLI, "count", "-t", "3", os.path.join(GOLD, "kat", "ref.fa"), rp, out], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        txt = r.stdout.decode(errors="replace")
        assert r.returncode == 0, (name, txt)
        assert "total %d kmers" % c["n_kmers"] in txt, name
        assert open(out + ".bin", "rb").read() == open(os.path.join(GOLD, "kat", name + ".bin"), "rb").read(), name
        assert md5_file(out + ".txt") == c["txt_md5"], name
        assert {"packed": "input packed on the host", "raw": "input raw text", "lines": "input ASCII lines"}[mode] in txt, txt
        if "raw input refused" in txt:
            refused.append(name)
    assert not refused, refused             # long lines included: the device keeps its own wrap list
    # a '>' at a FASTQ sequence position (QuicKmer.c:398 skips that line without the three discards) is followed
    # on the device; 70 of them in one batch are refused, and the CLI must go round again through the host path
    ref = synth.kat_reference().tobytes()
    for tag, n_gt, expect_refusal in (("few", 3, False), ("many", 70, raw == "1")):
        # n_gt consecutive '>' lines where record 700's sequence should be: each is at a sequence position (the line
        # after a skipped one is), then the records go on normally
        recs = [b"@r%d\n" % i + (b">oops\n" * n_gt if i == 700 else b"") + ref[i:i + 120] + b"\n+\n" + b"I" * 120 + b"\n"
                for i in range(2000)]
        p = tmp_path / ("desync_%s.fq" % tag)
        p.write_bytes(b"".join(recs))
        out = str(tmp_path / ("desync_" + tag))
        r = subprocess.run([CLI, "count", "-t", "3", os.path.join(GOLD, "kat", "ref.fa"), str(p), out], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0
        assert (b"raw input refused" in r.stdout) == expect_refusal, tag
        exp_depth, exp_st = ol.count_lines(kat_qm()["keys"], 30, ol.strip_reads(str(p)))
        assert ("total %d kmers" % exp_st["n_kmers"]).encode() in r.stdout
        order = ol.chain_order(kat_qm())                              # .bin = depth in chain order (QuicKmer.c:494-516)
        assert open(out + ".bin", "rb").read() == exp_depth[order].astype("<u2").tobytes(), tag
    # several device batches of raw FASTQ (1 MiB batches would need the API; the CLI's are >= 1 GiB): mini set
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    for tag, reads in (("fa", rfa), ("fq", rfq)):
        out = str(tmp_path / ("rawmini_" + tag))
        r = subprocess.run([CLI, "count", "-t", "8", os.path.join(GOLD, "mini", "ref.fa"), reads, out], env=env,
                           stdout=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and b"raw input refused" not in r.stdout
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]


def test_cli_mini_fasta_fastq_and_pipe(tmp_path):
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    for tag, reads in (("fa", rfa), ("fq", rfq)):
        out = str(tmp_path / tag)
        r = subprocess.run([CLI, "count", ref_prefix, reads, out], stdout=subprocess.PIPE, timeout=300)
        assert r.returncode == 0
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]
        assert ("total %d kmers" % m["n_kmers"]).encode() in r.stdout
    # stdin pipe, the README's usage (samtools ... | quicKmer2 count ref /dev/fd/0 out)
    # ... through a REAL pipe (a regular file on stdin is seekable: that is the path above), FASTQ and FASTA (whose
    # first line the reference loses on a pipe, QuicKmer.c:396: the rewind fails), packed by the reader threads and as lines
    for tag, reads in (("fq", rfq), ("fa", rfa)):
        qm = ol.load_qm(ref_prefix + ".qm")
        exp_depth, exp_st = ol.count_lines(qm["keys"], 30, ol.strip_reads(reads, seekable=False))
        exp_bin = exp_depth[ol.chain_order(qm)].astype("<u2").tobytes()
        for mode in ("packed", "lines"):
            out = str(tmp_path / ("pipe_%s_%s" % (tag, mode)))
            cat = subprocess.Popen(["cat", reads], stdout=subprocess.PIPE)
            r = subprocess.run([CLI, "count", "-t", "4", ref_prefix, "/dev/fd/0", out], stdin=cat.stdout, env=dict(os.environ, QM2_READER=mode, QM2_TIMING="1"),
                               stdout=subprocess.PIPE, timeout=300)
            cat.stdout.close()
            assert cat.wait() == 0 and r.returncode == 0, r.stdout.decode(errors="replace")
            assert ("total %d kmers" % exp_st["n_kmers"]).encode() in r.stdout
            assert open(out + ".bin", "rb").read() == exp_bin, (tag, mode)
            if tag == "fq":
                assert md5_file(out + ".bin") == m["bin_md5"]
            assert (b"input packed on the host" in r.stdout) == (mode == "packed")


def test_cli_devices_two_contexts_sum_to_the_reference(tmp_path):
    """`quicKmer2 count --devices 0,0`: two replicas of the dictionary (here on the one GPU of the box), one
    reader, the input dealt to the contexts in turns (small ones, so that all of them really count), the
    counters summed into the first before the epilogue: .bin / .txt are the reference's own, stdout too.
    By default the reader threads pack what they strip and the pieces are dealt in turns of positions; QM2_READER=raw
    sends regular files to the contexts as RAW blocks (line discipline on the devices, turns of consecutive blocks);
    `-t 0` and QM2_READER=lines strip on the host and deal bytes."""
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    small = dict(os.environ, QM2_TURN_BYTES="100000", QM2_BLOCK_BYTES="16384", QM2_TURN_BLOCKS="3", QM2_TIMING="1")
    for tag, reads, devs, thr, env in (("fa", rfa, "0,0", "2", small), ("fq", rfq, "0,0,0", "0", small), ("one", rfq, "0", "4", small),
                                       ("fq_packed", rfq, "0,0,0", "3", small), ("fa_packed", rfa, "0,0", "4", dict(small, QM2_READER="packed")),
                                       ("fq_raw", rfq, "0,0,0", "3", dict(small, QM2_READER="raw")), ("fa_raw", rfa, "0,0", "2", dict(small, QM2_READER="raw")),
                                       ("fq_host", rfq, "0,0", "3", dict(small, QM2_READER_RAW="0")),
                                       ("fq_default_turns", rfq, "0,0", "2", dict(os.environ, QM2_TIMING="1")),
                                       # (the replicas of distinct GPUs are loaded by one thread each: forced here on the one GPU)
                                       ("fq_threaded_loaders", rfq, "0,0,0", "3", dict(small, QM2_LOAD_THREADS="1")),
                                       ("fa_serial_loaders", rfa, "0,0", "2", dict(small, QM2_LOAD_THREADS="0"))):
        out = str(tmp_path / tag)
        r = subprocess.run([CLI, "count", "-t", thr, "--devices", devs, ref_prefix, reads, out], stdout=subprocess.PIPE,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stdout
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]
        text = r.stdout.decode()
        assert ("total %d kmers" % m["n_kmers"]) in text and text.count("Read 0x") == 1
        assert "raw input refused" not in text
        assert text.rstrip().endswith("Exit quicK-mer2 count")
    # a '>' line at a FASTQ sequence position (QuicKmer.c:398: skipped WITHOUT the three discards) shifts every later
    # sequence line of the text; a replica that did not see that batch cannot know, so the contexts refuse it
    # (QM2_FLAG_RAW_STRICT) and the CLI goes round again through the host line discipline: still the reference's counts
    ref = synth.kat_reference().tobytes()
    recs = [b"@r%d\n" % i + (b">oops\n" if i == 900 else b"") + ref[i:i + 120] + b"\n+\n" + b"I" * 120 + b"\n" for i in range(3000)]
    p = tmp_path / "desync.fq"
    p.write_bytes(b"".join(recs))
    out = str(tmp_path / "desync")
    exp_depth, exp_st = ol.count_lines(kat_qm()["keys"], 30, ol.strip_reads(str(p)))
    for mode in ("raw", "packed"):      # (packed: the reader sees the line itself and finishes the file serially)
        r = subprocess.run([CLI, "count", "-t", "3", "--devices", "0,0", os.path.join(GOLD, "kat", "ref.fa"), str(p), out], env=dict(small, QM2_READER=mode),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0, r.stdout
        assert (b"raw input refused" in r.stdout) == (mode == "raw")
        assert ("total %d kmers" % exp_st["n_kmers"]).encode() in r.stdout
        assert open(out + ".bin", "rb").read() == exp_depth[ol.chain_order(kat_qm())].astype("<u2").tobytes()
    # bad ordinals are refused, not read as device 0
    for devs in ("0,x", "-1", "0,,1x"):
        r = subprocess.run([CLI, "count", "--devices", devs, ref_prefix, rfq, str(tmp_path / "bad")], stdout=subprocess.PIPE, timeout=60)
        assert r.returncode != 0 and b"not a device ordinal" in r.stdout


def test_cli_shards_config4_from_the_c_host(tmp_path):
    """`quicKmer2 count --shards ...` (include/qm2_shards.h, libqm2_shards.so: hash-range shards, route -> exchange ->
    probe of the own span, fold into one table, epilogue) against the reference's vectors.  One rank runs the real path
    end to end (nothing to exchange).  Two and four LOGICAL ranks on the one GPU run everything but RCCL itself:
    QM2_SHARDS_FAKE_EXCHANGE=1 replaces the ncclSend / ncclRecv pairs by device-to-device copies; small rounds so that
    every rank is fed, poly-A reads so that a round overflows an owner's slabs and is cut and routed again."""
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    fake = dict(os.environ, QM2_SHARDS_FAKE_EXCHANGE="1", QM2_SHARDS_BATCH_BYTES="100000", QM2_TIMING="1")
    for tag, reads, devs, env in (("one_fa", rfa, "0", dict(os.environ, QM2_TIMING="1")), ("one_fq_small_rounds", rfq, "0", fake),
                                  ("two", rfq, "0,0", fake), ("four", rfa, "0,0,0,0", fake)):
        out = str(tmp_path / ("sh_" + tag))
        r = subprocess.run([CLI, "count", "-t", "3", "--shards", devs, ref_prefix, reads, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=600, env=env)
        text = r.stdout.decode(errors="replace")
        assert r.returncode == 0, text
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"], tag
        assert ("total %d kmers" % m["n_kmers"]) in text and text.count("Read 0x") == 1
        assert text.rstrip().endswith("Exit quicK-mer2 count")
        if env is fake:
            rounds = int(text.split(" rounds,")[0].split()[-1])
            assert rounds >= 2, text                                  # several rounds, every rank fed
    # every KAT case through four logical shards of the 16 K-slot table (spans of 4 K slots + overhang)
    for name, c in MAN["sets"]["kat"]["cases"].items():
        rp = _reads_path("kat", name, str(tmp_path))
        out = str(tmp_path / ("shk_" + name))
        # 600000: a staging buffer holds the longest line (140 kb).  The long-line cases once more with batches of 40 kB: the line
        # arrives through qm2_shards_submit, is staged in 20 kB pieces that stay on one rank, and the rank's batch ends under
        # it, so it goes on behind a round with its last 32 bytes as look-back (qm2_route_device_carry) — around the uint16
        # run-counter wraps at base 65536 / 131072 as well
        for bb in ("600000", "40000") if name.startswith("long") else ("600000",):
            r = subprocess.run([CLI, "count", "-t", "2", "--shards", "0,0,0,0", os.path.join(GOLD, "kat", "ref.fa"), rp, out],
                               env=dict(fake, QM2_SHARDS_BATCH_BYTES=bb), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            assert r.returncode == 0, (name, bb, r.stdout)
            assert ("total %d kmers" % c["n_kmers"]).encode() in r.stdout, (name, bb)
            assert open(out + ".bin", "rb").read() == open(os.path.join(GOLD, "kat", name + ".bin"), "rb").read(), (name, bb)
            assert md5_file(out + ".txt") == c["txt_md5"], (name, bb)
    # skew: 4000 poly-A reads in front of the mini reads send whole rounds to the owner of k-mer 0's home slot
    qm = ol.load_qm(ref_prefix + ".qm")
    lines = (b"A" * 150 + b"\n") * 4000 + ol.strip_reads(rfq)
    p = tmp_path / "polya.fa"
    p.write_bytes(b">r\n" + lines.replace(b"\n", b"\n>r\n")[:-3])
    exp_depth, exp_st = ol.count_lines(qm["keys"], qm["k"], lines)
    out = str(tmp_path / "sh_polya")
    r = subprocess.run([CLI, "count", "-t", "2", "--shards", "0,0,0,0", ref_prefix, str(p), out], env=fake, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0, text
    assert ("total %d kmers" % exp_st["n_kmers"]) in text
    assert int(text.split(" cut and routed again")[0].split()[-1]) >= 1, text
    assert open(out + ".bin", "rb").read() == exp_depth[ol.chain_order(qm)].astype("<u2").tobytes()
    # more than one GPU over RCCL is experimental until a run on two cards is on record: refused without the opt-in ...
    r = subprocess.run([CLI, "count", "--shards", "0,0", ref_prefix, rfq, str(tmp_path / "dup")], stdout=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"experimental" in r.stdout
    # ... and with it, the same GPU twice is refused when the exchange is RCCL's
    r = subprocess.run([CLI, "count", "--shards", "0,0", ref_prefix, rfq, str(tmp_path / "dup")], stdout=subprocess.PIPE, timeout=120,
                       env=dict(os.environ, QM2_SHARDS_EXPERIMENTAL="1"))
    assert r.returncode != 0 and b"listed twice" in r.stdout


def test_cli_shards_ten_bit_route_and_flat(tmp_path):
    """Config 4's two exchange layouts from the C host (QM2_SHARDS_ROUTE): the senders' 10-bit route — G owners x 1024 / G
    sub-slabs in 8 per-XCD sub-regions, whole blocks over the wire, ONE split behind it (qm2_route_xcd_device ->
    qm2_probe_subslabs_device) — is the default from 2^23 slots on; `flat` is rounds 1-4's one slab per owner.  A dictionary
    of the mini reference in 2^23 slots (qm2-mkdict, the reference's `search -e 0` re-stated), 1 / 2 / 4 logical ranks on
    the one GPU (device-to-device copies in RCCL's place), small rounds; then a poly-A flood, which overflows the small
    sub-regions however the round is cut and goes round with one slab per owner: .bin = the oracle's counts every time."""
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    r = subprocess.run([os.path.join(ROOT, "quick-mer2_amd", "qm2-mkdict"), "-k", "30", "-s", "8M", "-e", "0", "-w", "1000", "-c", ctrl, fa],
                       stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout
    qm = ol.load_qm(fa + ".qm")
    assert qm["H"] == 1 << 23
    order = ol.chain_order(qm)
    lines = ol.strip_reads(rfq)
    exp_depth, exp_st = ol.count_lines(qm["keys"], 30, lines)
    exp_bin = exp_depth[order].astype("<u2").tobytes()
    fake = dict(os.environ, QM2_SHARDS_FAKE_EXCHANGE="1", QM2_SHARDS_BATCH_BYTES="200000", QM2_TIMING="1")
    for devs in ("0", "0,0", "0,0,0,0"):
        for mode in ("xcd", "flat", None):
            env = dict(fake) if mode is None else dict(fake, QM2_SHARDS_ROUTE=mode)
            out = str(tmp_path / ("r_%d_%s" % (len(devs), mode)))
            r = subprocess.run([CLI, "count", "-t", "3", "--shards", devs, fa, rfq, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
            text = r.stdout.decode(errors="replace")
            assert r.returncode == 0, text
            assert ("total %d kmers" % exp_st["n_kmers"]) in text
            assert open(out + ".bin", "rb").read() == exp_bin, (devs, mode)
            assert ("10-bit route" in text) == (mode != "flat"), text
            assert int(text.split(" rounds,")[0].split()[-1]) >= 2, text
    # the flood: 30000 poly-A reads in front (4.5 M copies of k-mer 0 for ONE sub-region of a few thousand places)
    flood = (b"A" * 150 + b"\n") * 30000 + lines
    p = tmp_path / "flood.fa"
    p.write_bytes(b">r\n" + flood.replace(b"\n", b"\n>r\n")[:-3])
    e2, s2 = ol.count_lines(qm["keys"], 30, flood)
    for devs in ("0", "0,0,0,0"):
        out = str(tmp_path / ("flood_%d" % len(devs)))
        r = subprocess.run([CLI, "count", "-t", "2", "--shards", devs, fa, str(p), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                           env=dict(fake, QM2_SHARDS_ROUTE="xcd"))
        text = r.stdout.decode(errors="replace")
        assert r.returncode == 0, text
        assert ("total %d kmers" % s2["n_kmers"]) in text
        assert open(out + ".bin", "rb").read() == e2[order].astype("<u2").tobytes(), devs
        assert int(text.split(" pieces fell back")[0].split()[-1]) >= 1, text
    # asked for on a table that is too small for it: said so
    r = subprocess.run([CLI, "count", "--shards", "0", os.path.join(GOLD, "mini", "ref.fa"), rfq, str(tmp_path / "small")], stdout=subprocess.PIPE, timeout=120,
                       env=dict(os.environ, QM2_SHARDS_ROUTE="xcd"))
    assert r.returncode != 0 and b"2^23" in r.stdout


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="needs two GPUs: RCCL between two ranks")
def test_cli_shards_on_two_physical_gpus(tmp_path):
    """Config 4 over the wire: `--shards 0,1` (ncclSend / ncclRecv between two GPUs) against the reference's vectors."""
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    env = dict(os.environ, QM2_SHARDS_BATCH_BYTES="300000", QM2_TIMING="1", QM2_SHARDS_EXPERIMENTAL="1")
    for tag, reads in (("fa", rfa), ("fq", rfq)):
        out = str(tmp_path / ("sh2_" + tag))
        r = subprocess.run([CLI, "count", "-t", "3", "--shards", "0,1", ref_prefix, reads, out], stdout=subprocess.PIPE, timeout=600, env=env)
        assert r.returncode == 0, r.stdout
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]
        assert b"GB exchanged" in r.stdout


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="needs two GPUs (the peer-copy branch of qm2_merge_counters)")
def test_cli_devices_on_two_physical_gpus(qm2, tmp_path):
    """The cross-device paths (never run before round 3: every lease so far had one GPU): `--devices 0,1` — a replica
    per GPU, raw blocks in turns, qm2_merge_counters through hipMemcpyPeerAsync — against the reference's vectors,
    and the merge called directly on two contexts of different devices."""
    m = MAN["sets"]["mini"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    env = dict(os.environ, QM2_BLOCK_BYTES="16384", QM2_TURN_BLOCKS="3")
    for tag, reads in (("fa", rfa), ("fq", rfq)):
        out = str(tmp_path / ("two_" + tag))
        r = subprocess.run([CLI, "count", "-t", "3", "--devices", "0,1", ref_prefix, reads, out], stdout=subprocess.PIPE, timeout=300, env=env)
        assert r.returncode == 0, r.stdout
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]
    qm = ol.load_qm(ref_prefix + ".qm")
    lines = ol.strip_reads(rfq)
    exp_depth, exp_st = ol.count_lines(qm["keys"], qm["k"], lines)
    half = lines.rfind(b"\n", 0, len(lines) // 2) + 1
    with qm2.Counter(qm["keys"], k=qm["k"], device=0) as a, qm2.Counter(qm["keys"], k=qm["k"], device=1) as b:
        a.submit(lines[:half])
        b.submit(lines[half:])
        a.merge_counters(b)
        depth, _ = a.finish()
    assert np.array_equal(depth, exp_depth)


@pytest.mark.skipif(not os.path.exists(os.path.join(ol.ORACLE_DIR, "_ref", "quicKmer2")),
                    reason="real reference binary not shipped")
def test_cli_epilogue_in_pieces_with_qgc(qm2, torch_cuda, tmp_path):
    """The epilogue's loop over the counts in chain order (.bin, and with a .qgc the GC curve of the .txt,
    QuicKmer.c:498-542) runs on worker threads that take 2^22-entry pieces; the reference vectors are all smaller
    than one piece.  Here: a 10 M-k-mer dictionary built on the device (3 pieces), a random .qgc — once whole, once
    cut short in the middle of a 2^20-entry block — and `.bin` / `.txt` / the mean-depth line compared with what the
    REFERENCE BINARY itself writes for the same files.  A short .qgc is not an error in the reference: fread's return
    value is ignored (:502-503), so behind the end of the file its gc_control[] keeps the values of the last reads,
    and those stale entries go on counting towards the curve; the CLI reproduces that (qm2_formats.c, stale_block)."""
    torch = torch_cuda
    n, H, k = 14_000_000, 1 << 26, 30
    genome = torch.empty(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=91, n_blocks=[(0, 100_000)])
    keys = torch.empty(H, dtype=torch.int64, device="cuda")
    scratch = torch.empty(H, dtype=torch.int64, device="cuda")
    qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    n_reads, rl = 400_000, 150
    reads = torch.empty(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 91)
    lines = reads.cpu().numpy().tobytes()
    hk = keys.cpu().numpy().view(np.uint64)
    del genome, keys, reads
    torch.cuda.empty_cache()
    occ = np.flatnonzero(hk).astype(np.uint32)
    assert len(occ) > 2 * (1 << 22)                                   # more than two pieces
    chain = np.zeros(H, dtype=np.uint32)
    chain[occ] = np.roll(occ, -1)
    ref = str(tmp_path / "ref.fa")
    with open(ref + ".qm", "wb") as f:
        f.write(b"QM11" + bytes([k, 0, 100, 100]) + int(H).to_bytes(8, "little") + int(occ[0]).to_bytes(8, "little"))
        hk.tofile(f)
        chain.tofile(f)
    fa = tmp_path / "reads.fa"
    fa.write_bytes(b">r\n" + lines.replace(b"\n", b"\n>r\n")[:-3])       # a header line before every read
    with qm2.Counter(hk, k=k) as c:
        c.submit(lines)
        depth, st = c.finish()
    exp_bin = depth[occ]                                              # chain order = ascending slots here
    rng = np.random.default_rng(4)
    g = rng.integers(0, 401, len(occ), dtype=np.uint16)               # (a bin beyond 400 makes the reference write past its arrays, :505)
    g |= (rng.integers(0, 4, len(occ), dtype=np.uint16) == 0).astype(np.uint16) << 15
    ref_bin = os.path.join(ol.ORACLE_DIR, "_ref", "quicKmer2")
    texts = []
    for cut in (0, 3_000_001):
        gc = g[:len(g) - cut]
        assert cut == 0 or (len(gc) >> 20 >= 1 and len(gc) % (1 << 20))   # ends inside a block, behind a full one
        gc.tofile(ref + ".qgc")
        out, rout = str(tmp_path / ("o%d" % cut)), str(tmp_path / ("r%d" % cut))
        r = subprocess.run([CLI, "count", "-t", "4", ref, str(fa), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        txt = r.stdout.decode(errors="replace")
        assert r.returncode == 0, txt
        assert "total %d kmers" % st["n_kmers"] in txt
        assert np.array_equal(np.fromfile(out + ".bin", dtype=np.uint16), exp_bin)
        rr = subprocess.run([ref_bin, "count", "-t", "6", ref, str(fa), rout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        rtxt = rr.stdout.decode(errors="replace")
        assert rr.returncode == 0, rtxt
        assert open(out + ".bin", "rb").read() == open(rout + ".bin", "rb").read(), cut
        assert open(out + ".txt").read() == open(rout + ".txt").read(), cut
        mean = [l for l in rtxt.splitlines() if l.startswith("Mean sequencing depth")]
        assert mean and mean[0] in txt, cut
        texts.append(open(out + ".txt").read())
    assert texts[0] != texts[1]                                       # the cut changed the curve, and both ways match the reference


def test_cli_c1_against_live_reference(tmp_path):
    """Config C1: dictionary by the real `search`, counts by the real `count`, vs the GPU CLI."""
    c1 = json.load(open(os.path.join(GOLD, "c1_manifest.json")))
    ref = os.path.join(ol.ORACLE_DIR, "_ref", "quicKmer2")
    fa, ctrl, rfa, rfq = synth.c1_inputs(str(tmp_path))
    subprocess.run([ref, "search", "-k", "30", "-s", "2M", "-e", "0", "-w", "1000", "-c", ctrl, fa],
                   check=True, stdout=subprocess.DEVNULL)
    assert md5_file(fa + ".qm") == c1["qm_md5"]
    subprocess.run([ref, "count", "-t", "4", fa, rfq, str(tmp_path / "ref")], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([CLI, "count", fa, rfq, str(tmp_path / "gpu")], stdout=subprocess.PIPE, timeout=600)
    assert r.returncode == 0
    assert open(str(tmp_path / "gpu.bin"), "rb").read() == open(str(tmp_path / "ref.bin"), "rb").read()
    assert md5_file(str(tmp_path / "gpu.bin")) == c1["bin_md5"]
    assert md5_file(str(tmp_path / "gpu.txt")) == c1["txt_md5"]
    assert ("total %d kmers" % c1["n_kmers"]).encode() in r.stdout


# ---- 2. C-ABI vs oracle: whole counter array and statistics --------------------

def test_api_mini_full_depth_and_stats(qm2, tmp_path):
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    lines = ol.strip_reads(rfq)
    st = check_against_oracle(qm2, qm["keys"], 30, lines)
    assert st["n_kmers"] == MAN["sets"]["mini"]["n_kmers"]


@pytest.mark.parametrize("staging", [4096, 5000, 65536, 1 << 20])
def test_batch_boundaries_do_not_matter(qm2, staging, tmp_path):
    """Result independent of staging size / split points, including mid-line splits."""
    qm = kat_qm()
    ref = synth.kat_reference()
    cases = synth.kat_cases(ref)
    blob = b""
    for name in ("reads_fa", "long70000", "long_with_N", "polyA", "long99998", "junk_bytes"):
        p = tmp_path / (name + ".fa")
        p.write_bytes(cases[name][1])
        blob += ol.strip_reads(str(p))
    exp_depth, exp_st = ol.count_lines(qm["keys"], 30, blob)
    for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION, qm2.FLAG_DIRECT | qm2.FLAG_SUBMIT_ASCII, qm2.FLAG_PARTITION | qm2.FLAG_SUBMIT_ASCII):
        # (packed, the default: the lines longer than a staging buffer's worth of positions go the ASCII way in between)
        for batch in (staging, 3 * staging + 17, 1 << 22):
            with qm2.Counter(qm["keys"], k=30, staging_bytes=staging, n_staging=2, batch_bytes=batch,
                             flags=flags) as c:
                # feed in pieces cut at line boundaries
                rng = np.random.default_rng(staging)
                nl = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 10) + 1
                cuts = sorted(set([0, len(blob)] + list(rng.choice(nl, size=7, replace=False))))
                for a, b in zip(cuts[:-1], cuts[1:]):
                    c.submit(blob[a:b])
                depth, st = c.finish()
            assert st == exp_st and np.array_equal(depth, exp_depth), (flags, batch)


def records_fit(k, hash_size):
    """The partitioned pipeline's finest level holds bucket records (qm2_device.hip.h): slot (12 bits), code (2) and the
    hash bits above log2 H (+ two k-mer bits at k = 31) share a 32-bit word; a table too small for that is counted by
    the direct kernel whatever the flags say."""
    lgH = int(hash_size).bit_length() - 1
    return 12 + 2 + (32 - lgH) + (2 if k == 31 else 0) <= 32


def rehash(keys, lg_h):
    """The keys of a dictionary in a table of 2^lg_h slots, placed by the reference's rule (home = DJB & (H - 1),
    walk towards the middle to the first empty slot, QuicKmer.c:90-99 / :846-860)."""
    H = 1 << lg_h
    out = np.zeros(H, dtype=np.uint64)
    for key in keys[keys != 0].tolist():
        h = 5381
        for i in range(8):
            h = (h * 33 + ((key >> (8 * i)) & 0xFF)) & 0xFFFFFFFFFFFFFFFF
        i = h & (H - 1)
        step = -1 if i & (H >> 1) else 1
        while out[i] != 0:
            i += step
        out[i] = key
    return out


@pytest.mark.parametrize("setname,k", [("kat_k12", 12), ("kat_k25", 25), ("kat_k31", 31)])
def test_other_k(qm2, setname, k, tmp_path):
    qm = kat_qm(setname)
    assert qm["k"] == k
    cases = synth.kat_cases(synth.kat_reference())
    tables = [qm["keys"]]
    if not records_fit(k, len(qm["keys"])):
        tables.append(rehash(qm["keys"], 16))      # k = 31: the partitioned pipeline from 2^16 slots on
        assert records_fit(k, len(tables[-1]))
    for keys in tables:
        for name in ("reads_fa", "long70000", "polyA", "short_reads", "long_with_N"):
            p = tmp_path / (name + ".fa")
            p.write_bytes(cases[name][1])
            check_against_oracle(qm2, keys, k, ol.strip_reads(str(p)), partition_available=records_fit(k, len(keys)),
                                 staging_bytes=30000)


def test_fuzz_arbitrary_bytes(qm2):
    """Any byte is a base (QuicKmer.c:411); 'N' and newline are the only resets."""
    qm = kat_qm()
    rng = np.random.default_rng(5)
    ref = synth.kat_reference().tobytes()
    parts = []
    for i in range(400):
        L = int(rng.integers(0, 400))
        kind = rng.integers(0, 4)
        if kind == 0:
            s = rng.integers(0, 256, L, dtype=np.uint8).tobytes().replace(b"\n", b"A")
        elif kind == 1:
            p = int(rng.integers(0, len(ref) - 400))
            s = bytearray(ref[p:p + L])
            for _ in range(int(rng.integers(0, 4))):
                if L:
                    s[int(rng.integers(0, L))] = int(rng.choice(list(b"NnXacgt\r\x00\xff")))
            s = bytes(s)
        elif kind == 2:
            s = bytes([int(rng.choice(list(b"ACGT")))]) * L
        else:
            p = int(rng.integers(0, len(ref) - 400))
            s = ref[p:p + L].lower()
        parts.append(s + b"\n")
    blob = b"".join(parts)
    for k, setname in ((30, "kat"), (25, "kat_k25")):
        check_against_oracle(qm2, kat_qm(setname)["keys"], k, blob, staging_bytes=8192)


def test_packed_entry_points(qm2, torch_cuda, tmp_path):
    """ABI 6, include/qm2.h "packed sequence": pieces packed by the caller (qm2_pack_lines) and committed through the pinned
    buffers (qm2_staging_commit_packed: any piece size, pieces collect in device batches, run-counter wraps of long lines as
    positions), the same blocks handed over in device memory (qm2_submit_packed_device), mixed with ASCII and raw entry
    points on one context — every time the oracle's counters and statistics."""
    torch = torch_cuda
    qm = kat_qm()
    cases = synth.kat_cases(synth.kat_reference())
    blob = b""
    for name in ("reads_fa", "long70000", "long_with_N", "polyA", "junk_bytes", "short_reads", "crlf"):
        p = tmp_path / (name + ".fa")
        p.write_bytes(cases[name][1])
        blob += ol.strip_reads(str(p))
    for k, setname in ((30, "kat"), (25, "kat_k25")):
        keys = kat_qm(setname)["keys"]
        exp_depth, exp_st = ol.count_lines(keys, k, blob)
        for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION):
            # through the pinned buffers, pieces of every size; batches smaller and larger than a piece
            for piece, batch in ((None, 1 << 22), (100_000, 1 << 18), (80_000, 3 << 20), (300_000, 1 << 19)):
                with qm2.Counter(keys, k=k, flags=flags, staging_bytes=1 << 20, n_staging=3, batch_bytes=batch) as c:
                    c.submit_packed(blob, piece=piece)
                    depth, st = c.finish()
                assert st == exp_st and np.array_equal(depth, exp_depth), (k, flags, piece, batch)
            # qm2_submit_threads: the library's own threads pack, the caller commits nothing itself
            for thr, staging in ((1, 1 << 20), (3, 200_000), (8, 1 << 21), (0, 1 << 20)):
                with qm2.Counter(keys, k=k, flags=flags, staging_bytes=staging, n_staging=6, batch_bytes=1 << 20) as c:
                    c.submit_threads(blob, thr)
                    cut = blob.find(b"\n", len(blob) // 2) + 1
                    c.submit_threads(blob[:cut], thr)
                    depth, st = c.finish()
                    assert c.profile()["staged_bytes"] > 0
                e2, s2 = ol.count_lines(keys, k, blob + blob[:cut])
                assert st == s2 and np.array_equal(depth, e2), (k, flags, thr, staging)
            # already on the device
            blocks, p, wraps = qm2.pack_lines(blob)
            assert p.n_wraps > 0
            d_blocks = torch.from_numpy(blocks.view(np.int64).copy()).cuda()
            d_wraps = torch.from_numpy(wraps.view(np.int64).copy()).cuda()
            with qm2.Counter(keys, k=k, flags=flags) as c:
                c.submit_packed_device(d_blocks.data_ptr(), p, d_wraps.data_ptr())
                depth, st = c.finish()
            assert st == exp_st and np.array_equal(depth, exp_depth), (k, flags)
            # the entry points mix on one context: packed pieces, ASCII lines, packed again, device text
            cut = blob.find(b"\n", len(blob) // 3) + 1
            cut2 = blob.find(b"\n", 2 * len(blob) // 3) + 1
            d_text = torch.frombuffer(bytearray(blob[cut2:]), dtype=torch.uint8).cuda()
            with qm2.Counter(keys, k=k, flags=flags | qm2.FLAG_SUBMIT_ASCII, staging_bytes=1 << 20, n_staging=3) as c:
                c.submit_packed(blob[:cut], piece=50_000)
                c.submit(blob[cut:cut2])
                c.submit_packed(blob[:cut], piece=70_000)
                c.sync()
                wr = ol_wraps(blob[cut2:])
                d_wr = torch.tensor(wr, dtype=torch.int64).cuda() if wr else None
                c.submit_device(d_text.data_ptr(), len(blob) - cut2, d_wr.data_ptr() if wr else None, len(wr))
                depth, st = c.finish()
            e2, s2 = ol.count_lines(keys, k, blob + blob[:cut])
            assert st == s2 and np.array_equal(depth, e2), (k, flags)
    # bad arguments fail loudly
    with qm2.Counter(qm["keys"], k=30) as c:
        pk = qm2.Packed(1, 65, 0)
        with pytest.raises(qm2.Qm2Error):
            c.submit_packed_device(d_blocks.data_ptr(), pk)
        with pytest.raises(qm2.Qm2Error):
            c.submit_packed_device(d_blocks.data_ptr() + 8, qm2.Packed(1, 0, 0))


def ol_wraps(lines: bytes):
    """the positions where the reference's uint16 run counter wraps (qm2_plan.h find_wraps), for qm2_submit_device"""
    import test_pack_cpu as tp
    return tp.model_wraps(lines)


def test_zero_copy_staging_and_cli_threads(qm2, tmp_path):
    """qm2_staging_acquire/commit (what the CLI's reader uses) gives the same counters as
    qm2_submit, and the CLI gives the same .bin for every -t (strip threads)."""
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    lines = ol.strip_reads(rfq)
    exp_depth, exp_st = ol.count_lines(qm["keys"], 30, lines)
    for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION):
        with qm2.Counter(qm["keys"], k=30, flags=flags, staging_bytes=100_000, n_staging=3) as c:
            c.submit_zero_copy(lines)
            c.submit(lines[:lines.find(b"\n", 5000) + 1])          # the two entry points mix
            depth, st = c.finish()
        e2, s2 = ol.count_lines(qm["keys"], 30, lines + lines[:lines.find(b"\n", 5000) + 1])
        assert st == s2 and np.array_equal(depth, e2)
    m = MAN["sets"]["mini"]
    ref_prefix = os.path.join(GOLD, "mini", "ref.fa")
    for t in ("0", "1", "8", "200"):
        out = str(tmp_path / ("t" + t))
        r = subprocess.run([CLI, "count", "-t", t, ref_prefix, rfq, out], stdout=subprocess.PIPE, timeout=300)
        assert r.returncode == 0 and ("[Option] Set %d threads" % (int(t) & 0xFF)).encode() in r.stdout
        assert md5_file(out + ".bin") == m["bin_md5"] and md5_file(out + ".txt") == m["txt_md5"]


def test_prepare_peek_flush_and_merge(qm2, tmp_path):
    """Round-2 entry points: qm2_prepare (no allocation once the stream runs), qm2_stats_peek (a snapshot that
    never blocks and ends at the totals), qm2_flush (launch without waiting), qm2_merge_counters (two replicas
    of one table: the sum of their counts is the count of the whole stream)."""
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    lines = ol.strip_reads(rfq)
    exp_depth, exp_st = ol.count_lines(keys, qm["k"], lines)
    cut = lines.rfind(b"\n", 0, len(lines) // 3) + 1
    for flags in (qm2.FLAG_PARTITION, qm2.FLAG_DIRECT):
        with qm2.Counter(keys, k=qm["k"], flags=flags, staging_bytes=1 << 16, batch_bytes=1 << 18, n_staging=6) as a, \
                qm2.Counter(keys, k=qm["k"], flags=flags, staging_bytes=1 << 16, batch_bytes=1 << 18) as b:
            a.prepare(host=True, raw=True)
            n0 = a.profile()["allocations"]
            assert n0 > 0 and a.stats_peek()["n_kmers"] == 0
            a.submit_zero_copy(lines[:cut])
            a.flush()                                   # launches the pending batch, does not wait
            peek = a.stats_peek()
            assert 0 <= peek["n_kmers"] <= exp_st["n_kmers"]
            _, sa = a.finish(want_depth=False)
            assert a.profile()["allocations"] == n0     # nothing was allocated while the stream ran
            assert a.stats_peek() == sa                 # after a sync the snapshot is the total
            b.submit(lines[cut:])
            _, sb = b.finish(want_depth=False)
            assert {x: sa[x] + sb[x] for x in sa} == exp_st
            a.merge_counters(b)
            depth, _ = a.finish()
            assert np.array_equal(depth, exp_depth)
            d_b, _ = b.finish()                         # the source keeps its own counts
            assert int(d_b.astype(np.uint64).sum()) == sb["n_hits"]
        # the same without a qm2_finish in front of the merge: both contexts still hold part of their counts in the
        # 2-byte interior arrays (partitioned pipeline), and the merge must add those in first, on both sides
        with qm2.Counter(keys, k=qm["k"], flags=flags) as a, qm2.Counter(keys, k=qm["k"], flags=flags) as b:
            a.submit(lines[:cut]); a.sync()
            b.submit(lines[cut:]); b.sync()
            a.merge_counters(b)
            a.submit(lines[:cut]); a.sync()            # ... and keeps counting on top of the merged counts
            depth, st = a.finish()
            e2, _ = ol.count_lines(keys, qm["k"], lines + lines[:cut])
            assert np.array_equal(depth, e2)


def test_keys_changed_under_a_live_context(qm2, torch_cuda, tmp_path):
    """A caller's key table (qm2_opts.ext_keys) is digested once per context — the runs of occupied slots, the table image
    bucket_count streams; a caller that rewrites it says so (qm2_keys_changed, ABI 6) and the next batch counts against the
    NEW keys on both pipelines.  (Without the call the partitioned pipeline would go on matching the old image: ADVICE r04.)"""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    lines = ol.strip_reads(rfq)
    keys1 = qm["keys"]
    keys2 = keys1.copy()
    occ = np.flatnonzero(keys2)
    keys2[occ[::3]] = 0                                            # another table in the same array: a third of the keys gone
    for flags in (qm2.FLAG_PARTITION, qm2.FLAG_DIRECT):
        d_keys = torch.from_numpy(keys1.view(np.int64).copy()).cuda()
        d_cnt = torch.zeros(len(keys1), dtype=torch.int32, device="cuda")
        with qm2.Counter(None, hash_size=len(keys1), k=30, flags=flags, ext_keys=d_keys.data_ptr(), ext_counters=d_cnt.data_ptr()) as c:
            c.submit(lines)
            depth, st = c.finish()
            e1, s1 = ol.count_lines(keys1, 30, lines)
            assert st == s1 and np.array_equal(depth, e1)
            d_keys.copy_(torch.from_numpy(keys2.view(np.int64).copy()))
            torch.cuda.synchronize()
            c.keys_changed()
            c.reset()
            c.submit(lines)
            depth, st = c.finish()
            e2, s2 = ol.count_lines(keys2, 30, lines)
            assert st == s2 and np.array_equal(depth, e2), flags
            assert s2["n_hits"] < s1["n_hits"]
            assert c.profile()["refused_batches"] == 0


def test_bind_near_device_is_only_a_hint(qm2, tmp_path):
    """qm2_bind_near_device: the calling thread onto the CPUs of the GPU's NUMA node (a subset of what it had, never empty);
    counting is what it was.  Run in a child process: the affinity of the test runner stays alone."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import pyqm2
before = os.sched_getaffinity(0)
n = pyqm2.bind_near_device(0)
after = os.sched_getaffinity(0)
assert 1 <= n == len(after) and after <= before, (n, len(after), len(before))
os.environ["QM2_NO_BIND"] = "1"
assert pyqm2.bind_near_device(0) == len(after)
print("BIND_OK", len(before), len(after))
""" % os.path.join(ROOT, "quick-mer2_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BIND_OK" in r.stdout, r.stdout + r.stderr


def test_empty_and_degenerate_inputs(qm2):
    qm = kat_qm()
    with qm2.Counter(qm["keys"], k=30) as c:
        c.submit(b"")
        c.submit(b"\n")
        c.submit(b"\n\n\n")
        c.submit(b"ACGT\n")
        depth, st = c.finish()
        assert st == {"n_bases": 4, "n_kmers": 0, "n_probes": 0, "n_hits": 0}
        assert not depth.any()
        with pytest.raises(qm2.Qm2Error):
            c.submit(b"ACGT")          # not newline-terminated
    with pytest.raises(qm2.Qm2Error):
        qm2.Counter(qm["keys"][:1000], k=30)       # not a power of two
    with pytest.raises(qm2.Qm2Error):
        qm2.Counter(qm["keys"], k=32)


def test_wrap_mod_65536_and_reset(qm2):
    qm = kat_qm()
    one = synth.kat_reference()[1000:1030].tobytes()
    with qm2.Counter(qm["keys"], k=30) as c:
        c.submit((one + b"\n") * 70000)
        depth, st = c.finish()
        assert st["n_hits"] == 70000 and int(depth.max()) == 4464 and int((depth != 0).sum()) == 1
        c.reset()
        depth, st = c.finish()
        assert st["n_kmers"] == 0 and not depth.any()


def test_interior_counters_outgrow_16_bits(qm2, torch_cuda):
    """The private interior of a bucket is counted in 2 bytes per slot in HBM (the u32 LDS tile starts from them and
    is stored back), everything atomic in the caller's u32 array; a tile value above 65535 sends its upper part to the
    wide array.  One dictionary k-mer that sits in a bucket interior, 2000 times per batch — inside its bucket's
    capacity, so it goes through the tile and not through the spill list — over 40 batches: 80 000.  The u32 array
    behind qm2_finish must hold exactly that (a lost upper part would still pass every mod-65536 check), the uint16
    depth 80 000 - 65 536, every other slot the oracle's count."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    H = len(keys)
    inner = [s_ for s_ in np.flatnonzero(keys) if 512 <= (int(s_) % 4096) < 3584]
    slot = int(inner[len(inner) // 2])
    km = int(keys[slot])
    one = "".join("ACTG"[(km >> (2 * (29 - i))) & 3] for i in range(30)).encode()      # code = (c >> 1) & 3 (:411): A 0, C 1, T 2, G 3
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(11)
    normal = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 4000))
    batch = normal + (one + b"\n") * 2000
    d_keys = torch.from_numpy(keys.view(np.int64)).cuda()
    counters = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=30, ext_keys=d_keys.data_ptr(), ext_counters=counters.data_ptr(),
                     flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as c:
        for _ in range(40):
            c.submit(batch)
            c.sync()
        depth, st = c.finish()
        prof = c.profile()
    assert prof["partitioned_batches"] == 40 and prof["fallback_batches"] == 0 and prof["spilled_kmers"] == 0
    exp1, st1 = ol.count_lines(keys, 30, batch)
    wide = counters.cpu().numpy().view(np.uint32)
    assert np.array_equal(wide, 40 * exp1.astype(np.uint32))              # (one batch stays far below 65536 everywhere)
    assert int(wide[slot]) >= 80000 and int(depth[slot]) == int(wide[slot]) - 65536
    assert np.array_equal(depth, (wide & 0xFFFF).astype(np.uint16))
    assert st == {a: 40 * b for a, b in st1.items()}


def test_skewed_input_spills_and_falls_back(qm2):
    """Partitioned pipeline under skew: poly-A floods one bucket.  A mild flood goes through the
    spill list; a total one overflows even that and the batch is recounted by the direct kernel
    (decided on the device).  Either way the counts are exact."""
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(8)
    normal = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 4000))
    mild = normal + (b"A" * 150 + b"\n") * 600 + (b"T" * 150 + b"\n") * 150      # 16 % of the batch in one bucket: spill list
    total = (b"A" * 150 + b"\n") * 12000
    for name, blob in (("mild", mild), ("total", total)):
        exp_depth, exp_st = ol.count_lines(keys, 30, blob)
        with qm2.Counter(keys, k=30, flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as c:
            c.submit(blob)
            depth, st = c.finish()
            assert c.profile()["fallback_batches"] == (1 if name == "total" else 0), name   # spill list vs recount
            c.submit(normal)                       # the context keeps working after a fallback
            depth2, st2 = c.finish()
        assert st == exp_st, name
        assert np.array_equal(depth, exp_depth), name
        e2, s2 = ol.count_lines(keys, 30, blob + normal)
        assert st2 == s2 and np.array_equal(depth2, e2), name
    z = ol.zero_slot(keys)
    assert exp_depth[z] == (12000 * 121) % 65536       # k-mer 0 lands on the empty slot (QuicKmer.c:96-98)


def test_a_wrong_kmers_per_kib_hint_costs_speed_not_counts(qm2, torch_cuda):
    """qm2_opts.kmers_per_kib sizes the partition buffers from what the caller expects of its input (the CLI reads it off
    the head of the file).  It is only a bound on memory: reads that yield more k-mers than announced overflow their
    buckets into the spill list, and past that the batch is recounted by the direct kernel — the counters and statistics
    are the oracle's whatever the hint says, and the profile shows which way it went."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(11)
    blob = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 60000))      # 9 MB, k-mers per byte 0.8
    exp_depth, exp_st = ol.count_lines(keys, 30, blob)
    seen = {}
    for hint in (0, 840, 400, 100, 52):                 # none, right (0.82), half, an eighth, the minimum the library takes
        with qm2.Counter(keys, k=30, flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING, kmers_per_kib=hint) as c:
            c.submit(blob)
            depth, st = c.finish()
            pr = c.profile()
        assert st == exp_st, hint
        assert np.array_equal(depth, exp_depth), hint
        assert pr["partitioned_batches"] == 1
        seen[hint] = (pr["spilled_kmers"], pr["fallback_batches"])
    assert seen[0] == (0, 0) and seen[840] == (0, 0)                       # a right hint: nothing spills
    assert seen[400][0] > 0 or seen[400][1] == 1                           # half the room: the spill list (or the recount) takes the rest
    assert seen[52][1] == 1 or seen[52][0] > 0


def test_long_occupied_runs(qm2):
    """bucket_count stages a bucket's slots plus a 256-slot halo and flushes the interior with plain
    stores; that is only exact while no foreign probe walk reaches it.  A table with a longer run
    of occupied slots (here: the gaps of a 300- and a 9000-slot stretch filled with keys no 30-mer
    can equal, so that walks run on through two whole buckets) must still give the reference's
    counts: the buckets such a run touches are flushed with atomics.  A run too long to follow
    (1 M slots) sends the table to the direct kernel."""
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(81)
    blob = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 6000))
    H = len(qm["keys"])
    for name, start, length in (("lower_half", H // 4, 300), ("upper_half", 3 * H // 4, 300),
                                ("two_buckets_up", H // 8, 9000), ("two_buckets_down", 5 * H // 8, 9000)):
        keys = qm["keys"].copy()
        gap = np.flatnonzero(keys[start:start + length] == 0) + start
        keys[gap] = (np.uint64(1) << np.uint64(62)) + gap.astype(np.uint64)      # > any 60-bit k-mer
        exp_depth, exp_st = ol.count_lines(keys, 30, blob)
        with qm2.Counter(keys, k=30, flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as c:
            c.submit(blob)
            c.submit(blob)
            depth, st = c.finish()
            prof = c.profile()
        assert st == {a: 2 * v for a, v in exp_st.items()}, name
        assert np.array_equal(depth, (2 * exp_depth.astype(np.uint32)).astype(np.uint16)), name
        assert prof["partitioned_batches"] >= 1 and prof["direct_batches"] == 0, (name, prof)
    # hopeless: one run of 2^20 + 10 slots -> direct kernel, same answer
    H2 = 1 << 22
    keys = np.zeros(H2, dtype=np.uint64)
    run = np.arange(1000, 1000 + (1 << 20) + 10, dtype=np.uint64)
    keys[run] = (np.uint64(1) << np.uint64(62)) + run
    few = blob[:151 * 6]
    exp_depth, exp_st = ol.count_lines(keys, 30, few)
    with qm2.Counter(keys, k=30, flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as c:
        c.submit(few)
        depth, st = c.finish()
        prof = c.profile()
    assert st == exp_st and np.array_equal(depth, exp_depth)
    assert prof["partitioned_batches"] == 0 and prof["direct_batches"] >= 1, prof


def test_dictionary_streamed_from_the_file(qm2, torch_cuda, tmp_path):
    """qm2_keys_from_fd / qm2_chain_gather_fd (what the CLI uses: no host copy of the table) against
    the host-pointer calls, on the mini set and on a synthetic 600 MB .qm that takes several 64 MiB
    pieces per reader thread; counting before the keys are there and a file that ends early are
    errors."""
    path = os.path.join(GOLD, "mini", "ref.fa.qm")
    qm = ol.load_qm(path)
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(5)
    blob = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 3000))
    with qm2.Counter(qm["keys"], k=30) as c:
        c.submit(blob)
        depth, st = c.finish()
        exp_bin = c.chain_gather(qm["chain"], qm["first"])
    with open(path, "rb") as f, qm2.Counter(None, hash_size=len(qm["keys"]), k=30, flags=qm2.FLAG_KEYS_LATER) as c:
        with pytest.raises(qm2.Qm2Error, match="not been loaded"):
            c.submit(blob)
        c.keys_from_fd(f.fileno())
        with pytest.raises(qm2.Qm2Error):
            c.keys_from_fd(f.fileno())                     # only once
        c.submit(blob)
        d2, s2 = c.finish()
        assert s2 == st and np.array_equal(d2, depth)
        assert np.array_equal(c.chain_gather_fd(f.fileno(), qm["first"]), exp_bin)
    # truncated files
    cut = tmp_path / "cut.qm"
    data = open(path, "rb").read()
    cut.write_bytes(data[:24 + 8 * len(qm["keys"]) - 100])
    with open(cut, "rb") as f, qm2.Counter(None, hash_size=len(qm["keys"]), k=30, flags=qm2.FLAG_KEYS_LATER) as c:
        with pytest.raises(qm2.Qm2Error, match="truncated"):
            c.keys_from_fd(f.fileno())
    cut.write_bytes(data[:-100])
    with open(cut, "rb") as f, qm2.Counter(None, hash_size=len(qm["keys"]), k=30, flags=qm2.FLAG_KEYS_LATER) as c:
        c.keys_from_fd(f.fileno())
        with pytest.raises(qm2.Qm2Error, match="truncated"):
            c.chain_gather_fd(f.fileno(), qm["first"])
    # many pieces: a 12 Mb synthetic reference at H = 2^25 (256 MiB of keys + 128 MiB of chain = 6 pieces)
    torch = torch_cuda
    n, H = 12_000_000, 1 << 25
    genome = torch.empty(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=9)
    dkeys = torch.empty(H, dtype=torch.int64, device="cuda")
    scratch = torch.empty(H, dtype=torch.int64, device="cuda")
    qm2.tools.dict_build(genome.data_ptr(), n, 30, dkeys.data_ptr(), H, scratch.data_ptr())
    n_reads = 200_000
    reads = torch.empty(n_reads * 151, dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, 150, 150, 9)
    torch.cuda.synchronize()
    keys = dkeys.cpu().numpy().view(np.uint64)
    lines = reads.cpu().numpy().tobytes()
    del genome, dkeys, scratch, reads
    occ = np.flatnonzero(keys).astype(np.uint32)
    perm = rng.permutation(len(occ))
    chain = np.zeros(H, dtype=np.uint32)
    chain[occ[perm]] = occ[np.roll(perm, -1)]               # one cycle through the occupied slots, random order
    first = int(occ[perm[0]])
    big = tmp_path / "big.qm"
    with open(big, "wb") as f:
        f.write(b"QM11" + bytes([30, 0, 100, 100]) + int(H).to_bytes(8, "little") + first.to_bytes(8, "little"))
        keys.tofile(f)
        chain.tofile(f)
    with qm2.Counter(keys, k=30) as c:
        c.submit(lines)
        depth, st = c.finish()
        exp_bin = c.chain_gather(chain, first)
    assert st["n_hits"] > 0.5 * st["n_kmers"]
    with open(big, "rb") as f, qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER) as c:
        c.keys_from_fd(f.fileno())
        c.submit(lines)
        d2, s2 = c.finish()
        assert s2 == st and np.array_equal(d2, depth)
        assert np.array_equal(c.chain_gather_fd(f.fileno(), first), exp_bin)


def test_line_discipline_on_the_device(qm2, torch_cuda, tmp_path):
    """qm2_submit_raw_device: raw FASTA / FASTQ bytes, headers / '+' / qualities
    blanked by the device — against the oracle's line discipline + count on every reference edge
    case, on inputs that span many 4 KiB tiles (headers longer than a tile, lines cut at every
    tile phase, lines that wrap the uint16 run counter, '>' lines at FASTQ sequence positions), and the
    amounts of the last two it refuses rather than count wrongly."""
    torch = torch_cuda
    qm = kat_qm()
    keys = qm["keys"]
    cases = dict(synth.kat_cases(synth.kat_reference()))
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(77)

    def reads(n, lo, hi):
        return [ref[int(p):int(p) + int(l)] for p, l in zip(rng.integers(0, 4000, n), rng.integers(lo, hi, n))]

    cases.update({
        "fq_many_tiles": ("fq", b"".join(b"@r%d extra words\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n"
                                         for i, r in enumerate(reads(60000, 30, 300)))),
        "fa_many_tiles": ("fa", b"".join(b">" + b"h" * int(h) + b"\n" + r + b"\n"
                                         for h, r in zip(rng.integers(0, 9000, 3000), reads(3000, 1, 900)))),
        "fa_multiline_records": ("fa", b"".join(b">c%d\n" % i + b"".join(ref[j:j + 60] + b"\n" for j in range(0, 4800, 60))
                                                for i in range(40))),
        # a header and a sequence line longer than a GROUP of tiles (1 MiB): the start of the line in progress
        # comes from the scan over the groups; the 3 MB line wraps the uint16 run counter 45 times, with Ns
        "fa_huge_header": ("fa", b">" + b"x" * (3 << 20) + b"\n" + ref[:3000] + b"\n>y\n" + ref[1000:4000] + b"\n"),
        "fa_huge_line": ("fa", b">c\n" + ref[:4800] * 300 + b"N" + ref[:4800] * 250 + b"NN" + ref[:4800] * 150 + b"\n" + ref[:500] + b"\n"),
        "fq_no_final_newline": ("fq", b"@a\n" + ref[:150] + b"\n+\n" + b"I" * 150 + b"\n@b\n" + ref[200:290]),
        "fq_crlf": ("fq", b"".join(b"@r\r\n" + r + b"\r\n+\r\n" + b"I" * len(r) + b"\r\n" for r in reads(500, 40, 200))),
        "fq_empty_lines": ("fq", b"@a\n\n+\n\n@b\n" + ref[:100] + b"\n+\n" + b"I" * 100 + b"\n"),
    })
    refused = {}
    with qm2.Counter(keys, k=30) as c:
        for name, (kind, data) in cases.items():
            if not data:
                continue
            p = tmp_path / ("%s.%s" % (name, kind))
            p.write_bytes(data)
            exp_depth, exp_st = ol.count_lines(keys, 30, ol.strip_reads(str(p)))
            fastq = data[:1] == b"@"                                   # QuicKmer.c:395
            d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            c.reset()
            verdict = c.submit_raw_device(d.data_ptr(), d.numel(), fastq, skip=1 if fastq else 0)
            depth, st = c.finish()
            if verdict:
                refused[name] = verdict
                assert st["n_bases"] == 0 and not depth.any(), name     # nothing of a refused block is counted
                continue
            assert st == exp_st, name
            assert np.array_equal(depth, exp_depth), name
        # a FASTQ block that starts in the middle of a record: skip = lines left to drop
        body = cases["fq_many_tiles"][1]
        lines = body.split(b"\n")
        for drop, start in ((0, 1), (1, 4), (2, 3), (3, 2)):           # block starts at a sequence / header / quality / '+' line
            blob = b"\n".join(lines[start:])
            want = b"".join(l + b"\n" for l in blob.split(b"\n")[drop::4])
            exp_depth, exp_st = ol.count_lines(keys, 30, want)
            d = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            c.reset()
            assert c.submit_raw_device(d.data_ptr(), d.numel(), True, skip=drop) == 0
            depth, st = c.finish()
            assert st["n_kmers"] == exp_st["n_kmers"] and np.array_equal(depth, exp_depth), drop
    # lines beyond the uint16 run counter are counted on the device too (wrap list from the reset scan) ...
    assert not refused, refused
    # ... 8200 of them in one batch as well (the list is sorted on the host): all-A lines only touch k-mer 0's slot
    with qm2.Counter(keys, k=30) as c:
        big = torch.full((8200, 65537), ord("A"), dtype=torch.uint8, device="cuda")
        big[:, 65536] = 10
        torch.cuda.synchronize()
        assert c.submit_raw_device(big.data_ptr(), big.numel(), False) == 0
        depth, st = c.finish()
        _, one = ol.count_lines(keys, 30, b"A" * 65536 + b"\n")          # the wrap costs a line its last window (:402)
        assert one["n_kmers"] == 65536 - 30
        assert st == {a: 8200 * v for a, v in one.items()}
        assert int(depth.astype(np.uint64).sum()) == (8200 * one["n_hits"]) % 65536
        del big
    # '>' at FASTQ sequence positions (QuicKmer.c:398 skips such a line without the three discards): found one
    # at a time on the device, each shifting every later record by a line; more than 64 in a block are refused
    def fq_with_gt(n_rec, bad):
        # bad: record -> how many '>' lines stand in front of its sequence (each at a sequence position); a '>' INSTEAD
        # of the sequence (count -1) shifts the rest of the file against the record structure, as in the reference
        out = []
        for i in range(n_rec):
            g = bad.get(i, 0)
            seq = b">oops%d" % i if g < 0 else ref[i % 3000:i % 3000 + 100]
            out.append(b"@r%d\n" % i + b">x\n" * max(g, 0) + seq + b"\n+\n" + b"I" * 100 + b"\n")
        return b"".join(out)
    with qm2.Counter(keys, k=30) as c:
        for bad in ({7: 1}, {0: 1}, {3: 2, 4: 1, 2000: 5}, {10: -1}, {10: -1, 11: -1, 400: 3}, {100: 64}, {100: 40, 2500: 24}):
            data = fq_with_gt(3000, bad)
            p = tmp_path / "gt.fq"
            p.write_bytes(data)
            exp_depth, exp_st = ol.count_lines(keys, 30, ol.strip_reads(str(p)))
            d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            c.reset()
            assert c.submit_raw_device(d.data_ptr(), d.numel(), True, skip=1) == 0, sorted(bad)[:4]
            depth, st = c.finish()
            assert st == exp_st and np.array_equal(depth, exp_depth), sorted(bad)[:4]
        data = fq_with_gt(3000, {100: 65})                               # 65 of them
        d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        c.reset()
        assert c.submit_raw_device(d.data_ptr(), d.numel(), True, skip=1) == 1
        depth, st = c.finish()
        assert st["n_bases"] == 0 and not depth.any()
        with pytest.raises(qm2.Qm2Error, match="aligned"):
            c.submit_raw_device(d.data_ptr() + 1, d.numel() - 1, True, skip=1)


def test_raw_blocks_through_staging_cross_device_batches(qm2, tmp_path):
    """qm2_staging_commit_raw the way the CLI's reader drives it — line-aligned raw blocks with the
    line state at their first byte — but with device batches of 256 KiB, so that one FASTQ file
    becomes dozens of raw batches whose state comes from whichever block happens to open them."""
    import ctypes as C
    qm = kat_qm()
    keys = qm["keys"]
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(12)
    recs = [ref[int(p):int(p) + int(l)] for p, l in zip(rng.integers(0, 4000, 30000), rng.integers(30, 300, 30000))]
    gt = {5000, 5001, 17000}                 # '>' at sequence positions: the shift they cause is carried from batch to batch
    fq = b"".join(b"@r%d\n" % i + (b">x" if i in gt else r) + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(recs))
    fa = b"".join(b">r%d\n" % i + r + b"\n" for i, r in enumerate(recs))
    want = {True: None, False: b"".join(r + b"\n" for r in recs)}
    (tmp_path / "gt.fq").write_bytes(fq)
    want[True] = ol.strip_reads(str(tmp_path / "gt.fq"))
    exp = {f: ol.count_lines(keys, 30, w) for f, w in want.items()}
    L = qm2.lib()
    for data, fastq in ((fq, True), (fa, False)):
        with qm2.Counter(keys, k=30, staging_bytes=48 << 10, batch_bytes=256 << 10, n_staging=4) as c:
            at, lines_before = 0, 0
            while at < len(data):
                end = min(len(data), at + int(rng.integers(1, 48 << 10)))
                end = data.rfind(b"\n", at, end) + 1
                if end <= at:
                    end = data.find(b"\n", at) + 1
                skip = (1 - lines_before) % 4 if fastq else 0          # file starts at an '@' header: sequence lines are 1, 5, 9, ...
                buf, cap = C.c_void_p(), C.c_size_t()
                assert L.qm2_staging_acquire(c._h, C.byref(buf), C.byref(cap)) == 0
                assert end - at <= cap.value
                off = int(rng.integers(0, 64))                          # the reader's blocks do not start at the buffer's first byte
                off = min(off, cap.value - (end - at))
                C.memmove(buf.value + off, data[at:end], end - at)
                assert L.qm2_staging_commit_raw(c._h, buf, off, end - at, 1 if fastq else 0, skip) == 0
                lines_before += data.count(b"\n", at, end)
                at = end
            depth, st = c.finish()
            assert L.qm2_raw_verdict(c._h) == 0
            prof = c.profile()
            assert prof["direct_batches"] + prof["partitioned_batches"] >= 8      # really many batches
        exp_depth, exp_st = exp[fastq]
        assert st == exp_st, fastq
        assert np.array_equal(depth, exp_depth), fastq


# ---- 3. device-resident input, any alignment -----------------------------------

def test_submit_device_unaligned(qm2, torch_cuda, tmp_path):
    torch = torch_cuda
    qm = kat_qm()
    cases = synth.kat_cases(synth.kat_reference())
    p = tmp_path / "r.fa"
    p.write_bytes(cases["reads_fa"][1])
    blob = ol.strip_reads(str(p))[:200000]
    blob = blob[:blob.rfind(b"\n") + 1]
    exp_depth, exp_st = ol.count_lines(qm["keys"], 30, blob)
    for mis in (0, 1, 7, 15, 16, 33):
        buf = torch.zeros(len(blob) + 64, dtype=torch.uint8, device="cuda")
        buf[mis:mis + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION):
            with qm2.Counter(qm["keys"], k=30, flags=flags) as c:
                c.submit_device(buf.data_ptr() + mis, len(blob))
                depth, st = c.finish()
            assert st == exp_st and np.array_equal(depth, exp_depth), (mis, flags)


# ---- 4. split path: encode + probe, sharded spans -------------------------------

def test_encode_then_probe_equals_fused(qm2, torch_cuda, tmp_path):
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys = qm["keys"]
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    blob = ol.strip_reads(rfa)
    exp_kmers, nb = ol.extract(blob, 30)
    exp_depth, exp_st = ol.count_lines(keys, 30, blob)
    dbuf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    for n_shards in (1, 2, 4):
        cap = len(exp_kmers) + 16
        out = torch.zeros(n_shards * cap, dtype=torch.int64, device="cuda")
        counts = torch.zeros(n_shards, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()          # the library runs on its own stream
        with qm2.Counter(keys, k=30) as enc:
            enc.encode_device(dbuf.data_ptr(), len(blob), out.data_ptr(), cap, n_shards, counts.data_ptr())
            enc.sync()
            _, st = enc.finish(want_depth=False)
        assert st["n_kmers"] == len(exp_kmers) and st["n_bases"] == nb
        cnt = counts.cpu().numpy()
        assert int(cnt.sum()) == len(exp_kmers)
        got = np.concatenate([out[g * cap:g * cap + int(cnt[g])].cpu().numpy().view(np.uint64) for g in range(n_shards)])
        assert np.array_equal(np.sort(got), np.sort(exp_kmers))
        # every k-mer sits in the bucket of its home slot's owner
        lgH = int(np.log2(len(keys)))
        for g in range(n_shards):
            km = out[g * cap:g * cap + int(cnt[g])].cpu().numpy().view(np.uint64)
            L = ol.lib()
            homes = np.array([L.qm2o_djb(int(x)) & (len(keys) - 1) for x in km[:2000]], dtype=np.uint64)
            assert np.all((homes >> (lgH - int(np.log2(n_shards)))) == g) if n_shards > 1 else True
        # probe each bucket on a context that holds only that shard's span
        total = np.zeros(len(keys), dtype=np.uint64)
        hits = probes = 0
        for g in range(n_shards):
            own_lo, own_hi, lo, hi = qm2.shard_span(keys, n_shards, g)
            assert lo <= own_lo and hi >= own_hi
            with qm2.Counter(keys[lo:hi], hash_size=len(keys), k=30, span=(lo, hi)) as pc:
                pc.probe_device(out.data_ptr() + 8 * g * cap, int(cnt[g]))
                d, s = pc.finish()
            total[lo:hi] += d
            hits += s["n_hits"]
            probes += s["n_probes"]
        assert hits == exp_st["n_hits"] and probes == exp_st["n_probes"]
        assert np.array_equal(total.astype(np.uint16), exp_depth)


def _count_on_logical_shards(qm2, torch, keys, k, dparts, G, cap, flags):
    """qm2_route_device for every "sender" in dparts, then every shard's span probed with what was routed to
    it (qm2_probe_slabs_device); -> (uint64 depth[H] with the overhangs folded in, hits, probes, profiles)."""
    H = len(keys)
    spans = [qm2.shard_span(keys, G, g) for g in range(G)]
    routed = []
    with qm2.Counter(None, hash_size=H, k=k, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        for d in dparts:
            slabs = torch.zeros(G * cap, dtype=torch.int64, device="cuda")
            fill = torch.zeros(G, dtype=torch.int32, device="cuda")
            call = torch.zeros(3, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            router.route_device(d.data_ptr(), d.numel(), slabs.data_ptr(), cap, G, fill.data_ptr(), call.data_ptr())
            router.sync()
            assert int(call[2]) == 0
            routed.append((slabs, fill, call))
    total = np.zeros(H, dtype=np.uint64)
    hits = probes = 0
    profs = []
    for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
        recv = torch.cat([r[0][g * cap:(g + 1) * cap] for r in routed])
        rfill = torch.stack([r[1][g] for r in routed]).contiguous()
        torch.cuda.synchronize()
        with qm2.Counter(keys[lo:hi], hash_size=H, k=k, span=(lo, hi), own=(own_lo, own_hi) if G > 1 else None,
                         flags=flags | qm2.FLAG_TIMING) as pc:
            pc.probe_slabs_device(recv.data_ptr(), cap, len(routed), rfill.data_ptr())
            d, st = pc.finish()
            profs.append((hi - lo, pc.profile()))
        total[lo:hi] += d
        hits += st["n_hits"]
        probes += st["n_probes"]
    return total, hits, probes, profs


def test_shards_with_long_runs_across_their_edges(qm2, torch_cuda):
    """Hash-range shards whose walks leave the owned range by thousands of slots: a 9000-slot run of
    occupied slots across the edge between two lower-half shards (walks go up, into the neighbour's range)
    and one across an edge in the upper half (walks go down).  The span (qm2_shard_span) then carries a
    long overhang, the last buckets' halos lie outside the owned range, walks finish in global memory
    inside the span, and the long-run marking is done per span: counts, hits and probes must still be
    the oracle's, with the partitioned pipeline on every shard."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    ref = synth.kat_reference().tobytes()
    rng = np.random.default_rng(82)
    blob = b"".join(ref[int(p):int(p) + 150] + b"\n" for p in rng.integers(0, 4800, 8000))
    half = blob.rfind(b"\n", 0, len(blob) // 2) + 1
    dparts = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in (blob[:half], blob[half:])]
    H = len(qm["keys"])
    n_km = len(ol.extract(blob, 30)[0])
    for name, start, length in (("up_across_an_edge", H // 4 - 3000, 9000), ("down_across_an_edge", 3 * H // 4 - 6000, 9000),
                                ("short_of_the_halo", H // 4 - 100, 230)):
        keys = qm["keys"].copy()
        gap = np.flatnonzero(keys[start:start + length] == 0) + start
        keys[gap] = (np.uint64(1) << np.uint64(62)) + gap.astype(np.uint64)      # > any 60-bit k-mer
        # make sure reads hit slots INSIDE the run: plant the homes of real read k-mers there?  the mini table's own
        # keys inside the stretch stay where they are, so walks that start in it run on to its end
        exp_depth, exp_st = ol.count_lines(keys, 30, blob)
        for G in (2, 4):
            total, hits, probes, profs = _count_on_logical_shards(qm2, torch, keys, 30, dparts, G, (n_km + 64) & ~1, qm2.FLAG_PARTITION)
            assert hits == exp_st["n_hits"] and probes == exp_st["n_probes"], (name, G)
            assert np.array_equal(total.astype(np.uint16), exp_depth), (name, G)
            assert all(p["partitioned_batches"] == 1 for n_slots, p in profs if n_slots >= 2 * 4096), (name, G)


def test_route_then_probe_slabs_equals_fused(qm2, torch_cuda, tmp_path):
    """The multi-GPU driver's two legs (qm2_route_device: encode + bucket by owner into fixed slabs;
    qm2_probe_slabs_device: a shard's span probed with the partitioned pipeline) on 1 / 2 / 4 / 8 logical
    shards of the mini table, slabs from two "senders": counters, hits and probes equal the oracle's;
    a slab that is too small reports what did not fit instead of dropping it silently."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys, H = qm["keys"], len(qm["keys"])
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    blob = ol.strip_reads(rfa)
    half = blob.rfind(b"\n", 0, len(blob) // 2) + 1
    parts = [blob[:half], blob[half:]]
    exp_kmers, nb = ol.extract(blob, 30)
    exp_depth, exp_st = ol.count_lines(keys, 30, blob)
    dparts = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in parts]
    for G in (1, 2, 4, 8):
        cap = (len(exp_kmers) * 2 // 3 + 64) & ~1
        spans = [qm2.shard_span(keys, G, g) for g in range(G)]
        routed = []
        with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
            for d in dparts:
                slabs = torch.zeros(G * cap, dtype=torch.int64, device="cuda")
                fill = torch.zeros(G, dtype=torch.int32, device="cuda")
                call = torch.zeros(3, dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                router.route_device(d.data_ptr(), d.numel(), slabs.data_ptr(), cap, G, fill.data_ptr(), call.data_ptr())
                router.sync()
                routed.append((slabs, fill, call))
            assert router.finish(want_depth=False)[1]["n_kmers"] == 0        # the context's own statistics stay out of it
        assert sum(int(r[2][0]) for r in routed) == nb and sum(int(r[2][1]) for r in routed) == len(exp_kmers)
        assert all(int(r[2][2]) == 0 for r in routed)
        got = np.concatenate([r[0][g * cap:g * cap + int(r[1][g])].cpu().numpy().view(np.uint64) for r in routed for g in range(G)])
        assert np.array_equal(np.sort(got), np.sort(exp_kmers))
        for flags in (qm2.FLAG_PARTITION, qm2.FLAG_DIRECT):
            total = np.zeros(H, dtype=np.uint64)
            hits = probes = 0
            for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
                # what shard g receives: slab g of every sender, side by side
                recv = torch.cat([r[0][g * cap:(g + 1) * cap] for r in routed])
                rfill = torch.stack([r[1][g] for r in routed]).contiguous()
                torch.cuda.synchronize()
                with qm2.Counter(keys[lo:hi], hash_size=H, k=30, span=(lo, hi), own=(own_lo, own_hi) if G > 1 else None,
                                 flags=flags | qm2.FLAG_TIMING) as pc:
                    pc.probe_slabs_device(recv.data_ptr(), cap, len(routed), rfill.data_ptr())
                    d, st = pc.finish()
                    prof = pc.profile()
                if flags == qm2.FLAG_PARTITION and hi - lo >= 2 * 4096:
                    assert prof["partitioned_batches"] == 1 and prof["class_launches"]["bucket_count"] == 1, (G, g)
                total[lo:hi] += d
                hits += st["n_hits"]
                probes += st["n_probes"]
            assert hits == exp_st["n_hits"] and probes == exp_st["n_probes"], (G, flags)
            assert np.array_equal(total.astype(np.uint16), exp_depth), (G, flags)
    # too small a slab: the fill says what was routed, d_call[2] what did not fit
    small = 1000
    slabs = torch.zeros(2 * small, dtype=torch.int64, device="cuda")
    fill = torch.zeros(2, dtype=torch.int32, device="cuda")
    call = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        router.route_device(dparts[0].data_ptr(), dparts[0].numel(), slabs.data_ptr(), small, 2, fill.data_ptr(), call.data_ptr())
        router.sync()
    n0 = len(ol.extract(parts[0], 30)[0])
    assert int(fill.sum()) == n0 and int(call[2]) == n0 - 2 * small and int(call[1]) == n0


def test_route_by_sub_slabs_is_the_receivers_first_level(qm2, torch_cuda, tmp_path):
    """qm2_route_device with n_owners x n_subs "owners" + qm2_probe_subslabs_device: the senders cut every owner's slab
    by the next bits of the home slot, and what a shard receives is the first level of its own partition (no slab split
    on the receiving side).  1 / 2 / 4 / 8 logical shards of the mini table, two senders, every feasible width of the
    sub-slab level: counters, hits and probes equal the oracle's, and the partitioned pipeline really ran."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys, H = qm["keys"], len(qm["keys"])
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    blob = ol.strip_reads(rfa)
    half = blob.rfind(b"\n", 0, len(blob) // 2) + 1
    dparts = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in (blob[:half], blob[half:])]
    exp_kmers, nb = ol.extract(blob, 30)
    exp_depth, exp_st = ol.count_lines(keys, 30, blob)
    lgH = H.bit_length() - 1
    for G in (1, 2, 4, 8):
        lg_range = lgH - (G.bit_length() - 1)
        widest = lg_range - 12 - 1                        # buckets are 2^12 slots; one bit at least must be left to split
        spans = [qm2.shard_span(keys, G, g) for g in range(G)]
        for lg_subs in sorted({1, widest}):
            n_subs = 1 << lg_subs
            cap = (len(exp_kmers) // (G * n_subs) * 2 + 4096) & ~1
            routed = []
            with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
                for d in dparts:
                    slabs = torch.zeros(G * n_subs * cap, dtype=torch.int64, device="cuda")
                    fill = torch.zeros(G * n_subs, dtype=torch.int32, device="cuda")
                    call = torch.zeros(3, dtype=torch.int64, device="cuda")
                    torch.cuda.synchronize()
                    router.route_device(d.data_ptr(), d.numel(), slabs.data_ptr(), cap, G * n_subs, fill.data_ptr(), call.data_ptr())
                    router.sync()
                    assert int(call[2]) == 0
                    routed.append((slabs, fill))
            total = np.zeros(H, dtype=np.uint64)
            hits = probes = 0
            for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
                a, b = g * n_subs, (g + 1) * n_subs
                recv = torch.cat([r[0][a * cap:b * cap] for r in routed])            # [sender][sub][cap]
                rfill = torch.cat([r[1][a:b] for r in routed]).contiguous()
                torch.cuda.synchronize()
                with qm2.Counter(keys[lo:hi], hash_size=H, k=30, span=(lo, hi), own=(own_lo, own_hi), flags=qm2.FLAG_TIMING) as pc:
                    pc.probe_subslabs_device(recv.data_ptr(), cap, len(routed), n_subs, rfill.data_ptr())
                    d, st = pc.finish()
                    prof = pc.profile()
                    assert prof["partitioned_batches"] == 1 and prof["class_launches"]["bucket_count"] == 1 and prof["fallback_batches"] == 0
                    assert prof["class_launches"]["split_kmer"] == 1                 # one split behind the received level, none of the slabs
                    # a second call on the same plan, and the old entry point on the same context (it re-plans)
                    pc.probe_subslabs_device(recv.data_ptr(), cap, len(routed), n_subs, rfill.data_ptr())
                    d2, st2 = pc.finish()
                    assert np.array_equal(d2.astype(np.uint32), 2 * d.astype(np.uint32)) and st2["n_hits"] == 2 * st["n_hits"]
                total[lo:hi] += d
                hits += st["n_hits"]
                probes += st["n_probes"]
            assert hits == exp_st["n_hits"] and probes == exp_st["n_probes"], (G, lg_subs)
            assert np.array_equal(total.astype(np.uint16), exp_depth), (G, lg_subs)
    # the router's output in per-XCD sub-regions (qm2_route_xcd_device, x-major): a sender's 8 sub-regions count as 8 peers,
    # G * n_subs regions apart
    G, n_subs = 4, 4
    spans = [qm2.shard_span(keys, G, g) for g in range(G)]
    cap = (len(exp_kmers) // (G * n_subs * 8) * 3 + 4096) & ~1
    routed = []
    with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        for d in dparts:
            slabs = torch.zeros(8 * G * n_subs * cap, dtype=torch.int64, device="cuda")
            fill = torch.zeros(8 * G * n_subs, dtype=torch.int32, device="cuda")
            call = torch.zeros(3, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            router.route_xcd_device(d.data_ptr(), d.numel(), slabs.data_ptr(), cap, G * n_subs, fill.data_ptr(), call.data_ptr())
            router.sync()
            assert int(call[2]) == 0 and int(fill.sum()) == int(call[1])
            routed.append((slabs, fill))
    total = np.zeros(H, dtype=np.uint64)
    hits = probes = 0
    for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
        for slabs, fill in routed:                       # one call per sender: its 8 sub-regions are the peers
            pass
        with qm2.Counter(keys[lo:hi], hash_size=H, k=30, span=(lo, hi), own=(own_lo, own_hi), flags=qm2.FLAG_TIMING) as pc:
            for slabs, fill in routed:
                pc.probe_subslabs_device(slabs.data_ptr() + 8 * g * n_subs * cap, cap, 8, n_subs, fill.data_ptr() + 4 * g * n_subs,
                                         peer_stride=G * n_subs)
            d, st = pc.finish()
            assert pc.profile()["partitioned_batches"] == len(routed)
        total[lo:hi] += d
        hits += st["n_hits"]
        probes += st["n_probes"]
    assert hits == exp_st["n_hits"] and probes == exp_st["n_probes"]
    assert np.array_equal(total.astype(np.uint16), exp_depth)
    # skew through sub-slabs: 120 k poly-A reads put ~14 M copies of k-mer 0 into ONE sub-slab; the receiving shard's bucket and
    # its spill list overflow, and the whole call is recounted region by region (probe_regions_kernel), decided on the device
    flood = (b"A" * 150 + b"\n") * 120_000 + blob[:blob.rfind(b"\n", 0, 200_000) + 1]
    fexp_depth, fexp_st = ol.count_lines(keys, 30, flood)
    d_flood = torch.frombuffer(bytearray(flood), dtype=torch.uint8).cuda()
    n_subs = 2
    cap = (fexp_st["n_kmers"] + 4096) & ~1
    slabs = torch.zeros(n_subs * cap, dtype=torch.int64, device="cuda")
    fill = torch.zeros(n_subs, dtype=torch.int32, device="cuda")
    call = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        router.route_device(d_flood.data_ptr(), d_flood.numel(), slabs.data_ptr(), cap, n_subs, fill.data_ptr(), call.data_ptr())
        router.sync()
    assert int(call[2]) == 0 and int(call[1]) == fexp_st["n_kmers"]
    with qm2.Counter(keys, hash_size=H, k=30, own=(0, H), span=(0, H), flags=qm2.FLAG_TIMING) as pc:
        pc.probe_subslabs_device(slabs.data_ptr(), cap, 1, n_subs, fill.data_ptr())
        d, st = pc.finish()
        prof = pc.profile()
    assert prof["fallback_batches"] == 1, prof
    assert st["n_hits"] == fexp_st["n_hits"] and st["n_probes"] == fexp_st["n_probes"]
    assert np.array_equal(d, fexp_depth)
    # a sub-slab level that leaves nothing to split is refused with a message, not miscounted
    own_lo, own_hi, lo, hi = qm2.shard_span(keys, 8, 0)
    with qm2.Counter(keys[lo:hi], hash_size=H, k=30, span=(lo, hi), own=(own_lo, own_hi)) as pc:
        z = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
        f = torch.zeros(8, dtype=torch.int32, device="cuda")
        with pytest.raises(qm2.Qm2Error):
            pc.probe_subslabs_device(z.data_ptr(), 64, 1, 8, f.data_ptr())


def test_entry_points_mixed_on_one_context_replan(qm2, torch_cuda, tmp_path):
    """qm2_probe_subslabs_device plans the context with level 0 given from outside (no buffer, no cursors of its own
    there); sequence text or one-slab-per-owner input afterwards needs the context's own level 0 again.  The calls may be
    mixed in any order on one context: every one re-plans when the kind changes (a launch on the other kind's plan would
    reserve in the wrong cursors and count stale buffer contents), and the counts add up exactly."""
    torch = torch_cuda
    qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
    keys, H = qm["keys"], len(qm["keys"])
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    blob = ol.strip_reads(rfa)
    d_text = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    exp_kmers, nb = ol.extract(blob, 30)
    exp_depth, exp_st = ol.count_lines(keys, 30, blob)
    n_subs = 4
    cap = (len(exp_kmers) // n_subs * 2 + 4096) & ~1
    sub = torch.zeros(n_subs * cap, dtype=torch.int64, device="cuda")
    sub_fill = torch.zeros(n_subs, dtype=torch.int32, device="cuda")
    one = torch.zeros(n_subs * cap, dtype=torch.int64, device="cuda")
    one_fill = torch.zeros(1, dtype=torch.int32, device="cuda")
    call = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=30, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        router.route_device(d_text.data_ptr(), d_text.numel(), sub.data_ptr(), cap, n_subs, sub_fill.data_ptr(), call.data_ptr())
        router.route_device(d_text.data_ptr(), d_text.numel(), one.data_ptr(), n_subs * cap, 1, one_fill.data_ptr(), call.data_ptr())
        router.sync()
    assert int(call[2]) == 0 and int(one_fill[0]) == len(exp_kmers)
    orders = [("sub", "text", "sub", "slab", "text", "sub"), ("text", "sub", "slab", "sub"), ("slab", "sub", "text")]
    for order in orders:
        with qm2.Counter(keys, hash_size=H, k=30, own=(0, H), span=(0, H), flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as pc:
            for what in order:
                if what == "sub":
                    pc.probe_subslabs_device(sub.data_ptr(), cap, 1, n_subs, sub_fill.data_ptr())
                elif what == "slab":
                    pc.probe_slabs_device(one.data_ptr(), n_subs * cap, 1, one_fill.data_ptr())
                else:
                    pc.submit_device(d_text.data_ptr(), d_text.numel())
            d, st = pc.finish()
            prof = pc.profile()
        n = len(order)
        assert prof["partitioned_batches"] == n and prof["fallback_batches"] == 0 and prof["spilled_kmers"] == 0, (order, prof)
        assert st["n_hits"] == n * exp_st["n_hits"] and st["n_probes"] == n * exp_st["n_probes"], order
        assert np.array_equal(d.astype(np.uint32), (n * exp_depth.astype(np.uint32)) & 0xFFFF), order


def test_loader_waves_out_of_step(torch_cuda, tmp_path):
    """bucket_count's loader waves flush bucket k - 1 and stage bucket k + 1 into the same LDS buffer with no barrier in
    between; that is only right while a thread stages over nothing but what it has flushed itself, WHICHEVER way the two
    buckets walk (the slice of a bucket that walks down is staged mirrored: qm2_partition.hip.h, slice_quad).  A build in
    which the last loader wave runs late (-DQM2_TEST_LOADER_SKEW) makes the other waves stage before it flushes — at the
    table's middle the assignment by slice position that the kernel had until round 4 then corrupted the counts of the
    buckets in front of it (448 032 slots of this very workload with round 3's kernel built the same way,
    git show 702b9be:tools/history/r4/skew_old_kernel.py; by chance once before, with two processes on the GPU).  The skewed build must equal the direct kernel slot for slot on a table whose
    workgroups each cross the middle."""
    torch = torch_cuda
    pkg = os.path.join(ROOT, "quick-mer2_amd")
    lib = str(tmp_path / "libqm2_skew.so")
    subprocess.run(["make", "-C", pkg, "variant", "TUNE=-DQM2_TEST_LOADER_SKEW=8", "OUT=" + lib], check=True, stdout=subprocess.DEVNULL)   # (the Makefile's compiler, arch and flags)
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import pyqm2 as qm2
n, H, k = 6_000_000, 1 << 23, 30                  # 2048 finest buckets: every one of 256 workgroups takes 8, 4 either side of the middle
genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
qm2.tools.genome_generate(genome.data_ptr(), n, seed=5, n_blocks=[], dups=[])
keys = torch.zeros(H, dtype=torch.int64, device="cuda")
scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
nd, nu = qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
n_reads, rl = 400_000, 150
reads = torch.zeros(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 5)
torch.cuda.synchronize()
out = {}
for name, flags in (("direct", qm2.FLAG_DIRECT), ("partition", qm2.FLAG_PARTITION)):
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), flags=flags) as c:
        for _ in range(3):                        # (three batches: the counter tile starts from earlier counts)
            c.submit_device(reads.data_ptr(), reads.numel())
        d, st = c.finish()
        prof = c.profile()
    out[name] = (d, st, prof)
assert out["partition"][2]["partitioned_batches"] == 3 and out["direct"][2]["partitioned_batches"] == 0
assert out["partition"][1] == out["direct"][1], (out["partition"][1], out["direct"][1])
bad = np.flatnonzero(out["partition"][0] != out["direct"][0])
assert len(bad) == 0, (len(bad), bad[:8], bad[:8] >> 12)
assert int(out["direct"][0].astype(np.int64).sum()) %% 65536 >= 0 and nu > 1_000_000
print("SKEW_OK", len(out["direct"][0]), int(out["direct"][1]["n_hits"]))
""" % pkg
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QM2_LIB=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SKEW_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_waves_far_out_of_step(torch_cuda, tmp_path):
    """The partitioned pipeline with every wave idling a pseudo-random time at every workgroup barrier and between the
    phases in between (-DQM2_TEST_CHAOS): the kernels may rely on their barriers, never on waves running in step.  Both
    table sizes of the record formats (2- and 4-byte second words need 2^30 and fewer slots), a table whose workgroups
    cross the middle, a skewed stream that goes through the spill list: all equal to the direct kernel slot for slot."""
    pkg = os.path.join(ROOT, "quick-mer2_amd")
    lib = str(tmp_path / "libqm2_chaos.so")
    subprocess.run(["make", "-C", pkg, "variant", "TUNE=-DQM2_TEST_CHAOS=6", "OUT=" + lib], check=True, stdout=subprocess.DEVNULL)
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import pyqm2 as qm2
def case(n, lgH, n_reads, seed, poly=0):
    H, k, rl = 1 << lgH, 30, 150
    genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=seed, n_blocks=[], dups=[])
    keys = torch.zeros(H, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
    qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    reads = torch.zeros(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, seed)
    if poly:                                  # a flood of one k-mer: the first `poly` reads become poly-A lines
        v = reads.view(n_reads, rl + 1)
        v[:poly, :rl] = ord("A")
    torch.cuda.synchronize()
    res = {}
    for name, flags in (("direct", qm2.FLAG_DIRECT), ("partition", qm2.FLAG_PARTITION)):
        with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), flags=flags) as c:
            for _ in range(2):
                c.submit_device(reads.data_ptr(), reads.numel())
            d, st = c.finish()
            prof = c.profile()
        res[name] = (d, st, prof)
    assert res["partition"][2]["partitioned_batches"] == 2 and res["partition"][2]["fallback_batches"] == 0
    assert res["partition"][1] == res["direct"][1], (lgH, res["partition"][1], res["direct"][1])
    bad = np.flatnonzero(res["partition"][0] != res["direct"][0])
    assert len(bad) == 0, (lgH, len(bad), bad[:8])
    return res["partition"][2]["spilled_kmers"]
assert case(6_000_000, 23, 300_000, 5) == 0
assert case(6_000_000, 23, 300_000, 6, poly=20_000) > 0
assert case(1_500_000, 21, 100_000, 7) == 0
assert case(6_000_000, 30, 300_000, 8) == 0          # (2-byte second words: 2^30 slots)
print("CHAOS_OK")
""" % pkg
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QM2_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CHAOS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_replicas_two_ranks_on_one_gpu(torch_cuda, tmp_path):
    """bench.py's N > 1 replica path with TWO ranks (both on this card, collectives over gloo: QM2_REHEARSE_ON_ONE_GPU):
    every rank must complete its counters (fold the 2-byte bucket interiors: qm2_finish) BEFORE the job's one reduce,
    or the merged table lacks the other ranks' interior counts — `counts_conserved` is the check, on the merged result."""
    env = dict(os.environ, QM2_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # (`bench.py --gpus 2` with no rank environment: it starts its two ranks itself, on a free port)
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "c2", "--reads-per-step", "4000000", "--mode", "partition", "--no-streamed",
                        "--no-cpu-baseline", "--no-geometry-check"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["counts_conserved"] is True, (line["config"], r.stderr[-3000:])
    assert line["roofline"]["mode"] == "partitioned" and line["config"]["final_reduce_ms"] is not None
    assert line["value"] > 0


def test_multi_drivers_on_the_gpu_single_rank(qm2, torch_cuda, tmp_path):
    """quick-mer2_amd/multi.py with the real HipEngine and the RCCL backend (world size 1 is all a
    one-GPU box allows; ranks > 1 run under gloo in tests/test_multi_gloo.py): init, all_reduce,
    all_to_all_single of ragged k-mer slabs, overhang fold and assembly all execute on the device."""
    torch = torch_cuda
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    from quick_mer2_amd import multi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29577"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        qm = ol.load_qm(os.path.join(GOLD, "mini", "ref.fa.qm"))
        fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
        lines = ol.strip_reads(rfa)
        exp, _ = ol.count_lines(qm["keys"], qm["k"], lines)
        eng = multi.HipEngine(torch, torch.device("cuda", 0))
        for job in (multi.ReplicaJob(dist, eng, qm["keys"], qm["k"]),
                    multi.ShardedJob(dist, eng, qm["keys"], qm["k"], torch)):
            half = lines.rfind(b"\n", 0, len(lines) // 2) + 1
            job.count(lines[:half])
            job.count(lines[half:])
            got = job.finish()
            job.close()
            assert np.array_equal(got, exp), type(job).__name__
    finally:
        dist.destroy_process_group()


# ---- 4b. epilogue on the device: chain walk as list ranking ------------

def test_chain_gather_matches_reference_bin(qm2, tmp_path):
    for setname, reads_name in (("mini", None), ("kat", "reads_fa"), ("kat", "polyA"), ("kat_k25", "reads_fa")):
        qm = ol.load_qm(os.path.join(GOLD, setname, "ref.fa.qm"))
        if setname == "mini":
            fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
            lines = ol.strip_reads(rfq)
            exp = open(os.path.join(GOLD, "mini", "reads.bin"), "rb").read()
        else:
            p = tmp_path / "r.fa"
            p.write_bytes(synth.kat_cases(synth.kat_reference())[reads_name][1])
            lines = ol.strip_reads(str(p))
            exp = open(os.path.join(GOLD, setname, reads_name + ".bin"), "rb").read()
        for window in (None, 1000):
            with qm2.Counter(qm["keys"], k=qm["k"]) as c:
                c.submit(lines)
                out = c.chain_gather(qm["chain"], qm["first"], window=window)
                with pytest.raises(qm2.Qm2Error):
                    c.submit(lines)             # the key table is gone, as in the reference (:483):
                    c.sync()                    # the staged batch cannot be launched any more
            assert out.tobytes() == exp, (setname, reads_name)


def test_gc_sums_on_the_device(qm2, tmp_path):
    """qm2_chain_gc_sums_fd (QuicKmer.c:504-509 on the device) against the loop written out in numpy: the reference's
    .qgc; a .qgc that marks EVERY k-mer as control over deep counts (50 000^2 does not fit the reference's `int`
    product, :507: wrapped); bins >= 401 ignored; a file cut short, where the entries behind its end take the values the
    reference's 2^20-entry buffer still holds (the stale block; the CLI run against the reference itself pins the rule)."""
    qm = kat_qm()
    cases = synth.kat_cases(synth.kat_reference())
    rng = np.random.default_rng(3)
    for reads_name in ("reads_fa", "deep50000"):
        p = tmp_path / (reads_name + ".fa")
        p.write_bytes(cases[reads_name][1])
        lines = ol.strip_reads(str(p))
        with qm2.Counter(qm["keys"], k=qm["k"]) as c:
            c.submit(lines)
            depth = c.chain_gather(qm["chain"], qm["first"])
            n = len(depth)
            assert depth.tobytes() == open(os.path.join(GOLD, "kat", reads_name + ".bin"), "rb").read()
            ref_qgc = np.fromfile(os.path.join(GOLD, "kat", "ref.fa.qgc"), dtype=np.uint16)
            all_ctrl = (((np.arange(n) * 7) % 512) | 0x8000).astype(np.uint16)      # bins 0..511: 401.. are skipped (the deep k-mer, entry 1000: bin 344)
            stale = rng.integers(0, 1 << 16, 1 << 20).astype(np.uint16)
            for tag, gc, n_file in (("reference", ref_qgc, n), ("all control", all_ctrl, n), ("cut short", all_ctrl, n // 3),
                                    ("empty file", all_ctrl, 0), ("longer file", np.concatenate([ref_qgc, ref_qgc]), 2 * n)):
                f = tmp_path / "x.qgc"
                gc[:n_file].tofile(str(f))
                with open(str(f), "rb") as fh:
                    s_, q_, c_ = c.chain_gc_sums_fd(fh.fileno(), n_file, stale if n_file < n else None)
                eff = np.where(np.arange(n) < n_file, gc[:n] if len(gc) >= n else np.resize(gc, n), stale[np.arange(n) & ((1 << 20) - 1)])
                ctrl = ((eff & 0x8000) != 0) & ((eff & 0x1FF) < 401)
                b = (eff & 0x1FF)[ctrl].astype(np.int64)
                v = depth[ctrl].astype(np.int64)
                sq = ((v * v) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)      # the `int` product, wrapped
                assert (c_ == np.bincount(b, minlength=401).astype(np.uint64)).all(), (reads_name, tag)
                assert (s_ == np.array([v[b == i].sum() for i in range(401)])).all(), (reads_name, tag)
                assert (q_ == np.array([sq[b == i].sum() for i in range(401)])).all(), (reads_name, tag)
                if reads_name == "deep50000" and tag == "all control":
                    assert (sq < 0).any() and (v > 46340).any()                    # the wrap really happens
            with pytest.raises(qm2.Qm2Error):
                with open(str(f), "rb") as fh:
                    c.chain_gc_sums_fd(fh.fileno(), 0, None)                         # short file, no stale block


def test_chain_gather_large_random_cycle(qm2, torch_cuda):
    """2^24-slot table, ~6 M keys chained in a random order; ext_keys (the table survives)."""
    torch = torch_cuda
    n, H, k = 6_000_000, 1 << 24, 30
    genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=41)
    keys = torch.zeros(H, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
    qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    hk = keys.cpu().numpy().view(np.uint64)
    occ = np.flatnonzero(hk).astype(np.uint32)
    rng = np.random.default_rng(1)
    order = occ[rng.permutation(len(occ))]
    chain = np.zeros(H, dtype=np.uint32)
    chain[order] = np.roll(order, -1)
    reads = torch.zeros(300_000 * 151, dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, 300_000, 0, 150, 150, 5)
    torch.cuda.synchronize()
    for first in (int(order[0]), int(order[12345]), int(occ[occ % 64 == 0][3])):
        with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr()) as c:
            c.submit_device(reads.data_ptr(), reads.numel())
            depth, st = c.finish()
            out = c.chain_gather(chain, first)
            c.submit_device(reads.data_ptr(), reads.numel())       # ext_keys: the table is still there
            c.sync()
        pos = int(np.flatnonzero(order == first)[0])
        walk = np.roll(order, -pos)
        assert len(out) == len(occ) and np.array_equal(out, depth[walk])
    # a chain that never returns to the first index is reported, not looped on
    bad = chain.copy()
    bad[order[-1]] = order[5]
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr()) as c:
        with pytest.raises(qm2.Qm2Error):
            c.chain_gather(bad, int(order[0]))


# ---- 5. device tooling ----------------------------------------------------------

def test_tools_dictionary_is_valid_and_unique(qm2, torch_cuda):
    torch = torch_cuda
    n, H, k = 200_000, 1 << 19, 30
    genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=11, n_blocks=[(50_000, 700)],
                              dups=[(120_000, 20_000, 3000)])
    g = genome.cpu().numpy().tobytes()
    assert set(g) <= set(b"ACGTN") and g[50_000:50_700] == b"N" * 700 and g[120_000:123_000] == g[20_000:23_000]
    keys = torch.zeros(H, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
    n_distinct, n_unique = qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    hk = keys.cpu().numpy().view(np.uint64)
    # CPU truth: canonical k-mers of the genome as one line, unique ones only, k-mer 0 excluded
    # (`search` has no uint16 run counter, QuicKmer.c:863: feed the count-style oracle encoder
    # overlapping pieces shorter than 65536 so that its wrap never triggers)
    step = 60_000
    km = np.concatenate([ol.extract(g[max(0, a - (k - 1)):a + step] + b"\n", k)[0] for a in range(0, n, step)])
    vals, cnts = np.unique(km[km != 0], return_counts=True)
    assert n_distinct == len(vals) and n_unique == int((cnts == 1).sum())
    assert np.array_equal(np.sort(hk[hk != 0]), vals[cnts == 1])
    # table obeys the probe rule: every stored key is found by the oracle's Find_hash
    L = ol.lib()
    import ctypes as C
    idx = C.c_uint64()
    for s in np.flatnonzero(hk)[::97]:
        assert L.qm2o_find(hk.ctypes.data, H, int(hk[s]), C.byref(idx), None) == 1 and idx.value == s
    # reads from it: deterministic, hit the dictionary, GPU count == oracle count
    n_reads, rl = 20_000, 150
    off = qm2.tools.read_offsets(n_reads, 0, rl, rl, 5)
    reads = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 5)
    torch.cuda.synchronize()
    blob = reads.cpu().numpy().tobytes()
    assert blob.count(b"\n") == n_reads and len(blob) == n_reads * (rl + 1)
    exp_depth, exp_st = ol.count_lines(hk, k, blob)
    assert exp_st["n_hits"] > 0.7 * exp_st["n_kmers"]
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr()) as c:
        c.submit_device(reads.data_ptr(), len(blob))
        depth, st = c.finish()
    assert st == exp_st and np.array_equal(depth, exp_depth)
    # variable-length (HiFi-like) reads
    off = qm2.tools.read_offsets(300, 7, 5000, 9000, 9)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    reads = torch.zeros(int(off[-1]), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), d_off.data_ptr(), 300, 7, 5000, 9000, 9,
                             sub_rate=0.001)
    torch.cuda.synchronize()
    blob = reads.cpu().numpy().tobytes()
    assert blob.count(b"\n") == 300
    exp_depth, exp_st = ol.count_lines(hk, k, blob)
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr()) as c:
        c.submit_device(reads.data_ptr(), len(blob))
        depth, st = c.finish()
    assert st == exp_st and np.array_equal(depth, exp_depth)


# ---- 6. full-size, size-independent properties (config C2 shape) -----------------

def test_c2_scale_properties(qm2, torch_cuda):
    """H=2^28 dictionary of a 50.8 Mb chr22-like genome, 2M x 150 bp reads:
    linearity, hit conservation, strand symmetry, batch independence, and an
    exact oracle comparison on a sample of the stream."""
    torch = torch_cuda
    n, H, k = 50_818_468, 1 << 28, 30
    genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=22,
                              n_blocks=[(0, 10_500_000), (30_000_000, 600_000), (50_300_000, 518_468)],
                              dups=[(40_000_000 + i * 100_000, 12_000_000 + i * 37_000, 1000 + 190 * i) for i in range(100)])
    keys = torch.zeros(H, dtype=torch.int64, device="cuda")
    scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
    n_distinct, n_unique = qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    assert 35_000_000 < n_unique <= n_distinct < 40_000_000
    n_reads, rl = 2_000_000, 150
    reads = torch.zeros(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 22)
    torch.cuda.synchronize()
    nbytes = reads.numel()
    counters = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()              # the library runs on its own stream
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), flags=qm2.FLAG_DIRECT) as cd:
        cd.submit_device(reads.data_ptr(), nbytes)
        d_direct, s_direct = cd.finish()
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=counters.data_ptr(),
                     flags=qm2.FLAG_TIMING) as c:
        c.submit_device(reads.data_ptr(), nbytes)
        d1, s1 = c.finish()
        assert c.profile()["partitioned_batches"] == 1          # this size takes the partitioned pipeline
        assert s1 == s_direct and np.array_equal(d1, d_direct)   # ... and agrees with the direct kernel
        assert s1["n_bases"] == n_reads * rl
        assert int(counters.sum(dtype=torch.int64)) == s1["n_hits"]           # every hit is one increment
        assert s1["n_kmers"] <= n_reads * (rl - k + 1) and s1["n_probes"] >= s1["n_kmers"]
        # linearity: the same stream again doubles every counter
        c.submit_device(reads.data_ptr(), nbytes)
        d2, s2 = c.finish()
        assert s2 == {a: 2 * b for a, b in s1.items()}
        assert np.array_equal(d2, (2 * d1.astype(np.uint32)).astype(np.uint16))
        # counts only land on occupied slots (or on the one empty slot k-mer 0 resolves to)
        occ = (keys != 0)
        stray = torch.count_nonzero(counters[~occ])
        assert int(stray) <= 1
        # batch independence: two halves cut at a line boundary, through the host pipeline
        c.reset()
        half = (n_reads // 2) * (rl + 1)
        host = reads.cpu().numpy().tobytes()
        c.submit(host[:half])
        c.submit(host[half:])
        d3, s3 = c.finish()
        assert s3 == s1 and np.array_equal(d3, d1)
    # exact oracle comparison on the first 100k reads
    hk = keys.cpu().numpy().view(np.uint64)
    sample = host[:100_000 * (rl + 1)]
    exp_depth, exp_st = ol.count_lines(hk, k, sample, threads=8)
    exp_depth = ol.remove_drain_padding(exp_depth, hk, exp_st["n_kmers"])
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr()) as c:
        c.submit(sample)
        depth, st = c.finish()
    assert st["n_kmers"] == exp_st["n_kmers"] and st["n_bases"] == exp_st["n_bases"]
    assert np.array_equal(depth, exp_depth)


# ---- 7. hg38-scale table (configs C3, C4, C5): H = 2^32, built once per module ---------------

@pytest.fixture(scope="module")
def hg38_scale(qm2, torch_cuda):
    """H = 2^32 dictionary of a 3.1 Gb synthetic reference (~2.27 G unique 30-mers, 53 % fill): device
    genome + keys, the direct kernel's counters for 5 M x 150 bp reads (the truth the other tests
    compare against), and the host copy of the keys for the oracle."""
    torch = torch_cuda
    free, total = torch.cuda.mem_get_info()
    if free < 150 * (1 << 30):
        pytest.skip("needs ~150 GiB of free HBM")
    n, H, k = 3_100_000_000, 1 << 32, 30
    genome = torch.empty(n, dtype=torch.uint8, device="cuda")
    qm2.tools.genome_generate(genome.data_ptr(), n, seed=38, n_blocks=[(0, 10_000_000), (1_500_000_000, 30_000_000)])
    keys = torch.empty(H, dtype=torch.int64, device="cuda")
    scratch = torch.empty(H, dtype=torch.int64, device="cuda")
    n_distinct, n_unique = qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    torch.cuda.empty_cache()
    assert 2.0e9 < n_unique <= n_distinct < 3.1e9
    n_reads, rl = 5_000_000, 150
    reads = torch.empty(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 38)
    a = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=a.data_ptr(),
                     flags=qm2.FLAG_DIRECT) as cd:
        cd.submit_device(reads.data_ptr(), reads.numel())
        _, sd = cd.finish(want_depth=False)
    hk = keys.cpu().numpy().view(np.uint64)          # 32 GiB on the host: the oracle's table
    env = dict(n=n, H=H, k=k, genome=genome, keys=keys, hk=hk, reads=reads, n_reads=n_reads, rl=rl,
               direct=a, direct_stats=sd, n_unique=n_unique)
    yield env
    env.clear()


def _oracle_sample_equal(qm2, env, sample: bytes, flags):
    """Exact equality with the oracle on `sample` (whole lines), all 2^32 counters and the statistics."""
    hk, k, H = env["hk"], env["k"], env["H"]
    exp_depth, exp_st = ol.count_lines(hk, k, sample, threads=8)
    exp_depth = ol.remove_drain_padding(exp_depth, hk, exp_st["n_kmers"])
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=env["keys"].data_ptr(), flags=flags) as c:
        c.submit(sample)
        depth, st = c.finish()
    assert st["n_kmers"] == exp_st["n_kmers"] and st["n_bases"] == exp_st["n_bases"]
    assert np.array_equal(depth, exp_depth)
    return st


def test_c3_scale_partitioned_equals_direct(qm2, torch_cuda, hg38_scale):
    """C3: the partitioned pipeline and the direct kernel must produce the same 4 G counters and the
    same statistics; the same stream twice doubles every counter; and both equal the ORACLE on a
    100 k-read sample of the stream (the only CPU-checkable size at H = 2^32)."""
    torch, env = torch_cuda, hg38_scale
    H, k, keys, reads, a, sd = env["H"], env["k"], env["keys"], env["reads"], env["direct"], env["direct_stats"]
    b = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=b.data_ptr(),
                     flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as cp:
        cp.submit_device(reads.data_ptr(), reads.numel())
        _, sp = cp.finish(want_depth=False)
        assert cp.profile()["partitioned_batches"] == 1 and cp.profile()["fallback_batches"] == 0
        assert sp == sd and sd["n_bases"] == env["n_reads"] * env["rl"] and sd["n_hits"] > 0.7 * sd["n_kmers"]
        assert torch.equal(a, b)
        assert int(b.sum(dtype=torch.int64)) == sp["n_hits"]
        cp.submit_device(reads.data_ptr(), reads.numel())
        _, s2 = cp.finish(want_depth=False)
        assert s2 == {x: 2 * y for x, y in sp.items()}
        assert torch.equal(b, 2 * a)
    del b
    torch.cuda.empty_cache()
    sample = reads[:100_000 * (env["rl"] + 1)].cpu().numpy().tobytes()
    for flags in (qm2.FLAG_PARTITION, qm2.FLAG_DIRECT):
        _oracle_sample_equal(qm2, env, sample, flags)


@pytest.mark.parametrize("batch_gib", [6, 2], ids=["library_default_6GiB", "cli_2GiB"])
def test_c3_partitioned_equals_direct_at_the_cli_batch_geometry(qm2, torch_cuda, hg38_scale, batch_gib):
    """The plan geometries the product really runs at hg38 scale — the library's default device batch, 6 GiB of
    sequence lines = 42.6 M x 150 bp reads (8192 level-1 sub-regions, 2^20 buckets, 58 + 64 + 13 GB of partition
    buffers with more than 2^32 k-mer places), and the 2 GiB batch the CLI asks for since round 4 (14.2 M reads: the
    same buckets a third as full, another group shape in bucket_count) — compared SLOT FOR SLOT with the direct
    kernel, which shares nothing with the partitioned pipeline but the front end (every k-mer probed exactly once,
    QuicKmer.c:283-293).  The smaller C3 tests run plans an eighth of the large one; bench.py makes the same
    comparison at its own batch (60 M reads) through tools.checksum_u32, which is checked against torch here."""
    torch, env = torch_cuda, hg38_scale
    H, k, keys, genome, n, rl = env["H"], env["k"], env["keys"], env["genome"], env["n"], env["rl"]
    free, _ = torch.cuda.mem_get_info()
    batch_bytes = batch_gib << 30
    if free < (190 if batch_gib == 6 else 100) * (1 << 30):
        pytest.skip("needs ~190 GiB of free HBM (two counter arrays, a 6 GiB batch, 135 GB of partition buffers)")
    n_reads = batch_bytes // (rl + 1)
    reads = torch.empty(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 7_000_000, rl, rl, 38)
    a = torch.zeros(H, dtype=torch.int32, device="cuda")
    b = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=a.data_ptr(),
                     flags=qm2.FLAG_DIRECT) as cd:
        cd.submit_device(reads.data_ptr(), reads.numel())
        _, sd = cd.finish(want_depth=False)
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=b.data_ptr(),
                     flags=qm2.FLAG_TIMING) as cp:
        cp.prepare(batch_bytes)                                   # what the CLI does before the dictionary streams in
        cp.submit_device(reads.data_ptr(), reads.numel())
        _, sp = cp.finish(want_depth=False)
        pr = cp.profile()
    assert pr["partitioned_batches"] == 1 and pr["fallback_batches"] == 0 and pr["direct_batches"] == 0
    assert sp == sd and sd["n_bases"] == n_reads * rl and sd["n_hits"] > 0.7 * sd["n_kmers"]
    assert torch.equal(a, b)
    ca, cb = qm2.tools.checksum_u32(a.data_ptr(), H), qm2.tools.checksum_u32(b.data_ptr(), H)
    assert ca == cb and ca[1] == sd["n_hits"] == int(sum(int(a[i:i + (1 << 28)].sum(dtype=torch.int64)) for i in range(0, H, 1 << 28)))
    assert ca[2] == int(sum(int((a[i:i + (1 << 28)] != 0).sum()) for i in range(0, H, 1 << 28)))
    # the checksum is position-weighted: one count moved to the neighbouring slot must change it
    i = int(torch.nonzero(a[:1 << 20])[0])
    b[i] -= 1
    b[i + 1] += 1
    cc = qm2.tools.checksum_u32(b.data_ptr(), H)
    assert cc[1] == ca[1] and cc[0] != ca[0]
    del a, b, reads
    torch.cuda.empty_cache()


def test_c3_scale_skew_overflows_the_sub_regions(qm2, torch_cuda, hg38_scale):
    """Two-level partition at hg38 scale with a skewed stream: 4 % poly-A / poly-T reads put 24 M copies of k-mer 0
    into ONE level-1 child, i.e. into its 8 per-XCD sub-regions (~90 k k-mers each at this batch size), which
    overflow into the spill list; a run of one repeated real read floods another child the same way.  Counters and
    statistics must equal the direct kernel's on the same bytes."""
    torch, env = torch_cuda, hg38_scale
    H, k, keys, reads, rl = env["H"], env["k"], env["keys"], env["reads"], env["rl"]
    poly = torch.frombuffer(bytearray((b"A" * rl + b"\n") * 150_000 + (b"T" * rl + b"\n") * 50_000), dtype=torch.uint8).cuda()
    one = reads[:rl + 1].repeat(100_000)                                        # the same 121 k-mers 100 k times
    blob = torch.cat([reads[:2_000_000 * (rl + 1)], poly, one, reads[2_000_000 * (rl + 1):3_000_000 * (rl + 1)]])
    out = []
    for flags in (qm2.FLAG_DIRECT, qm2.FLAG_PARTITION | qm2.FLAG_TIMING):
        cnt = torch.zeros(H, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=cnt.data_ptr(), flags=flags) as c:
            c.submit_device(blob.data_ptr(), blob.numel())
            _, st = c.finish(want_depth=False)
            if flags & qm2.FLAG_PARTITION:
                pr = c.profile()
                assert pr["partitioned_batches"] == 1 and pr["fallback_batches"] == 0     # spill list, not a recount
                assert pr["class_launches"]["probe"] >= 1
        out.append((cnt, st))
    assert out[0][1] == out[1][1]
    assert out[0][1]["n_bases"] == 3_300_000 * rl
    assert torch.equal(out[0][0], out[1][0])


def test_c5_hifi_reads_at_hg38_scale(qm2, torch_cuda, hg38_scale):
    """C5 at its stated shape: 50 k HiFi-like reads U(15 000, 20 000) bp (0.87 G bases, one line per
    read, QuicKmer.c:388) against the H = 2^32 dictionary: both pipelines equal, conservation,
    linearity, and exact oracle equality on the first 200 reads."""
    torch, env = torch_cuda, hg38_scale
    H, k, keys, genome, n = env["H"], env["k"], env["keys"], env["genome"], env["n"]
    n_reads, lo, hi = 50_000, 15_000, 20_000
    off = qm2.tools.read_offsets(n_reads, 500, lo, hi, 55)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    nbytes = int(off[-1])
    reads = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), d_off.data_ptr(), n_reads, 500, lo, hi, 55,
                             sub_rate=0.001)
    a = torch.zeros(H, dtype=torch.int32, device="cuda")
    b = torch.zeros(H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=a.data_ptr(),
                     flags=qm2.FLAG_DIRECT) as cd:
        cd.submit_device(reads.data_ptr(), nbytes)
        _, sd = cd.finish(want_depth=False)
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), ext_counters=b.data_ptr(),
                     flags=qm2.FLAG_PARTITION | qm2.FLAG_TIMING) as cp:
        cp.submit_device(reads.data_ptr(), nbytes)
        _, sp = cp.finish(want_depth=False)
        assert cp.profile()["partitioned_batches"] == 1 and cp.profile()["fallback_batches"] == 0
        assert sp == sd and sd["n_bases"] == nbytes - n_reads
        # kappa ~ 0.998 for 17.5 kb reads, minus the windows with a substitution-made N: the long-read path
        assert sd["n_kmers"] > 0.97 * sd["n_bases"] and sd["n_hits"] > 0.7 * sd["n_kmers"]
        assert torch.equal(a, b)
        assert int(b.sum(dtype=torch.int64)) == sp["n_hits"]
        cp.submit_device(reads.data_ptr(), nbytes)
        _, s2 = cp.finish(want_depth=False)
        assert s2 == {x: 2 * y for x, y in sp.items()}
        assert torch.equal(b, 2 * a)
    del a, b
    torch.cuda.empty_cache()
    sample = reads[:int(off[200])].cpu().numpy().tobytes()
    assert sample.count(b"\n") == 200 and min(len(x) for x in sample.split(b"\n")[:-1]) >= lo
    for flags in (qm2.FLAG_PARTITION, qm2.FLAG_DIRECT):
        st = _oracle_sample_equal(qm2, env, sample, flags)
        assert st["n_kmers"] > 0.97 * st["n_bases"]


def test_c4_eight_logical_shards_at_hg38_scale(qm2, torch_cuda, hg38_scale):
    """C4's shape without 8 GPUs: the H = 2^32 table cut into 8 hash-range
    shards; qm2_route_device buckets the k-mers of the C3 stream by owner into fixed slabs, every shard
    probes its slab against its own span (qm2_shard_span: owned range + the run of occupied slots past
    the middle-facing edge, QuicKmer.c:90-99) with the PARTITIONED pipeline in its own context, overhang
    counts are folded into the neighbour that owns those slots — and all 4 G counters, the hits and the
    probes equal the single-context result."""
    torch, env = torch_cuda, hg38_scale
    H, k, keys, reads, truth, sd = env["H"], env["k"], env["keys"], env["reads"], env["direct"], env["direct_stats"]
    G, per = 8, H // 8
    # qm2_shard_span only looks at the slots around the shard edges: give it a sparse host image
    W = 1 << 16
    sparse = np.zeros(H, dtype=np.uint64)            # untouched pages stay unmapped
    for g in range(1, G):
        e = g * per
        sparse[e - W:e + W] = keys[e - W:e + W].cpu().numpy().view(np.uint64)
    spans = [qm2.shard_span(sparse, G, g) for g in range(G)]
    for own_lo, own_hi, lo, hi in spans:
        assert own_lo - lo < W - 64 and hi - own_hi < W - 64      # every walk ended inside the copied window
    del sparse
    n_km = sd["n_kmers"]
    cap = (int(n_km / G * 1.05) + 65536) & ~1
    slabs = torch.empty(G * cap, dtype=torch.int64, device="cuda")
    fill = torch.zeros(G, dtype=torch.int32, device="cuda")
    call = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        router.route_device(reads.data_ptr(), reads.numel(), slabs.data_ptr(), cap, G, fill.data_ptr(), call.data_ptr())
        router.sync()
    assert [int(x) for x in call] == [sd["n_bases"], n_km, 0]
    fl = [int(x) for x in fill.cpu()]
    assert sum(fl) == n_km and max(fl) < cap
    full = torch.zeros(H, dtype=torch.int32, device="cuda")
    hits = probes = 0
    for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
        cnt = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr() + 8 * lo, ext_counters=cnt.data_ptr(),
                         span=(lo, hi), own=(own_lo, own_hi), flags=qm2.FLAG_TIMING) as pc:
            pc.probe_slabs_device(slabs.data_ptr() + 8 * g * cap, cap, 1, fill.data_ptr() + 4 * g)
            _, s = pc.finish(want_depth=False)
            prof = pc.profile()
        assert prof["partitioned_batches"] == 1 and prof["fallback_batches"] == 0    # a shard is counted like a table
        hits += s["n_hits"]
        probes += s["n_probes"]
        # owned slots and overhang alike: slot lo + i of the table (the fold of ShardedJob.finish)
        full[lo:hi] += cnt
        del cnt
    assert hits == sd["n_hits"] and probes == sd["n_probes"]
    assert torch.equal(full, truth)
    # the same with the senders' 10-bit route: 8 owners x 128 sub-slabs, every shard's 128 sub-slabs are the first level
    # of its partition (qm2_probe_subslabs_device): one split + count per shard instead of two splits + count
    del slabs, fill, full
    torch.cuda.empty_cache()
    n_subs = 1024 // G
    mean = n_km / 1024
    sub_cap = (int(mean * 1.12 + 8 * mean ** 0.5) + 256) & ~1
    slabs = torch.empty(1024 * sub_cap, dtype=torch.int64, device="cuda")
    fill = torch.zeros(1024, dtype=torch.int32, device="cuda")
    call.zero_()
    torch.cuda.synchronize()
    with qm2.Counter(None, hash_size=H, k=k, flags=qm2.FLAG_KEYS_LATER, span=(0, 64)) as router:
        router.route_device(reads.data_ptr(), reads.numel(), slabs.data_ptr(), sub_cap, 1024, fill.data_ptr(), call.data_ptr())
        router.sync()
    assert [int(x) for x in call] == [sd["n_bases"], n_km, 0]
    full = torch.zeros(H, dtype=torch.int32, device="cuda")
    hits = probes = 0
    for g, (own_lo, own_hi, lo, hi) in enumerate(spans):
        cnt = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr() + 8 * lo, ext_counters=cnt.data_ptr(),
                         span=(lo, hi), own=(own_lo, own_hi), flags=qm2.FLAG_TIMING) as pc:
            pc.probe_subslabs_device(slabs.data_ptr() + 8 * g * n_subs * sub_cap, sub_cap, 1, n_subs, fill.data_ptr() + 4 * g * n_subs,
                                     kmers_bound=n_subs * sub_cap)
            _, s = pc.finish(want_depth=False)
            prof = pc.profile()
        assert prof["partitioned_batches"] == 1 and prof["fallback_batches"] == 0 and prof["class_launches"]["split_kmer"] == 1
        hits += s["n_hits"]
        probes += s["n_probes"]
        full[lo:hi] += cnt
        del cnt
    assert hits == sd["n_hits"] and probes == sd["n_probes"]
    assert torch.equal(full, truth)
