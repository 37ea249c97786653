"""CPU-only checks of the PACKED input format's host side (quick-mer2_amd/csrc/qm2_pack.c; include/qm2.h, "packed
sequence"): what qm2_pack_lines / qm2_pack_reads write is, position for position, the reference's reset test and
2-bit code (QuicKmer.c:403-411) of the text the reference's line discipline (:393-398, :451-455) hands to its encoder —
checked against a plain numpy statement of the format and the oracle's strip, in every instruction-set build (portable, AVX2, AVX-512, AVX-512 VBMI2).
No GPU call is made here; the device side of the format is pinned by tests/test_emul_cpu.py (unpack16 on the host)."""
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import synth


def pkg():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ol.ROOT, "quick-mer2_amd", "libqm2_hip.so")):
        ge.build()
    return ge.load_package()

ISAS = ["portable", "avx2", "avx512", "avx512vbmi2"]


@pytest.fixture(params=ISAS)
def isa(request):
    L = pkg().lib()
    got = L.qm2_pack_isa_select(request.param.encode()).decode()
    if got != request.param:
        L.qm2_pack_isa_select(None)
        pytest.skip("this CPU has no %s" % request.param)
    yield request.param
    L.qm2_pack_isa_select(None)


def unpack(blocks):
    """blocks (uint64[3 n]) -> (codes uint8[64 n], bad bool[64 n]) by the format's definition"""
    b = np.asarray(blocks, np.uint64).reshape(-1, 3)
    n = len(b)
    codes = np.zeros((n, 64), np.uint8)
    bad = np.zeros((n, 64), bool)
    for half in range(2):
        w = b[:, half]
        for q in range(32):
            hi = (w >> np.uint64(2 * q)) & np.uint64(1)
            lo = (w >> np.uint64(2 * q + 1)) & np.uint64(1)
            codes[:, 32 * half + q] = (hi * 2 + lo).astype(np.uint8)
    for p in range(64):
        bad[:, p] = ((b[:, 2] >> np.uint64(p)) & np.uint64(1)).astype(bool)
    return codes.reshape(-1), bad.reshape(-1)


def expect(lines: bytes):
    t = np.frombuffer(lines, np.uint8)
    return (t >> 1) & 3, (t == ord("N")) | (t == ord("\n"))


def model_wraps(lines: bytes):
    """qm2_plan.h find_wraps, line by line: the position of run position 65536 m of every N-free run"""
    out, at = [], 0
    for line in lines.split(b"\n")[:-1]:
        r0 = 0
        for run in line.split(b"N"):
            out += [at + r0 + m - 1 for m in range(65536, len(run) + 1, 65536)]
            r0 += len(run) + 1
        at += len(line) + 1
    return out


def check_lines(lines: bytes):
    P = pkg().pyqm2
    blocks, p, wraps = P.pack_lines(lines, wraps_cap=1 << 16)
    assert p.n_blocks == (len(lines) + 63) // 64
    codes, bad = unpack(blocks)
    ec, eb = expect(lines)
    n = len(lines)
    assert np.array_equal(bad[:n], eb)
    assert bad[n:].all()                                             # the filler is all resets
    assert np.array_equal(codes[:n][~eb], ec[~eb])                   # (the code bits of a reset are never looked at)
    assert p.n_newlines == lines.count(b"\n") + 64 * p.n_blocks - n
    assert list(wraps) == model_wraps(lines)
    return blocks, p


def test_pack_lines_is_the_format(isa):
    rng = np.random.default_rng(5)
    check_lines(b"")
    check_lines(b"\n")
    check_lines(b"ACGT\n")
    check_lines(b"ACGTNNacgtn\r\x00\xff>@+\n" * 7)
    for n_lines in (1, 3, 50):
        for _ in range(6):
            parts = []
            for _ in range(n_lines):
                L = int(rng.integers(0, 300))
                parts.append(rng.integers(0, 256, L, dtype=np.uint8).tobytes().replace(b"\n", b"N") + b"\n")
            check_lines(b"".join(parts))
    # every length around the block size
    base = rng.integers(0, 256, 400, dtype=np.uint8).tobytes().replace(b"\n", b"C")
    for n in list(range(0, 70)) + [127, 128, 129, 191, 192, 193]:
        check_lines(base[:n] + b"\n")


def test_pack_lines_run_counter_wraps(isa):
    rng = np.random.default_rng(6)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    long1 = acgt[rng.integers(0, 4, 70000)].tobytes()
    long2 = bytearray(acgt[rng.integers(0, 4, 200000)].tobytes())
    long2[100] = ord("N")
    long2[140000] = ord("N")
    lines = b"ACGT\n" + long1 + b"\n" + b"TTTT\n" * 3 + bytes(long2) + b"\n" + acgt[rng.integers(0, 4, 65535)].tobytes() + b"\n" + \
        acgt[rng.integers(0, 4, 65536)].tobytes() + b"\n"
    _, p = check_lines(lines)
    assert p.n_wraps == len(model_wraps(lines)) and p.n_wraps >= 4
    # a wrap list that is too small is refused (the caller takes the ASCII path), not truncated
    L = pkg().lib()
    P = pkg().pyqm2
    out = np.zeros(L.qm2_packed_bound(len(lines)) // 8, np.uint64)
    w = np.zeros(2, np.uint64)
    pk = P.Packed()
    src = np.frombuffer(lines, np.uint8)
    import ctypes as C
    assert L.qm2_pack_lines(src.ctypes.data, len(lines), out.ctypes.data, out.nbytes, w.ctypes.data, 2, C.byref(pk)) == 2
    assert L.qm2_pack_lines(src.ctypes.data, len(lines), out.ctypes.data, 24, w.ctypes.data, 2, C.byref(pk)) == 2   # nor an output too small
    assert L.qm2_pack_lines(src.ctypes.data, len(lines) - 1, out.ctypes.data, out.nbytes, w.ctypes.data, 2, C.byref(pk)) == 1   # whole lines only


def model_strip(text: bytes, fastq, skip):
    """The reference's line discipline from a block boundary on (QuicKmer.c:398, :451-455; csrc/qm2_reader.c take_line)."""
    out, desync = [], False
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for ln in lines:
        if skip:
            skip -= 1
            continue
        if ln[:1] == b">":
            desync = desync or bool(fastq)
            continue
        out.append(ln + b"\n")
        if fastq:
            skip = 3
    return b"".join(out), skip, desync


def check_reads(text: bytes, fastq, skip):
    P = pkg().pyqm2
    blocks, p, wraps, st = P.pack_reads(text, fastq, skip, wraps_cap=1 << 16)
    lines, skip_after, desync = model_strip(text, fastq, skip)
    codes, bad = unpack(blocks)
    ec, eb = expect(lines)
    n = len(lines)
    assert p.n_blocks == (n + 63) // 64
    assert np.array_equal(bad[:n], eb) and bad[n:].all()
    assert np.array_equal(codes[:n][~eb], ec[~eb])
    assert p.n_newlines == lines.count(b"\n") + 64 * p.n_blocks - n
    assert (st.skip, bool(st.desync)) == (skip_after, desync)
    assert st.lines == text.count(b"\n") + (1 if text and not text.endswith(b"\n") else 0)     # whatever the state on entry
    assert list(wraps) == model_wraps(lines)
    return lines


def fastq_text(rng, n_reads, lo=1, hi=260, weird=False):
    acgt = np.frombuffer(b"ACGTN", np.uint8)
    recs = []
    for i in range(n_reads):
        L = int(rng.integers(lo, hi))
        seq = acgt[rng.integers(0, 5 if i % 7 == 0 else 4, L)].tobytes()
        qual = rng.integers(33, 74, L, dtype=np.uint8).tobytes()
        if weird and i % 5 == 0:
            qual = b"@" + qual[1:]                       # quality lines may start with '@' or '>'
        if weird and i % 11 == 0:
            qual = b">" + qual[1:]
        recs.append(b"@r%d some text\n" % i + seq + b"\n+\n" + qual + b"\n")
    return b"".join(recs)


def test_pack_reads_fastq(isa):
    rng = np.random.default_rng(7)
    text = fastq_text(rng, 400, weird=True)
    body = text[text.index(b"\n") + 1:]                  # the first header is consumed by the format sniff (:395)
    got = check_reads(body, True, 0)
    assert got.count(b"\n") == 400
    # any block boundary, with the state the line count gives it
    lines = body.split(b"\n")[:-1]
    for cut in (1, 2, 3, 4, 5, 177, 801, 1598):
        skip = (4 - (cut & 3)) & 3                        # qm2_reader.c pf_state
        check_reads(b"\n".join(lines[cut:]) + b"\n", True, skip)
    # a sequence-position line that starts with '>' is skipped WITHOUT the three discards (:398 before :451)
    odd = b"ACGT\n+\nIIII\n@r\n>weird\n+\nIIII\n@r2\nGGGG\n+\nIIII\n"
    check_reads(odd, True, 0)
    _, _, _, st = pkg().pyqm2.pack_reads(odd, True, 0)
    assert st.desync == 1
    # empty sequence lines, a last line without its newline, a block that is one line
    check_reads(b"\n+\n\n@r\nAC\n+\nII\n@r\nNNNN", True, 0)
    check_reads(b"ACGTACGTAC", True, 0)
    check_reads(b"ACGTACGTAC", True, 2)
    check_reads(b"", True, 1)


def test_pack_reads_fasta(isa):
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGTacgtN", np.uint8)
    parts = []
    for i in range(300):
        if i % 3 == 0:
            parts.append(b">chr%d description\n" % i)
        L = int(rng.integers(0, 200))
        parts.append(acgt[rng.integers(0, 9, L)].tobytes() + b"\n")
    text = b"".join(parts)
    check_reads(text, False, 0)
    check_reads(text + b"ACGTNACGT", False, 0)            # no final newline: supplied
    check_reads(b">only a header\n", False, 0)
    check_reads(b"\n\n\n", False, 0)
    # long lines: the run-counter wraps of taken lines only, as packed positions
    long_line = acgt[rng.integers(0, 4, 150000)].tobytes()
    check_reads(b">h\n" + long_line + b"\nAC\n>" + long_line + b"\n" + long_line[:70000] + b"N" + long_line[:66000] + b"\n", False, 0)


def test_pack_reads_matches_the_oracles_strip(isa):
    """whole files of the reference vectors: pack_reads of everything behind the sniffed first line == pack_lines of
    what the oracle's restatement of :393-398 / :451-455 strips them to"""
    P = pkg().pyqm2
    with tempfile.TemporaryDirectory() as td:
        fa, ctrl, rfa, rfq = synth.mini_inputs(td)
        for path in (rfa, rfq):
            text = open(path, "rb").read()
            fastq = text[:1] == b"@"
            body = text[text.index(b"\n") + 1:] if fastq else text
            b1, p1, w1, _ = P.pack_reads(body, fastq, 0)
            b2, p2, w2 = P.pack_lines(ol.strip_reads(path))
            assert np.array_equal(b1, b2) and (p1.n_blocks, p1.n_newlines, p1.n_wraps) == (p2.n_blocks, p2.n_newlines, p2.n_wraps)


def test_pack_reads_output_too_small_is_refused():
    P = pkg().pyqm2
    with pytest.raises(P.Qm2Error):
        P.pack_reads(b"ACGT\n" * 100, False, 0, room=48)


def test_the_three_builds_agree():
    L = pkg().lib()
    P = pkg().pyqm2
    rng = np.random.default_rng(9)
    text = fastq_text(rng, 2000, weird=True)
    body = text[text.index(b"\n") + 1:]
    lines, _, _ = model_strip(body, True, 0)
    outs = []
    for name in ISAS:
        if L.qm2_pack_isa_select(name.encode()).decode() != name:
            continue
        b1, p1, _, _ = P.pack_reads(body, True, 0)
        b2, p2, _ = P.pack_lines(lines)
        assert np.array_equal(b1, b2)
        outs.append((b1.tobytes(), p1.n_newlines))
    L.qm2_pack_isa_select(None)
    assert len(set(outs)) == 1 and len(outs) >= 1


def test_pack_reads_fuzz_against_the_line_discipline(isa):
    """byte soup with newlines, '>' and '@' wherever they fall — FASTA and FASTQ from every line state, desynchronising
    lines included (the model follows :398 / :451-455 literally): position for position the model's strip, packed"""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTNacgtn>@+\n\n\n\r\x00\xff", np.uint8)
    for trial in range(120):
        n = int(rng.integers(0, 3000))
        text = alphabet[rng.integers(0, len(alphabet), n)].tobytes()
        if trial % 3 == 0:
            text += b"\n"
        for fastq, skip in ((False, 0), (True, int(rng.integers(0, 4)))):
            check_reads(text, fastq, skip)
    # long stretches without a newline across many 64-byte groups, group-aligned and not
    for n in (63, 64, 65, 127, 128, 129, 4096, 4097):
        body = alphabet[rng.integers(0, 10, n)].tobytes()
        for lead in (b"", b"\n", b">x\n", b"AC\n+\nII\n@h\n"):
            check_reads(lead + body + b"\n" + body[:7], True, 0)
            check_reads(lead + body, False, 0)
