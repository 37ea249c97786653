"""Deterministic synthetic inputs shared by the tests, the golden generator and
bench.py.  Pure numpy; no reference code involved.

c1_inputs() reproduces SURVEY.md Appendix C.1 verbatim (numpy seed 20261003) so
that the md5s recorded from the real reference can be regained anywhere.
"""
import os

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def revcomp_bytes(s: bytes) -> bytes:
    return _COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def write_fasta_reads(reads, header=b">") -> bytes:
    return b"".join(header + b"\n" + r + b"\n" for r in reads)


def write_fastq_reads(reads, tricky_quality=False) -> bytes:
    out = []
    for i, r in enumerate(reads):
        q = b"I" * len(r)
        if tricky_quality and len(r) > 0:
            # quality strings may legally start with '@' or '>'
            q = (b"@" if i % 3 == 0 else b">" if i % 3 == 1 else b"I") + q[1:]
        out.append(b"@r%d\n" % i + r + b"\n+\n" + q + b"\n")
    return b"".join(out)


def _write_ref(path, contigs):
    with open(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name + b"\n")
            s = seq.tobytes()
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + b"\n")


def _reads_from(rng, ref, n_reads, read_len, sub_rate, n_frac):
    reads = []
    for _ in range(n_reads):
        p = rng.integers(0, len(ref) - read_len)
        r = ref[p:p + read_len].copy()
        if rng.random() < 0.5:
            r = _COMP[r][::-1]
        e = rng.random(read_len) < sub_rate
        r[e] = ACGT[rng.integers(0, 4, e.sum())]
        if rng.random() < n_frac:
            r[rng.integers(0, read_len)] = ord("N")
        reads.append(r.tobytes())
    return reads


def c1_inputs(outdir):
    """Config C1: returns (ref.fa, ctrl.bed, reads.fa, reads.fq)."""
    rng = np.random.default_rng(20261003)

    def randseq(n):
        return ACGT[rng.integers(0, 4, n)]

    c1 = randseq(600_000).copy()
    c1[100_000:100_500] = ord("N")
    c1[300_000:305_000] = c1[200_000:205_000]
    c2 = randseq(400_000)
    fa = os.path.join(outdir, "ref.fa")
    _write_ref(fa, [(b"chrA", c1), (b"chrB", c2)])
    ctrl = os.path.join(outdir, "ctrl.bed")
    with open(ctrl, "wb") as f:
        f.write(b"chrA\t1000\t90000\nchrA\t400000\t590000\nchrB\t1000\t390000\n")
    ref = np.concatenate([c1, c2])
    reads = _reads_from(rng, ref, 100_000, 150, 0.005, 0.01)
    rfa = os.path.join(outdir, "reads.fa")
    rfq = os.path.join(outdir, "reads.fq")
    with open(rfa, "wb") as f:
        f.write(write_fasta_reads(reads))
    with open(rfq, "wb") as f:
        f.write(b"".join(b"@r%d\n" % i + r + b"\n+\n" + b"I" * 150 + b"\n"
                         for i, r in enumerate(reads)))
    return fa, ctrl, rfa, rfq


def mini_inputs(outdir):
    """A scaled-down C1: 2 contigs (40 kb + 20 kb), N gap, duplicated segment;
    6000 x 150 bp reads.  Small enough that the reference's dictionary for it
    (-s 128K, 1.5 MB) is committed under tests/golden/mini/."""
    rng = np.random.default_rng(20261004)
    a = ACGT[rng.integers(0, 4, 40_000)].copy()
    a[10_000:10_200] = ord("N")
    a[30_000:31_000] = a[20_000:21_000]
    b = ACGT[rng.integers(0, 4, 20_000)]
    fa = os.path.join(outdir, "ref.fa")
    _write_ref(fa, [(b"chrA", a), (b"chrB", b)])
    ctrl = os.path.join(outdir, "ctrl.bed")
    with open(ctrl, "wb") as f:
        f.write(b"chrA\t500\t9000\nchrA\t22000\t39000\nchrB\t500\t19000\n")
    ref = np.concatenate([a, b])
    reads = _reads_from(rng, ref, 6000, 150, 0.005, 0.01)
    rfa = os.path.join(outdir, "reads.fa")
    rfq = os.path.join(outdir, "reads.fq")
    with open(rfa, "wb") as f:
        f.write(write_fasta_reads(reads))
    with open(rfq, "wb") as f:
        f.write(write_fastq_reads(reads, tricky_quality=True))
    return fa, ctrl, rfa, rfq


def kat_reference():
    """5 kb random reference for the known-answer set."""
    rng = np.random.default_rng(7)
    return ACGT[rng.integers(0, 4, 5000)].copy()


def random_reads_blob(seed, n_reads, read_len, ref=None, sub_rate=0.005, n_frac=0.01):
    """Sequence-lines buffer (what qm2_submit takes): reads joined by '\\n'."""
    rng = np.random.default_rng(seed)
    if ref is None:
        ref = ACGT[rng.integers(0, 4, max(4 * read_len, 100_000))]
    reads = _reads_from(rng, ref, n_reads, read_len, sub_rate, n_frac)
    return b"".join(r + b"\n" for r in reads)


def kat_cases(ref):
    """name -> (kind, bytes)   kind in {'fa','fq'}"""
    rng = np.random.default_rng(77)
    rd = ref[2000:2150].tobytes()
    cases = {}
    one = ref[1000:1030].tobytes()
    cases["wrap70000"] = ("fa", b"".join(b">\n" + one + b"\n" for _ in range(70000)))
    # a control-region k-mer at depth 50000: its square overflows the `int` the reference squares it in (QuicKmer.c:507)
    cases["deep50000"] = ("fa", b"".join(b">\n" + one + b"\n" for _ in range(50000)))
    cases["upper"] = ("fa", b">r\n" + rd + b"\n")
    cases["lower"] = ("fa", b">r\n" + rd.lower() + b"\n")
    cases["crlf"] = ("fa", b">r\r\n" + rd + b"\r\n")
    rn = bytearray(rd); rn[75] = ord("n")
    rg = bytearray(rd); rg[75] = ord("G")
    rN = bytearray(rd); rN[75] = ord("N")
    cases["n_lower"] = ("fa", b">r\n" + bytes(rn) + b"\n")
    cases["g_upper"] = ("fa", b">r\n" + bytes(rg) + b"\n")
    cases["n_upper"] = ("fa", b">r\n" + bytes(rN) + b"\n")
    cases["split_lines"] = ("fa", b">r\n" + rd[:80] + b"\n" + rd[80:] + b"\n")
    cases["two_records"] = ("fa", b">a\n" + rd[:80] + b"\n>b\n" + rd[80:] + b"\n")
    cases["empty_lines"] = ("fa", b">r\n\n" + rd + b"\n\n\n>q\n" + rd[10:60] + b"\n")
    cases["short_reads"] = ("fa", b">a\n" + rd[:29] + b"\n>b\n" + rd[:30] + b"\n>c\nA\n>d\n" + rd[5:36] + b"\n")
    cases["polyA"] = ("fa", b">a\n" + b"A" * 200 + b"\n>t\n" + b"T" * 90 + b"\n>g\n" + b"G" * 64 + b"\n>c\n" + b"C" * 31 + b"\n")
    cases["junk_bytes"] = ("fa", b">j\n" + rd[:50] + b"*-.RYKM@>#" + rd[60:] + b"\x80\xff\x01 \t" + rd[:40] + b"\n")
    cases["headerless_first_line"] = ("fa", rd + b"\n>x\n" + rd[3:70] + b"\n")
    # long single lines built from the reference so that hits occur around the
    # uint16 run-length wrap (base 65536 of an N-free run)
    tile = ref.tobytes()
    for L in (20000, 65535, 65536, 65566, 70000, 99998):
        s = (tile * (L // len(tile) + 1))[:L]
        cases["long%d" % L] = ("fa", b">L\n" + s + b"\n")
    s = bytearray((tile * 20)[:99000]); s[1234] = ord("N"); s[70000] = ord("N")
    cases["long_with_N"] = ("fa", b">L\n" + bytes(s) + b"\n")
    # random read set with errors / Ns, as FASTA and FASTQ (quality lines that
    # start with '@' and '>' on purpose)
    reads = []
    for i in range(3000):
        L = int(rng.integers(25, 260))
        p = int(rng.integers(0, len(ref) - L))
        r = ref[p:p + L].copy()
        if rng.random() < 0.5:
            r = np.frombuffer(revcomp_bytes(r.tobytes()), dtype=np.uint8).copy()
        e = rng.random(L) < 0.01
        r[e] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(e.sum()))]
        if rng.random() < 0.05:
            r[int(rng.integers(0, L))] = ord("N")
        reads.append(r.tobytes())
    cases["reads_fa"] = ("fa", write_fasta_reads(reads))
    cases["reads_fq"] = ("fq", write_fastq_reads(reads, tricky_quality=True))
    return cases
