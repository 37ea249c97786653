"""CPU-only checks of the product boundary: libqm2_hip.so loads, exports every
symbol include/qm2.h and include/qm2_tools.h declare, and fails LOUDLY (no CPU
fallback) when no GPU is usable.  No compute calls are made here."""
import os
import re
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "quick-mer2_amd", "libqm2_hip.so")):
        ge.build()
    return ge.load_package()


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qm2t?_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(pkg):
    L = pkg.lib()
    names = declared("qm2.h") + declared("qm2_tools.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(pkg.pyqm2.ABI_SYMBOLS) == declared("qm2.h")
    assert sorted(pkg.pyqm2.TOOL_SYMBOLS) == declared("qm2_tools.h")
    assert L.qm2_abi_version() == 6
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "quick-mer2_amd", "libqm2_hip.so")],
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = set(re.findall(r" T (qm2t?_[a-z_0-9]+)", out))
    assert set(names) <= exported


def test_shards_library_exports_its_header(pkg):
    """libqm2_shards.so (config 4 from the C host: include/qm2_shards.h) exports every symbol its header declares and
    is built on the C-ABI of libqm2_hip.so + RCCL; the plain CLI does not link it (it is loaded by --shards)."""
    lib = os.path.join(ROOT, "quick-mer2_amd", "libqm2_shards.so")
    assert os.path.exists(lib), "make -C quick-mer2_amd builds it"
    names = declared("qm2_shards.h")
    assert sorted(names) == ["qm2_shards_acquire", "qm2_shards_close", "qm2_shards_commit", "qm2_shards_finish", "qm2_shards_get_info",
                             "qm2_shards_open", "qm2_shards_result", "qm2_shards_submit"]
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert set(names) <= set(re.findall(r" T (qm2_[a-z_0-9]+)", out))
    needed = subprocess.run(["readelf", "-d", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "libqm2_hip.so" in needed and "librccl" in needed
    cli = subprocess.run(["readelf", "-d", os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "librccl" not in cli and "libqm2_shards" not in cli


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the loud-failure path cannot be shown")
    with pytest.raises(pkg.pyqm2.Qm2Error) as e:
        pkg.Counter(np.zeros(1 << 10, dtype=np.uint64), k=30)
    assert "no usable HIP device" in str(e.value) and "no CPU fallback" in str(e.value)


def test_cli_fails_loudly_without_gpu(pkg, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    cli = os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")
    r = subprocess.run([cli, "count", os.path.join(HERE, "golden", "kat", "ref.fa"),
                        os.path.join(HERE, "golden", "kat", "upper.fa"), str(tmp_path / "o")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode != 0
    assert b"no usable HIP device" in r.stdout
    assert not os.path.exists(str(tmp_path / "o.bin"))
    # usage / dispatch text (QuicKmer.c:1496-1519 surface, count only)
    r = subprocess.run([cli], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"count" in r.stdout
    r = subprocess.run([cli, "count", "-h"], stdout=subprocess.PIPE)
    assert b"quicKmer2 count [Options] ref.fa sample.fast[a/q] Out_prefix" in r.stdout


def test_product_never_references_the_oracle():
    """The product tree must not import, link or call anything under oracle/."""
    bad = []
    for dp, dn, fn in os.walk(os.path.join(ROOT, "quick-mer2_amd")):
        for f in fn:
            if f.endswith((".so", ".pyc")) or f == "quicKmer2":
                continue
            txt = open(os.path.join(dp, f), errors="replace").read()
            if re.search(r"qm2o_|oracle_lib|libqm2_oracle|oracle/", txt):
                if f == "multi.py" and not re.search(r"qm2o_|oracle_lib|libqm2_oracle|import oracle", txt):
                    continue       # multi.py mentions the word in a docstring only
                bad.append(f)
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "quick-mer2_amd", "quicKmer2")], stdout=subprocess.PIPE).stdout
    assert b"oracle" not in out


def test_host_reader_matches_reference_line_discipline(tmp_path):
    """qm2_reader.c vs the oracle's restatement of QuicKmer.c:393-398/:451-455, via a tiny
    host-only harness (no GPU): the stripped stream must be identical."""
    src = os.path.join(ROOT, "quick-mer2_amd", "csrc")
    harness = tmp_path / "h.c"
    harness.write_text('''
#include <stdio.h>
#include <stdlib.h>
#include "qm2_host.h"
static size_t CAP;
static FILE *O;
static int acq(void *u, char **b, size_t *c) { (void)u; *b = malloc(CAP); *c = CAP; return *b == NULL; }
static int com(void *u, char *b, size_t n) { (void)u; fwrite(b, 1, n, O); free(b); return 0; }
static int sub(void *u, const char *p, size_t n) { (void)u; fwrite(p, 1, n, O); return 0; }
static int raw(void *u, char *b, size_t off, size_t n, int fastq, unsigned skip) { (void)u;
  if (n) { unsigned hdr[3] = {(unsigned)n, (unsigned)fastq, skip}; fwrite(hdr, 4, 3, O); fwrite(b + off, 1, n, O); }
  free(b); return 0; }
static int pk(void *u, char *b, const qm2_packed *p, const uint64_t *w) { (void)u;
  unsigned long long hdr[3] = {~0ull - p->n_blocks, p->n_newlines, p->n_wraps}; fwrite(hdr, 8, 3, O); fwrite(b, 24, p->n_blocks, O); fwrite(w, 8, p->n_wraps, O);
  free(b); return 0; }
/* packed mode: ASCII pieces (the serial parts of the reader) are framed too, so that the order is kept */
static int com_f(void *u, char *b, size_t n) { (void)u; if (n) { unsigned long long hdr[3] = {n, 0, 0}; fwrite(hdr, 8, 3, O); fwrite(b, 1, n, O); } free(b); return 0; }
static int sub_f(void *u, const char *p, size_t n) { (void)u; unsigned long long hdr[3] = {n, 0, 0}; fwrite(hdr, 8, 3, O); fwrite(p, 1, n, O); return 0; }
int main(int argc, char **argv) { O = fopen(argv[2], "wb"); CAP = (size_t)atol(argv[3]);
  qm2_reader_note_affinity();   /* (as the CLI does: the block reader then looks where the file's cached pages lie) */
  const int packed = argc > 5 && argv[5][0] == 'p';
  qm2_lines_out out = {acq, packed ? com_f : com, packed ? sub_f : sub, NULL, argc > 5 && !packed ? raw : NULL, packed ? pk : NULL};
  int rc = qm2_stream_reads(argv[1], &out, atoi(argv[4])); fclose(O); return rc; }
''')
    exe = str(tmp_path / "h")
    subprocess.run(["gcc", "-O1", "-pthread", "-I", src, "-o", exe, str(harness),
                    os.path.join(src, "qm2_reader.c"), os.path.join(src, "qm2_pack.c")], check=True)

    def runs_of(codes, bad):
        """What of a position stream the k-mers depend on, whatever order its pieces came in: the multiset of its runs of
        non-reset positions (a packed piece does not tell a line end from an 'N', ends in filler, and the reader's workers
        hand their pieces over as they fill up)."""
        good = np.flatnonzero(~bad)
        if len(good) == 0:
            return []
        cut = np.flatnonzero(np.diff(good) > 1) + 1
        c = codes[good].astype(np.uint8)
        return sorted(x.tobytes() for x in np.split(c, cut))

    def packed_content(blob):
        import test_pack_cpu as tp
        runs, at, n_newlines, n_pos = [], 0, 0, 0
        while at < len(blob):
            a, nl, nw = np.frombuffer(blob, dtype=np.uint64, count=3, offset=at)
            at += 24
            if int(a) > (1 << 62):                      # a packed piece
                nb = (1 << 64) - 1 - int(a)
                c, b = tp.unpack(np.frombuffer(blob, dtype=np.uint64, count=3 * nb, offset=at))
                at += 24 * nb + 8 * int(nw)
                n_newlines += int(nl)
            else:                                       # ASCII lines
                c, b = tp.expect(blob[at:at + int(a)])
                n_newlines += blob[at:at + int(a)].count(b"\n")
                at += int(a)
            runs += runs_of(c, b)
            n_pos += len(c)
        return sorted(runs), n_pos - n_newlines

    import oracle_lib as ol
    import synth
    cases = synth.kat_cases(synth.kat_reference())
    extra = {
        "no_final_newline": ("fa", b">a\nACGTACGT\nGGGG"),
        "fq_truncated": ("fq", b"@a\nACGT\n+\nIIII\n@b\nGG"),
        "gt_in_fastq_seq_pos": ("fq", b"@a\nACGT\n+\nIIII\n>odd\nTTTT\n+\nIIII\n@c\nCCCC\n+\nIIII\n"),
        "empty_seq_line": ("fq", b"@a\n\n+\n\n@b\nACGT\n+\nIIII\n"),
        # blocks are 32 MiB with a 1 MiB carry reserve: lines that straddle blocks, a line longer
        # than the reserve, and one longer than a whole block
        "straddle": ("fa", b">x\n" + b"ACGT" * ((33 << 20) // 4 // 150 * 37) + b"\n" + (b"ACGTTGCA" * 19 + b"\n") * 250000),
        "huge_line": ("fa", b">h\n" + b"ACGTA" * (8 << 20) + b"\nACGT\n"),
        # 100 MB FASTQ = 4 blocks: exercises the newline-count state hand-over between jobs
        "fq_blocks": ("fq", b"".join(b"@r%d\n" % i + b"ACGTTGCATG" * 15 + b"\n+\n" + b"I" * 150 + b"\n" for i in range(320000))),
        # ... and the same with a '>' at a sequence position in the second block (QuicKmer.c:398
        # skips it WITHOUT the three discards): every later block must be redone serially
        # records whose lines cannot be told apart by their first bytes ('@' header of one byte, a sequence that starts with '@' —
        # any byte is a base, :411 —, '+', qualities that start with '+'): the packed block reader's GUESS of a block's line
        # state is wrong for some blocks, and the line count of the blocks before must put it right
        "fq_ambiguous": ("fq", b"".join(b"@\n@" + b"ACGTTGCATG" * 14 + b"\n+\n+" + b"I" * 140 + b"\n" for i in range(40000))),
        "fq_desync": ("fq", b"".join(b"@r%d\n" % i + (b">oops" if i == 150000 else b"ACGTTGCATG" * 15) + b"\n+\n" + b"I" * 150 + b"\n"
                                     for i in range(320000))),
    }
    for name, (kind, data) in list(cases.items()) + list(extra.items()):
        p = tmp_path / ("%s.%s" % (name, kind))
        p.write_bytes(data)
        exp = ol.strip_reads(str(p), seekable=True)
        for batch, thr in ((64, 0), (4096, 0), (1 << 20, 0), (64 << 20, 1), (64 << 20, 3), (64 << 20, 8),
                           (1 << 20, 3), (40 << 20, 2)):    # staging buffers smaller than a block / than one line
            outp = str(tmp_path / "out.bin")
            # threaded: the default choice (pread blocks for FASTQ files, read(2) stream otherwise), then each
            # path forced for both formats; blocks are committed in order, so even the order is the serial one
            for env in ({},) if thr == 0 else ({}, {"QM2_READER_BLOCKS": "1"}, {"QM2_READER_STREAM": "1"}):
                subprocess.run([exe, str(p), outp, str(batch), str(thr)], check=True, env=dict(os.environ, **env))
                assert open(outp, "rb").read() == exp, (name, batch, thr, env)
        # packed hand-over (the CLI's default since ABI 6): the workers strip AND pack; what arrives must hold the
        # oracle's stripped stream, position for position (modulo filler), and its base count
        import test_pack_cpu as tp
        ec, eb = tp.expect(exp)
        want, want_bases = runs_of(ec, eb), len(exp) - exp.count(b"\n")
        big = len(data) > (8 << 20)
        for batch, thr in ((64 << 20, 3), (40 << 20, 8)) if big else ((64 << 20, 1), (64 << 20, 3), (64 << 20, 8), (40 << 20, 2)):
            for env in ({}, {"QM2_READER_STREAM": "1"}, {"QM2_BLOCK_BYTES": "65536"}) + (() if big else ({"QM2_READER_OLD_BLOCKS": "1"},)):
                outp = str(tmp_path / "packed.bin")
                r = subprocess.run([exe, str(p), outp, str(batch), str(thr), "packed"], env=dict(os.environ, QM2_READER_TIMING="1", **env), stdout=subprocess.PIPE)
                if name == "fq_ambiguous" and env == {"QM2_BLOCK_BYTES": "65536"}:
                    assert int(r.stdout.decode().split(" blocks packed again")[0].split()[-1]) > 0, r.stdout   # the guess WAS wrong somewhere
                if r.returncode == 3:
                    # a '>' at a FASTQ sequence position: the block reader's pieces are in no order, it gives the input up
                    # (the CLI then starts over on the in-order path); the streaming arrangement follows it serially
                    assert name == "fq_desync" and "QM2_READER_STREAM" not in env and "QM2_READER_OLD_BLOCKS" not in env, (name, env)
                    continue
                assert r.returncode == 0, (name, thr, env)
                assert not (name == "fq_desync" and ("QM2_READER_STREAM" not in env and "QM2_READER_OLD_BLOCKS" not in env)), (name, env)
                got, got_bases = packed_content(open(outp, "rb").read())
                assert got_bases == want_bases, (name, thr, env)
                assert got == want, (name, thr, env)
        # raw hand-over (QM2_READER=raw): blocks of untouched text + the line state at their
        # first byte; applying the device's rule (tests restate it here) to each block must give `exp`
        if name not in ("fq_desync", "gt_in_fastq_seq_pos"):
            for batch, thr in ((64 << 20, 1), (64 << 20, 3), (64 << 20, 8)):
                outp = str(tmp_path / "raw.bin")
                r = subprocess.run([exe, str(p), outp, str(batch), str(thr), "raw"])
                blob = open(outp, "rb").read()
                if r.returncode == 2:             # a line that does not fit a buffer: the caller starts over
                    assert name == "huge_line" or max(len(l) for l in data.split(b"\n")) > (32 << 20), name
                    continue
                assert r.returncode == 0, name
                kept, at = [], 0
                while at < len(blob):
                    n, fastq, skip = np.frombuffer(blob, dtype=np.uint32, count=3, offset=at)
                    text = blob[at + 12:at + 12 + int(n)]
                    at += 12 + int(n)
                    lines = text.split(b"\n")
                    if lines and lines[-1] == b"":
                        lines.pop()
                    for i, l in enumerate(lines):
                        if fastq:
                            if i >= skip and (i - skip) % 4 == 0 and not l.startswith(b">"):
                                kept.append(l)
                        elif not l.startswith(b">"):
                            kept.append(l)
                assert b"".join(l + b"\n" for l in kept) == exp, (name, thr)
        # pipe: first line is lost unless FASTQ (QuicKmer.c:396)
        outp = str(tmp_path / "out_pipe.bin")
        with open(str(p), "rb") as f:
            pr = subprocess.Popen(["cat"], stdin=f, stdout=subprocess.PIPE)
            subprocess.run([exe, "/dev/fd/0", outp, str(64 << 20), "2"], stdin=pr.stdout, check=True)
            pr.wait()
        assert open(outp, "rb").read() == ol.strip_reads(str(p), seekable=False), name


def test_dispatcher_deals_blocks_in_turns(tmp_path):
    """qm2_dispatch.c (`quicKmer2 count --devices a,b,..`): the one reader's buffers go to the sinks in
    turns of a fixed number of bytes, every buffer is committed to the sink that lent it, a sink is
    flushed when its turn ends, and together the sinks receive exactly the reader's stream."""
    src = os.path.join(ROOT, "quick-mer2_amd", "csrc")
    harness = tmp_path / "d.c"
    harness.write_text('''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qm2_dispatch.h"
#include "qm2_host.h"
typedef struct { int id; FILE *f; long lent, flushes, bytes; } Sink;
static FILE *LOG;
static int acq(void *u, char **b, size_t *c) { Sink *s = u; *c = 4096; *b = malloc(*c + 8); memcpy(*b, &s->id, sizeof s->id); *b += 8; s->lent++; return 0; }
static int com(void *u, char *b, size_t n) { Sink *s = u; int id; memcpy(&id, b - 8, sizeof id);
  if (id != s->id) { puts("commit to the wrong sink"); return 1; }
  fwrite(b, 1, n, s->f); fprintf(LOG, "%d %zu\\n", s->id, n); s->bytes += (long)n; s->lent--; free(b - 8); return 0; }
static int sub(void *u, const char *p, size_t n) { Sink *s = u; fwrite(p, 1, n, s->f); fprintf(LOG, "%d %zu\\n", s->id, n); s->bytes += (long)n; return 0; }
static int flu(void *u) { Sink *s = u; s->flushes++; fprintf(LOG, "%d flush\\n", s->id); return 0; }
int main(int argc, char **argv) {
  int n = atoi(argv[2]); size_t turn = (size_t)atol(argv[3]); Sink S[8]; qm2_sink sinks[8]; char name[512];
  LOG = fopen(argv[4], "w");
  for (int i = 0; i < n; i++) { snprintf(name, sizeof name, "%s.%d", argv[4], i); S[i].id = i; S[i].f = fopen(name, "wb"); S[i].lent = S[i].flushes = S[i].bytes = 0;
    sinks[i].acquire = acq; sinks[i].commit = com; sinks[i].submit = sub; sinks[i].flush = flu; sinks[i].user = &S[i]; sinks[i].commit_raw = NULL; sinks[i].commit_packed = NULL; }
  qm2_dispatch *d = qm2_dispatch_new(sinks, n, turn);
  qm2_lines_out out = {qm2_dispatch_acquire, qm2_dispatch_commit, qm2_dispatch_submit, d, NULL, NULL};
  int rc = qm2_stream_reads(argv[1], &out, atoi(argv[5]));
  qm2_dispatch_free(d);
  for (int i = 0; i < n; i++) { if (S[i].lent) rc = 3; fclose(S[i].f); }
  fclose(LOG); return rc; }
''')
    exe = str(tmp_path / "d")
    subprocess.run(["gcc", "-O1", "-pthread", "-I", src, "-o", exe, str(harness), os.path.join(src, "qm2_reader.c"),
                    os.path.join(src, "qm2_dispatch.c"), os.path.join(src, "qm2_pack.c")], check=True)
    import oracle_lib as ol
    import synth
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    exp = ol.strip_reads(rfq)
    for n_sinks, turn, thr in ((1, 10_000, 0), (2, 10_000, 0), (3, 50_000, 2), (4, 1, 3), (2, 1 << 30, 1)):
        log = str(tmp_path / "log")
        subprocess.run([exe, rfq, str(n_sinks), str(turn), log, str(thr)], check=True)
        parts = [open("%s.%d" % (log, i), "rb").read() for i in range(n_sinks)]
        # every sink got whole lines, and all of them together are the reader's stream (as a multiset of lines)
        assert all(p == b"" or p.endswith(b"\n") for p in parts)
        assert sorted(b"".join(parts).split(b"\n")) == sorted(exp.split(b"\n"))
        events = [l.split() for l in open(log).read().splitlines()]
        if n_sinks > 1 and turn < len(exp):
            assert all(len(p) > 0 for p in parts)                           # everybody had a turn
            # a sink is flushed once it has been given its share, and then it is the next one's turn
            share, cur = 0, 0
            for e in events:
                if e[1] == "flush":
                    assert int(e[0]) == cur and share >= turn
                    cur, share = (cur + 1) % n_sinks, 0
                elif int(e[0]) == cur:
                    share += int(e[1])
        else:
            assert len(parts[0]) == len(exp) or n_sinks > 1


def test_dispatcher_deals_raw_blocks_in_consecutive_turns(tmp_path):
    """`--devices` with the line discipline on the devices: the reader's raw blocks go to the sinks in turns of a fixed
    number of BLOCKS, decided when a buffer is acquired; a sink is flushed behind the last block of its turn.  What a
    device then does with a batch — number its lines from the state at its first block on — must give exactly the
    reader's sequence lines: restated here per batch (FASTQ: line L is a sequence line iff L >= skip and (L - skip) % 4
    == 0, QuicKmer.c:451-455; FASTA: iff it does not start with '>', :398) and compared with the oracle's strip."""
    src = os.path.join(ROOT, "quick-mer2_amd", "csrc")
    harness = tmp_path / "r.c"
    harness.write_text('''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qm2_dispatch.h"
#include "qm2_host.h"
typedef struct { int id; long lent; } Sink;
static FILE *LOG, *DATA;
static int acq(void *u, char **b, size_t *c) { Sink *s = u; *c = 1 << 20; *b = malloc(*c + 8); memcpy(*b, &s->id, sizeof s->id); *b += 8; s->lent++; return 0; }
static int own(Sink *s, char *b) { int id; memcpy(&id, b - 8, sizeof id); if (id != s->id) { puts("commit to the wrong sink"); return 1; } return 0; }
static int com(void *u, char *b, size_t n) { Sink *s = u; if (own(s, b)) return 1; if (n) { puts("stripped lines in raw mode"); return 1; } s->lent--; free(b - 8); return 0; }
static int raw(void *u, char *b, size_t off, size_t n, int fastq, unsigned skip) { Sink *s = u; if (own(s, b)) return 1;
  fprintf(LOG, "%d block %ld %zu %d %u\\n", s->id, ftell(DATA), n, fastq, skip); fwrite(b + off, 1, n, DATA); s->lent--; free(b - 8); return 0; }
static int sub(void *u, const char *p, size_t n) { (void)u; (void)p; (void)n; puts("submit in raw mode"); return 1; }
static int flu(void *u) { Sink *s = u; fprintf(LOG, "%d flush\\n", s->id); return 0; }
int main(int argc, char **argv) {
  int n = atoi(argv[2]); Sink S[8]; qm2_sink sinks[8]; char name[512];
  LOG = fopen(argv[4], "w"); snprintf(name, sizeof name, "%s.data", argv[4]); DATA = fopen(name, "wb");
  for (int i = 0; i < n; i++) { S[i].id = i; S[i].lent = 0;
    sinks[i].acquire = acq; sinks[i].commit = com; sinks[i].submit = sub; sinks[i].flush = flu; sinks[i].user = &S[i]; sinks[i].commit_raw = raw; sinks[i].commit_packed = NULL; }
  qm2_dispatch *d = qm2_dispatch_new(sinks, n, 1);
  qm2_dispatch_raw_turn(d, (size_t)atol(argv[3]));
  qm2_lines_out out = {qm2_dispatch_acquire, qm2_dispatch_commit, qm2_dispatch_submit, d, qm2_dispatch_commit_raw, NULL};
  int rc = qm2_stream_reads(argv[1], &out, atoi(argv[5]));
  qm2_dispatch_free(d);
  for (int i = 0; i < n; i++) if (S[i].lent) rc = 3;
  fclose(LOG); fclose(DATA); return rc; }
''')
    exe = str(tmp_path / "r")
    subprocess.run(["gcc", "-O1", "-pthread", "-I", src, "-o", exe, str(harness), os.path.join(src, "qm2_reader.c"),
                    os.path.join(src, "qm2_dispatch.c"), os.path.join(src, "qm2_pack.c")], check=True)
    import oracle_lib as ol
    import synth
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    for path in (rfq, rfa):
        exp = ol.strip_reads(path)
        whole = open(path, "rb").read()
        for n_sinks, per_turn, thr, block in ((2, 3, 2, 8192), (3, 1, 4, 4096), (4, 5, 3, 20000), (1, 2, 1, 8192)):
            log = str(tmp_path / "rlog")
            subprocess.run([exe, path, str(n_sinks), str(per_turn), log, str(thr)], check=True,
                           env=dict(os.environ, QM2_BLOCK_BYTES=str(block)))
            data = open(log + ".data", "rb").read()
            events = [l.split() for l in open(log).read().splitlines()]
            blocks = [e for e in events if e[1] == "block"]
            # the blocks, in commit order, are the file's text (minus the FASTQ header line the reader consumed)
            assert data == (whole[whole.index(b"\n") + 1:] if whole[:1] == b"@" else whole)
            # block j belongs to sink (j // per_turn) % n; a flush of that sink follows the last block of every full turn
            got, pending, j = [], {}, 0
            for e in events:
                sink = int(e[0])
                if e[1] == "block":
                    assert sink == (j // per_turn) % n_sinks
                    at, n, fastq, skip = int(e[2]), int(e[3]), int(e[4]), int(e[5])
                    if n and sink not in pending:
                        pending[sink] = [at, at, fastq, skip]           # a batch opens with the state of its first block
                    if n:
                        assert pending[sink][1] == at                    # ... and holds consecutive text
                        pending[sink][1] = at + n
                    j += 1
                else:
                    assert n_sinks > 1 and j % per_turn == 0 and sink == ((j - 1) // per_turn) % n_sinks
                    if sink in pending:
                        got.append(pending.pop(sink))
            got += list(pending.values())
            lines = []
            for lo, hi, fastq, skip in got:                               # what a device makes of each batch
                for L, line in enumerate(data[lo:hi].split(b"\n")[:-1]):
                    if fastq:
                        if L >= skip and (L - skip) % 4 == 0:
                            lines.append(line)
                    elif not line.startswith(b">"):
                        lines.append(line)
            assert sorted(lines) == sorted(exp.split(b"\n")[:-1])
            if n_sinks > 1:
                assert len({(k // per_turn) % n_sinks for k in range(j)}) == min(n_sinks, (j + per_turn - 1) // per_turn)


def test_reader_looks_where_the_files_pages_lie(pkg, tmp_path):
    """csrc/qm2_reader.c, file_pages_cpus: 64 pages of a regular file are mapped (never touched) and move_pages(2) says which
    NUMA node each lies on; a 3/4 majority sends the block reader's workers there.  Here: the path runs, agrees with itself
    (the file was just written by this process: one node) and the reader's result is what it is without it."""
    probe = os.path.join(ROOT, "tools", "micro", "reader_probe")
    if not os.path.exists(probe):
        pytest.skip("tools/micro/reader_probe not built")
    fq = tmp_path / "t.fq"
    fq.write_bytes(b"".join(b"@r%d\n" % i + b"ACGT" * 37 + b"AC\n+\n" + b"I" * 150 + b"\n" for i in range(30000)))
    outs = []
    for env in ({"QM2_READER_TIMING": "1"}, {"QM2_READER_TIMING": "1", "QM2_NO_BIND": "1"}):
        r = subprocess.run([probe, str(fq), "4", "packed", "1"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
        txt = r.stdout.decode()
        assert r.returncode == 0 and " rc=0 " in txt, txt
        outs.append(txt)
    m = re.search(r"(\d+) of 64 sampled pages of the file lie on NUMA node (\d+): the workers run on its (\d+) CPUs", outs[0])
    if m:                                        # (a sandbox may refuse move_pages: then nothing is said and nothing changes)
        assert int(m.group(1)) >= 48 and int(m.group(3)) >= 1
    assert "sampled pages" not in outs[1]        # QM2_NO_BIND leaves the affinities alone
    bases = [re.search(r"= ([0-9.]+) G bases/s", t) is not None for t in outs]
    assert all(bases)


def test_bench_gpus_means_ranks(tmp_path):
    """`python bench.py --gpus N` (the driver's own form): N > 1 without a rank environment starts N ranks of the script under
    torch.distributed.run as a child process — checked with a stand-in launcher — and relays rank 0's one JSON line and the
    exit code; a rank environment that disagrees with --gpus fails loudly instead of printing a line for another N."""
    import json
    import sys
    stub = tmp_path / "launcher.py"
    stub.write_text('''
import json, os, sys
a = sys.argv[1:]
sys.stderr.write("STUB " + " ".join(a) + "\\n")
print("some native library banner")
n = int(a[a.index("--nproc-per-node") + 1])
print(json.dumps({"n_gpus": n, "script": a[a.index("--master-port") + 2], "rest": a[a.index("--master-port") + 3:], "addr": os.environ.get("MASTER_ADDR")}))
sys.exit(int(os.environ.get("STUB_RC", "0")))
''')
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["QM2_BENCH_LAUNCHER"] = "%s %s" % (sys.executable, stub)
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                    # ONE line on stdout; the banner went to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and os.path.samefile(d["script"], bench) and d["rest"] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and d["addr"] == "127.0.0.1"
    assert b"--nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, STUB_RC="7"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 7
    # the launcher's rank count and --gpus disagree (e.g. torchrun --nproc-per-node 2 bench.py --gpus 8): no line, an error
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], env=dict(env, WORLD_SIZE="2", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == b"" and b"disagree" in r.stderr
    r = subprocess.run([sys.executable, bench], env=dict(env, WORLD_SIZE="2", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"disagree" in r.stderr              # (--gpus defaults to 1)
