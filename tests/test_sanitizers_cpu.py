"""The host side under the sanitizers (CPU only; the GPU pool has none): the threaded reader (qm2_reader.c: pread
blocks, newline-count hand-over, carries between blocks) and the dispatcher of `--devices` (qm2_dispatch.c) built with
ThreadSanitizer and with AddressSanitizer + UBSan and driven over FASTA / FASTQ inputs with small blocks and several
thread counts, both hand-overs (stripped lines, raw blocks in turns); and tools/qm2_mkdict.c under ASan + UBSan on the
dictionaries the reference wrote (incl. the growth at 80 % fill), `est` (qm2_est.c) on the reference's KAT counts, and the
threaded .bin / GC-curve loop of the epilogue (qm2_formats.c) under both, fed the reference's own counts through stand-ins
for the device calls it makes.  A report from a sanitizer fails the test; so does
output that differs from the oracle's strip / the reference's files."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "quick-mer2_amd", "csrc")
GOLD = os.path.join(HERE, "golden")

HARNESS = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qm2_dispatch.h"
#include "qm2_host.h"
/* argv: reads, n_sinks, strip threads, out prefix, "raw" | "host" | "packed", blocks per raw turn / bytes (positions) per host (packed) turn */
typedef struct { int id; long lent; FILE *f; } Sink;
static size_t CAP = 1 << 18;
static int acq(void *u, char **b, size_t *c) { Sink *s = u; *c = CAP; *b = malloc(*c + 8); if (!*b) return 1; memcpy(*b, &s->id, sizeof s->id); *b += 8; s->lent++; return 0; }
static int own(Sink *s, char *b) { int id; memcpy(&id, b - 8, sizeof id); return id != s->id; }
static int com(void *u, char *b, size_t n) { Sink *s = u; if (own(s, b)) return 1; fwrite(b, 1, n, s->f); s->lent--; free(b - 8); return 0; }
static int raw(void *u, char *b, size_t off, size_t n, int fastq, unsigned skip) { Sink *s = u; if (own(s, b)) return 1;
  unsigned hdr[3] = {(unsigned)n, (unsigned)fastq, skip}; if (n) { fwrite(hdr, 4, 3, s->f); fwrite(b + off, 1, n, s->f); } s->lent--; free(b - 8); return 0; }
static int sub(void *u, const char *p, size_t n) { Sink *s = u; fwrite(p, 1, n, s->f); return 0; }
static int flu(void *u) { Sink *s = u; const unsigned mark[3] = {0xFFFFFFFFu, 0, 0}; fwrite(mark, 4, 3, s->f); return 0; }
/* packed mode: every piece framed — {0xFFFFFFFE, blocks, line ends + filler} + blocks, or {0xFFFFFFFD, bytes, 0} + ASCII lines */
static int pk(void *u, char *b, const qm2_packed *p, const uint64_t *w) { Sink *s = u; (void)w; if (own(s, b)) return 1;
  unsigned hdr[3] = {0xFFFFFFFEu, (unsigned)p->n_blocks, (unsigned)p->n_newlines}; fwrite(hdr, 4, 3, s->f); fwrite(b, 24, p->n_blocks, s->f); s->lent--; free(b - 8); return 0; }
static int com_f(void *u, char *b, size_t n) { Sink *s = u; if (own(s, b)) return 1; unsigned hdr[3] = {0xFFFFFFFDu, (unsigned)n, 0};
  if (n) { fwrite(hdr, 4, 3, s->f); fwrite(b, 1, n, s->f); } s->lent--; free(b - 8); return 0; }
static int sub_f(void *u, const char *p, size_t n) { Sink *s = u; unsigned hdr[3] = {0xFFFFFFFDu, (unsigned)n, 0}; fwrite(hdr, 4, 3, s->f); fwrite(p, 1, n, s->f); return 0; }
int main(int argc, char **argv) {
  if (argc < 7) return 9;
  qm2_reader_note_affinity();
  const int n = atoi(argv[2]), is_raw = strcmp(argv[5], "raw") == 0, is_packed = strcmp(argv[5], "packed") == 0; Sink S[8]; qm2_sink sinks[8]; char name[600];
  for (int i = 0; i < n; i++) { snprintf(name, sizeof name, "%s.%d", argv[4], i); S[i].id = i; S[i].lent = 0; S[i].f = fopen(name, "wb");
    sinks[i].acquire = acq; sinks[i].commit = is_packed ? com_f : com; sinks[i].submit = is_packed ? sub_f : sub; sinks[i].flush = flu; sinks[i].user = &S[i];
    sinks[i].commit_raw = is_raw ? raw : NULL; sinks[i].commit_packed = is_packed ? pk : NULL; }
  qm2_dispatch *d = qm2_dispatch_new(sinks, n, is_raw ? 1 : (size_t)atol(argv[6]));
  if (is_raw) qm2_dispatch_raw_turn(d, (size_t)atol(argv[6]));
  qm2_lines_out out = {qm2_dispatch_acquire, qm2_dispatch_commit, qm2_dispatch_submit, d, is_raw ? qm2_dispatch_commit_raw : NULL, is_packed ? qm2_dispatch_commit_packed : NULL};
  int rc = qm2_stream_reads(argv[1], &out, atoi(argv[3]));
  qm2_dispatch_free(d);
  for (int i = 0; i < n; i++) { if (S[i].lent) rc = 3; fclose(S[i].f); }
  return rc; }
'''

SAN = {
    "tsan": (["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66 second_deadlock_stack=1"}),
    "asan_ubsan": (["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                   {"ASAN_OPTIONS": "detect_leaks=1 exitcode=67", "UBSAN_OPTIONS": "print_stacktrace=1"}),
}


def sanitizer_works(flags, tmp_path):
    p = tmp_path / "probe.c"
    p.write_text("int main(void) { return 0; }\n")
    exe = str(tmp_path / "probe")
    if subprocess.run(["gcc"] + flags + ["-o", exe, str(p)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode:
        return False
    return subprocess.run([exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode == 0


def device_rule(blob):
    """What a device makes of the raw blocks a sink received (batches end at the flush marks): the sequence lines."""
    kept, at, batch = [], 0, []

    def close(batch):
        if not batch:
            return
        fastq, skip = batch[0][1], batch[0][2]                  # a batch opens with the state of its first block
        text = b"".join(t for t, _, _ in batch)
        lines = text.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()
        for i, l in enumerate(lines):
            if fastq:
                if i >= skip and (i - skip) % 4 == 0 and not l.startswith(b">"):
                    kept.append(l)
            elif not l.startswith(b">"):
                kept.append(l)

    while at < len(blob):
        n, fastq, skip = (int(x) for x in np.frombuffer(blob, dtype=np.uint32, count=3, offset=at))
        at += 12
        if n == 0xFFFFFFFF:
            close(batch)
            batch = []
            continue
        batch.append((blob[at:at + n], fastq, skip))
        at += n
    close(batch)
    return kept


@pytest.mark.parametrize("san", sorted(SAN))
def test_reader_and_dispatcher_under_sanitizers(san, tmp_path):
    import oracle_lib as ol
    flags, env = SAN[san]
    if not sanitizer_works(flags, tmp_path):
        pytest.skip("this gcc has no %s run-time" % san)
    harness = tmp_path / "h.c"
    harness.write_text(HARNESS)
    exe = str(tmp_path / "h")
    subprocess.run(["gcc", "-O1", "-g", "-pthread"] + flags + ["-I", SRC, "-o", exe, str(harness), os.path.join(SRC, "qm2_reader.c"),
                    os.path.join(SRC, "qm2_dispatch.c"), os.path.join(SRC, "qm2_pack.c")], check=True)
    fa, ctrl, rfa, rfq = synth.mini_inputs(str(tmp_path))
    # ragged FASTQ: line lengths from 0 to 400 (empty sequence lines too), N's, no final newline
    rng = np.random.default_rng(5)
    recs = []
    for i in range(20000):
        n = int(rng.integers(0, 400))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n, p=[.24, .24, .24, .24, .04]))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * n + b"\n")
    ragged = tmp_path / "ragged.fq"
    ragged.write_bytes(b"".join(recs)[:-1])
    for path in (rfq, rfa, str(ragged)):
        exp = ol.strip_reads(path)
        exp_lines = sorted(exp.split(b"\n"))
        for n_sinks, thr, mode, turn, block in ((2, 4, "raw", 3, 8192), (3, 8, "raw", 1, 4096), (1, 3, "raw", 4, 65536),
                                                (2, 4, "host", 20000, 16384), (3, 0, "host", 5000, 8192), (1, 6, "host", 1 << 30, 4096),
                                                (1, 5, "packed", 1 << 30, 8192), (3, 4, "packed", 30000, 4096), (2, 8, "packed", 5000, 16384)):
            out = str(tmp_path / "o")
            r = subprocess.run([exe, path, str(n_sinks), str(thr), out, mode, str(turn)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, QM2_BLOCK_BYTES=str(block), **env), timeout=600)
            assert r.returncode == 0, (san, path, mode, r.stderr.decode()[-3000:])
            assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode()[-3000:]
            parts = [open("%s.%d" % (out, i), "rb").read() for i in range(n_sinks)]
            if mode == "packed":
                # pieces of 2-bit codes + reset flags (qm2_pack.c on the worker threads): every sink's positions together
                # hold the stripped stream's bases, the same codes between the same resets
                import test_pack_cpu as tp
                bases, runs = 0, []

                def runs_of(c, r):
                    good = np.flatnonzero(~r)
                    if len(good) == 0:
                        return []
                    return [x.tobytes() for x in np.split(c[good].astype(np.uint8), np.flatnonzero(np.diff(good) > 1) + 1)]

                for blob in parts:
                    at = 0
                    while at < len(blob):
                        tag, a, b = (int(x) for x in np.frombuffer(blob, dtype=np.uint32, count=3, offset=at))
                        at += 12
                        if tag == 0xFFFFFFFF:
                            continue
                        if tag == 0xFFFFFFFE:
                            c, r = tp.unpack(np.frombuffer(blob, dtype=np.uint64, count=3 * a, offset=at))
                            at += 24 * a
                            bases += 64 * a - b
                        else:
                            assert tag == 0xFFFFFFFD
                            c, r = tp.expect(blob[at:at + a])
                            bases += a - blob[at:at + a].count(b"\n")
                            at += a
                        runs += runs_of(c, r)
                assert bases == len(exp) - exp.count(b"\n"), (san, path, n_sinks, thr)
                # the same runs of bases between resets, whichever sink got them in whatever order
                assert sorted(runs) == sorted(runs_of(*tp.expect(exp))), (san, path, n_sinks, thr)
            elif mode == "raw" and thr > 0:
                got = [l for p in parts for l in device_rule(p)]
                assert sorted(got + [b""]) == exp_lines, (san, path, n_sinks, thr)
            else:
                # stripped lines, flush marks in between (12 bytes, never a sequence line)
                text = b"".join(parts).replace(b"\xff\xff\xff\xff" + b"\0" * 8, b"")
                assert sorted(text.split(b"\n")) == exp_lines, (san, path, n_sinks, thr)


def test_mkdict_under_address_and_ub_sanitizers(tmp_path):
    flags, env = SAN["asan_ubsan"]
    if not sanitizer_works(flags, tmp_path):
        pytest.skip("this gcc has no asan/ubsan run-time")
    exe = str(tmp_path / "mkdict")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11"] + flags + ["-o", exe, os.path.join(ROOT, "tools", "qm2_mkdict.c"), "-lm"], check=True)
    growth = json.load(open(os.path.join(GOLD, "growth_manifest.json")))
    runs = [("kat", 30, "16K"), ("kat_k12", 12, "16K"), ("kat_k31", 31, "16K")]
    for setname, k, size in runs:
        d = os.path.join(GOLD, setname)
        w = tmp_path / setname
        w.mkdir()
        shutil.copy(os.path.join(d, "ref.fa"), w / "ref.fa")
        shutil.copy(os.path.join(d, "ctrl.bed"), w / "ctrl.bed")
        r = subprocess.run([exe, "-k", str(k), "-s", size, "-e", "0", "-w", "1000", "-c", str(w / "ctrl.bed"), str(w / "ref.fa")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode()[-3000:]
        for ext in (".qm", ".qgc", ".bed"):
            assert open(str(w / "ref.fa") + ext, "rb").read() == open(os.path.join(d, "ref.fa" + ext), "rb").read(), (setname, ext)
    # the table grows at 80 % fill (QuicKmer.c:891-895): the cases of tests/golden/growth_manifest.json (-s 16K doubles three times)
    import hashlib
    for size, g in sorted(growth["cases"].items()):
        w = tmp_path / ("grow_" + size)
        w.mkdir()
        fa, ctrl, rfa, rfq = synth.mini_inputs(str(w))
        r = subprocess.run([exe, "-k", "30", "-s", size, "-e", "0", "-w", "1000", "-c", ctrl, fa],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=900)
        assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode()[-3000:]
        assert r.stdout.decode().count("Resize ") == g["resizes"]
        assert hashlib.md5(open(fa + ".qm", "rb").read()).hexdigest() == g["qm_md5"]


def test_est_under_address_and_ub_sanitizers(tmp_path):
    flags, env = SAN["asan_ubsan"]
    if not sanitizer_works(flags, tmp_path):
        pytest.skip("this gcc has no asan/ubsan run-time")
    main = tmp_path / "m.c"
    main.write_text('#include "qm2_host.h"\nint main(int argc, char **argv) { return qm2_main_est(argc, argv); }\n')
    exe = str(tmp_path / "est")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11"] + flags + ["-I", SRC, "-o", exe, str(main), os.path.join(SRC, "qm2_est.c"), "-lm"], check=True)
    helper = os.path.join(ROOT, "tests", "est_helper")
    for sample in ("reads_fa", "polyA", "wrap70000", "short_reads"):
        w = tmp_path / sample
        w.mkdir()
        for f in ("ref.fa.qgc", "ref.fa.bed", sample + ".bin", sample + ".txt"):
            shutil.copy(os.path.join(GOLD, "kat", f), w)
        r = subprocess.run([exe, "ref.fa", sample, "out.bed"], cwd=str(w), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, PATH=os.pathsep.join([helper, os.environ["PATH"]]), **env))
        assert r.returncode == 0 and b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode()[-3000:]
        assert open(w / "out.bed", "rb").read() == open(os.path.join(GOLD, "est", sample + ".bed"), "rb").read()


EPILOGUE = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "qm2.h"
#include "qm2_host.h"
/* stand-ins for the device: the counts in chain order ARE the reference's .bin */
static uint16_t *BIN; static uint64_t N;
int qm2_chain_gather_fd(qm2_ctx *c, int fd, uint64_t off, uint64_t first, uint64_t *n_out) { (void)c; (void)fd; (void)off; (void)first; *n_out = N; return 0; }
int qm2_chain_fetch(qm2_ctx *c, uint16_t *out, uint64_t offset, uint64_t count) { (void)c; if (offset + count > N) return 1; memcpy(out, BIN + offset, count * 2); return 0; }
int qm2_keys_from_fd(qm2_ctx *c, int fd, uint64_t off) { (void)c; (void)fd; (void)off; return 1; }
/* (the device's sums, QuicKmer.c:504-509, serially: what the kernel is compared with on the GPU box is the reference's .txt) */
int qm2_chain_gc_sums_fd(qm2_ctx *c, int fd, uint64_t n_file, const uint16_t *stale, int64_t *sum, int64_t *sumsq, uint64_t *cnt) { (void)c;
  memset(sum, 0, 401 * 8); memset(sumsq, 0, 401 * 8); memset(cnt, 0, 401 * 8);
  FILE *f = fdopen(dup(fd), "rb"); if (!f) return 1; rewind(f);
  for (uint64_t e = 0; e < N; e++) { uint16_t g;
    if (e < n_file) { if (fread(&g, 2, 1, f) != 1) { fclose(f); return 1; } } else { if (!stale) { fclose(f); return 1; } g = stale[e & 0xFFFFF]; }
    if ((g & 0x8000) && (g & 0x1FF) < 401) { const uint32_t v = BIN[e]; sum[g & 0x1FF] += v; cnt[g & 0x1FF]++; sumsq[g & 0x1FF] += (int32_t)(v * v); } }
  fclose(f); return 0; }
int main(int argc, char **argv) {      /* ref prefix, reference .bin, out prefix */
  if (argc < 4) return 9;
  FILE *f = fopen(argv[2], "rb"); if (!f) return 8;
  fseek(f, 0, SEEK_END); N = (uint64_t)ftell(f) / 2; rewind(f);
  BIN = malloc(N * 2 + 2); if (fread(BIN, 2, N, f) != N) return 7; fclose(f);
  qm2_dict_file d; if (qm2_dict_open(argv[1], &d)) return 6;
  int rc = qm2_write_outputs(&d, NULL, argv[1], argv[3]);
  qm2_dict_close(&d); free(BIN); return rc; }
'''


@pytest.mark.parametrize("san", sorted(SAN))
def test_epilogue_threads_under_sanitizers(san, tmp_path):
    flags, env = SAN[san]
    if not sanitizer_works(flags, tmp_path):
        pytest.skip("this gcc has no %s run-time" % san)
    (tmp_path / "e.c").write_text(EPILOGUE)
    exes = {}
    for piece in (64, 1 << 22):                                   # many pieces on many threads / the product's one piece
        exes[piece] = str(tmp_path / ("e%d" % piece))
        subprocess.run(["gcc", "-O1", "-g", "-pthread", "-DQM2_PIECE_ENTRIES=%du" % piece] + flags + ["-I", SRC, "-I", os.path.join(ROOT, "include"),
                        "-o", exes[piece], str(tmp_path / "e.c"), os.path.join(SRC, "qm2_formats.c")], check=True)
    kat = os.path.join(GOLD, "kat")
    for f in ("ref.fa.qm", "ref.fa.qgc"):
        shutil.copy(os.path.join(kat, f), tmp_path)

    def run(piece, sample, out):
        r = subprocess.run([exes[piece], str(tmp_path / "ref.fa"), os.path.join(kat, sample + ".bin"), out], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=300)
        assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-3000:])
        assert b"Sanitizer" not in r.stderr and b"runtime error" not in r.stderr, r.stderr.decode()[-3000:]

    for sample in ("reads_fa", "deep50000", "wrap70000", "polyA"):
        out = str(tmp_path / ("o_" + sample))
        run(64, sample, out)
        assert open(out + ".bin", "rb").read() == open(os.path.join(kat, sample + ".bin"), "rb").read()
        assert open(out + ".txt", "rb").read() == open(os.path.join(kat, sample + ".txt"), "rb").read()
    # a .qgc cut short (the stale-block rule, pinned against the reference itself on the GPU box): the pieces must not matter
    full = open(os.path.join(kat, "ref.fa.qgc"), "rb").read()
    for keep in (len(full) // 2 & ~1, 2 * 100, 0):
        (tmp_path / "ref.fa.qgc").write_bytes(full[:keep])
        run(64, "reads_fa", str(tmp_path / "s64"))
        run(1 << 22, "reads_fa", str(tmp_path / "s1"))
        assert open(tmp_path / "s64.txt", "rb").read() == open(tmp_path / "s1.txt", "rb").read()
        assert open(tmp_path / "s64.bin", "rb").read() == open(os.path.join(kat, "reads_fa.bin"), "rb").read()
