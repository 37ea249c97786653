/*
 * qm2_reader.c — block reader that reduces a FASTA/FASTQ stream to the
 * sequence lines main_count feeds its encoder (QuicKmer.c:393-398, :451-455).
 *
 * The reference does per-base work on its single producer thread, which is
 * what caps it at ~6 threads (README.md:95).  Here the host only finds line
 * boundaries (memchr) and copies sequence lines — straight into the pinned
 * staging buffer the backend hands out (qm2_staging_acquire), so the bytes are
 * touched once between the page cache and the DMA engine — while a second
 * thread keeps read(2) going; every per-base operation happens on the GPU.
 *
 * Line rules reproduced:
 *   - first line starts with '@'  => FASTQ; that line is consumed        (:393-395)
 *   - otherwise FASTA and the stream is rewound; on a pipe the rewind
 *     fails and the first line is simply lost                           (:396)
 *   - a would-be sequence line starting with '>' is skipped             (:398)
 *   - FASTQ: after each sequence line three lines are discarded         (:451-455)
 * Divergence (documented): the reference reads with fgets(100000) and runs
 * past the buffer on lines >= 99 999 chars or a last line without '\n'
 * (undefined behaviour).  Here lines are unbounded and a missing final
 * newline is supplied.
 */
#define _GNU_SOURCE
#include <errno.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <fcntl.h>
#include <pthread.h>
#include <semaphore.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include "qm2_host.h"

#define QM2_BLOCK   ((size_t)32 << 20)     /* bytes per read(2) block                          */
#define QM2_RESERVE ((size_t)1 << 20)      /* room in front of a block for the unfinished line
                                              of the block before it                          */
#define QM2_MAX_WORKERS 32
#define QM2_RING_MAX (QM2_MAX_WORKERS + 3)
#define QM2_WRAPS_CAP 512                   /* run-counter wraps a packed piece may carry (lines of >= 65536 bases only) */

/* ---- the I/O thread -------------------------------------------------------- */

typedef struct {
    int     fd;
    int     n_slots;                       /* workers + 3: a block stays put until its job is collected */
    char   *blk[QM2_RING_MAX];             /* each QM2_RESERVE + QM2_BLOCK bytes               */
    ssize_t len[QM2_RING_MAX];             /* bytes read into blk[i] + QM2_RESERVE; 0 = EOF, <0 = error */
    sem_t   free_slots, filled;
} io_ring;

static void *io_main(void *arg)
{
    io_ring *r = (io_ring *)arg;
    for (int i = 0;; i = (i + 1) % r->n_slots) {
        sem_wait(&r->free_slots);
        size_t have = 0;
        ssize_t got = 0;
        while (have < QM2_BLOCK) {         /* fill the block: pipes return short reads */
            got = read(r->fd, r->blk[i] + QM2_RESERVE + have, QM2_BLOCK - have);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) break;
            have += (size_t)got;
        }
        r->len[i] = got < 0 ? -1 : (ssize_t)have;
        sem_post(&r->filled);
        if (got < 0 || have == 0) return NULL;          /* error or EOF */
        if (have < QM2_BLOCK) {                         /* EOF after a partial block */
            i = (i + 1) % r->n_slots;
            sem_wait(&r->free_slots);
            r->len[i] = 0;
            sem_post(&r->filled);
            return NULL;
        }
    }
}

/* ---- output: one acquired buffer at a time --------------------------------- */

typedef struct {
    const qm2_lines_out *out;
    char  *buf;
    size_t cap, len;
} out_state;

static int out_flush(out_state *o)
{
    if (!o->buf) return 0;
    int rc = o->out->commit(o->out->user, o->buf, o->len);
    o->buf = NULL;
    o->len = 0;
    return rc;
}

static int out_line(out_state *o, const char *line, size_t body)
{
    if (o->buf && o->len + body + 1 > o->cap && out_flush(o)) return 1;
    if (!o->buf) {
        if (o->out->acquire(o->out->user, &o->buf, &o->cap)) return 1;
        o->len = 0;
    }
    if (body + 1 > o->cap) {
        /* one line longer than a staging buffer: hand it over through the copying path */
        char *tmp = malloc(body + 1);
        if (!tmp) { puts("Memory allocation failed"); return 1; }
        memcpy(tmp, line, body);
        tmp[body] = '\n';
        int rc = o->out->submit(o->out->user, tmp, body + 1);
        free(tmp);
        return rc;
    }
    memcpy(o->buf + o->len, line, body);
    o->len += body;
    o->buf[o->len++] = '\n';
    return 0;
}

/* ---- the line discipline ---------------------------------------------------- */

typedef struct {
    int first_line, fastq, skip, seekable;
    int desync;     /* FASTQ: a would-be sequence line started with '>' and was skipped WITHOUT the
                       three discards (:398 comes before :451): from here on the state is no longer
                       a function of the line count, which the parallel path relies on */
    unsigned lines; /* lines seen (pack_whole_lines*, strip_range): whatever the state on entry */
} line_state;

static int take_line(line_state *s, out_state *o, const char *line, size_t body)
{
    if (s->first_line) {
        s->first_line = 0;
        if (body && line[0] == '@') { s->fastq = 1; return 0; }   /* :395 */
        if (!s->seekable) return 0;                               /* :396 on a pipe */
    }
    if (s->skip) { s->skip--; return 0; }                         /* :452-454 */
    if (body && line[0] == '>') { if (s->fastq) s->desync = 1; return 0; }   /* :398 */
    if (out_line(o, line, body)) return 1;
    if (s->fastq) s->skip = 3;
    return 0;
}

/* ---- parallel strip -------------------------------------------------------- *
 * With n_threads > 0 the line discipline runs on worker threads, one 32 MiB block per job.
 * Counting is order-independent, but the line discipline is not: a job needs the state at
 * its first line.  In FASTQ mode that state advances by exactly one step per line
 * (skip -> skip-1 -> ... -> 0 -> emit -> 3), so the coordinator only counts the newlines of a
 * block to know where the next one starts; the one exception (a sequence-position line that
 * starts with '>', see line_state.desync) is detected by the worker and makes the
 * coordinator redo the later blocks serially with the exact state.  Outputs are committed in
 * block order. */
typedef struct {
    const char *data;                      /* whole lines */
    size_t      n;
    line_state  st;                        /* in: state at the first line; out: state after the last */
    char       *obuf;                      /* pinned output from out->acquire */
    size_t      ocap, olen;
    int         rc;
    int         pack_mode;                 /* pack what the line discipline keeps (qm2_lines_out.commit_packed) */
    int         is_packed;                 /* obuf holds `packed` (else olen bytes of ASCII lines) */
    qm2_packed  packed;
    uint64_t    wraps[QM2_WRAPS_CAP];
    sem_t       start, done;
    int         quit;
} strip_job;

static int buf_acquire(void *u, char **b, size_t *c) { (void)u; (void)b; (void)c; return 1; }
static int buf_commit(void *u, char *b, size_t n) { (void)u; (void)b; (void)n; return 0; }
static int buf_submit(void *u, const char *p, size_t n) { (void)u; (void)p; (void)n; return 1; }

/* the line discipline and the packing in one pass over whole lines (qm2_pack.c); 0 = packed, 1 = error, 2 = take the ASCII way */
static int pack_whole_lines(const char *p, size_t n, line_state *st, char *obuf, size_t ocap, qm2_packed *packed, uint64_t *wraps)
{
    qm2_line_state ps = {(uint32_t)st->fastq, (uint32_t)st->skip, 0, 0};
    const int rc = qm2_pack_reads(p, n, &ps, obuf, ocap, wraps, QM2_WRAPS_CAP, packed);
    if (rc) return rc;
    st->skip = (int)ps.skip;
    st->lines += ps.lines;
    if (ps.desync) st->desync = 1;
    return 0;
}

static int pack_whole_lines_cap(const char *p, size_t n, line_state *st, char *obuf, size_t ocap, qm2_packed *packed, uint64_t *wraps, size_t wraps_cap)
{
    qm2_line_state ps = {(uint32_t)st->fastq, (uint32_t)st->skip, 0, 0};
    const int rc = qm2_pack_reads(p, n, &ps, obuf, ocap, wraps, wraps_cap, packed);
    if (rc) return rc;
    st->skip = (int)ps.skip;
    st->lines += ps.lines;
    if (ps.desync) st->desync = 1;
    return 0;
}

static void strip_block(strip_job *j)
{
    j->is_packed = 0;
    if (j->pack_mode) {
        line_state st = j->st;
        const int prc = pack_whole_lines(j->data, j->n, &st, j->obuf, j->ocap, &j->packed, j->wraps);
        if (prc == 0) { j->st = st; j->is_packed = 1; j->rc = 0; j->olen = 0; return; }
        if (prc == 1) { j->rc = 1; j->olen = 0; return; }
    }
    /* an out_state pinned to the job's buffer: the output never exceeds the input, so it fits */
    static const qm2_lines_out none = {buf_acquire, buf_commit, buf_submit, NULL, NULL, NULL};
    out_state os = {&none, j->obuf, j->ocap, 0};
    size_t at = 0;
    j->rc = 0;
    while (at < j->n) {
        const char *nl = memchr(j->data + at, '\n', j->n - at);
        const size_t body = (size_t)(nl - (j->data + at));
        if (take_line(&j->st, &os, j->data + at, body)) { j->rc = 1; break; }
        at += body + 1;
    }
    j->olen = os.len;
}

static void *strip_main(void *arg)
{
    strip_job *j = (strip_job *)arg;
    for (;;) {
        sem_wait(&j->start);
        if (j->quit) return NULL;
        strip_block(j);
        sem_post(&j->done);
    }
}

/* Block size of the regular-file path: 32 MiB; QM2_BLOCK_BYTES (4 KiB .. 32 MiB) makes small test inputs span
 * several blocks (and several turns of `--devices`). */
size_t qm2_reader_block_bytes(void)
{
    const char *env = getenv("QM2_BLOCK_BYTES");
    if (env) {
        const unsigned long long v = strtoull(env, NULL, 10);
        if (v >= 4096 && v <= QM2_BLOCK) return (size_t)v;
    }
    return QM2_BLOCK;
}

/* Newlines in p[0, n): the one pass the host makes over raw FASTQ text (the line number at a block's first byte decides
 * which lines are sequence lines, :451-455).  SSE2 (part of every x86-64): 16 bytes per compare, the matches summed in
 * byte lanes for up to 255 rounds, then across lanes — the loop runs at the speed the memory delivers; the word-wise
 * zero-byte count this replaces ran at 3 - 4 GB/s and was, with 16 threads, what bounded the CLI's count phase. */
static size_t count_newlines(const char *p, size_t n)
{
    size_t c = 0, i = 0;
#if defined(__SSE2__)
    const __m128i nl = _mm_set1_epi8('\n'), zero = _mm_setzero_si128();
    while (n - i >= 64) {
        size_t rounds = (n - i) / 64;
        if (rounds > 255) rounds = 255;
        __m128i a0 = zero, a1 = zero, a2 = zero, a3 = zero;
        for (size_t r = 0; r < rounds; r++, i += 64) {
            a0 = _mm_sub_epi8(a0, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
            a1 = _mm_sub_epi8(a1, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 16)), nl));
            a2 = _mm_sub_epi8(a2, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 32)), nl));
            a3 = _mm_sub_epi8(a3, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 48)), nl));
        }
        const __m128i s = _mm_add_epi64(_mm_add_epi64(_mm_sad_epu8(a0, zero), _mm_sad_epu8(a1, zero)),
                                        _mm_add_epi64(_mm_sad_epu8(a2, zero), _mm_sad_epu8(a3, zero)));
        c += (size_t)_mm_cvtsi128_si64(s) + (size_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(s, s));
    }
#else
    const uint64_t nlw = 0x0A0A0A0A0A0A0A0Aull, lo7 = 0x7F7F7F7F7F7F7F7Full;
    for (; i + 8 <= n; i += 8) {                       /* 8 bytes per word, exact zero-byte test */
        uint64_t w;
        memcpy(&w, p + i, 8);
        const uint64_t t = w ^ nlw;
        c += (size_t)__builtin_popcountll(~(((t & lo7) + lo7) | t | lo7));
    }
#endif
    for (; i < n; i++) c += p[i] == '\n';
    return c;
}

/* ---- regular files: pread + fully parallel blocks ---------------------------- *
 * A regular file is cut into 32 MiB blocks; block b OWNS the lines that start inside it.
 * Worker threads take blocks round-robin and do everything for their block themselves:
 * pread it (plus the byte before, plus the rest of its last line) into a private, warm
 * buffer, count the owned lines, wait for the line count of the blocks before it (a chain of
 * cheap hand-overs), derive the FASTQ state from that count, and strip into a pinned staging
 * buffer.  The calling thread only lends buffers and commits them in block order.  (Mapping
 * the file instead was measured slower: populating page tables for page-cache pages costs
 * more than the copy pread makes into a buffer that stays hot.)  As in the streaming path, a
 * FASTQ sequence-position line that starts with '>' (line_state.desync) makes the rest of the
 * file go through the exact serial state machine. */
typedef struct pf_shared {
    int     fd;
    size_t  block;                         /* bytes per block (qm2_reader_block_bytes) */
    size_t  size, start;                   /* start: first byte after the consumed FASTQ header (or 0) */
    size_t  n_blocks;
    int     fastq;
    size_t     *lines_before;              /* [n_blocks + 1], filled in block order; published through known[] */
    atomic_int *known;                     /* lines_before[b] is valid (release / acquire) */
    atomic_int  stop;
} pf_shared;

typedef struct {
    pf_shared  *sh;
    char       *raw;                       /* private read buffer */
    size_t      raw_cap;
    /* per job */
    size_t      block;
    char       *obuf;
    size_t      ocap, olen;
    size_t      lo, hi;                    /* owned bytes: raw[lo, hi) */
    size_t      file_hi;                   /* file offset of raw[hi] */
    size_t      before;                    /* lines before this block */
    line_state  st_end;
    int         too_big, rc;
    int         raw_mode;                  /* read straight into obuf, do not strip */
    double      t_read, t_count, t_wait, t_pack;   /* QM2_READER_TIMING */
    int         pack_mode;                 /* strip and pack (qm2_lines_out.commit_packed) */
    int         is_packed;                 /* obuf holds `packed` */
    qm2_packed  packed;
    uint64_t    wraps[QM2_WRAPS_CAP];
    sem_t       start, done;
    int         quit;
} pf_worker;

static ssize_t pread_full(int fd, char *buf, size_t n, size_t off)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, buf + got, n - got, (off_t)(off + got));
        if (r < 0) { if (errno == EINTR) continue; return -1; }
        if (r == 0) break;
        got += (size_t)r;
    }
    return (ssize_t)got;
}

/* read block b and settle raw[lo, hi) = the lines that start inside it */
static int pf_load(pf_worker *w)
{
    pf_shared *sh = w->sh;
    const size_t s = sh->start + w->block * sh->block;
    size_t e = s + sh->block;
    if (e > sh->size) e = sh->size;
    const size_t base = w->block ? s - 1 : s;              /* the byte before tells whether a line starts at s */
    if (w->raw_mode) {                                      /* the pinned buffer itself; it cannot grow */
        w->raw = w->obuf;
        w->raw_cap = w->ocap;
        if (e - base > w->raw_cap) { w->too_big = 1; w->lo = w->hi = 0; w->file_hi = e; return 0; }
    }
    ssize_t r = pread_full(sh->fd, w->raw, e - base, base);
    if (r < 0) return 1;
    size_t n = (size_t)r;
    if (base + n < e) e = base + n;                        /* the file shrank under us: treat as EOF */
    size_t lo = 0;
    if (w->block) {
        if (n && w->raw[0] == '\n') lo = 1;
        else {
            const char *nl = n > 1 ? memchr(w->raw + 1, '\n', n - 1) : NULL;
            if (!nl) { w->lo = w->hi = 0; w->file_hi = e; return 0; }      /* no line starts in this block */
            lo = (size_t)(nl - w->raw) + 1;
        }
    }
    /* the rest of the last line lives in the following block(s) */
    while (n && w->raw[n - 1] != '\n' && base + n < sh->size) {
        const size_t step = 1u << 16;
        if (n + step > w->raw_cap) {
            if (w->raw_mode) { w->too_big = 1; break; }
            char *g = realloc(w->raw, w->raw_cap * 2);
            if (!g) return 1;
            w->raw = g; w->raw_cap *= 2;
        }
        r = pread_full(sh->fd, w->raw + n, step, base + n);
        if (r < 0) return 1;
        if (r == 0) break;
        const char *nl = memchr(w->raw + n, '\n', (size_t)r);
        if (nl) { n = (size_t)(nl - w->raw) + 1; break; }
        n += (size_t)r;
    }
    w->lo = lo; w->hi = n; w->file_hi = base + n;
    return 0;
}

static int strip_range(const char *p, size_t n, line_state *st, out_state *os)
{
    size_t at = 0;
    while (at < n) {
        const char *nl = memchr(p + at, '\n', n - at);
        const size_t body = nl ? (size_t)(nl - (p + at)) : n - at;
        if (take_line(st, os, p + at, body)) return 1;
        st->lines++;
        at += body + 1;
    }
    return 0;
}

static line_state pf_state(const pf_shared *sh, size_t before)
{
    line_state st = {0, sh->fastq, sh->fastq ? (int)((4 - (before & 3)) & 3) : 0, 1, 0, 0};
    return st;
}

/* QM2_READER_TIMING=1: where the block workers' time goes (seconds summed over the workers, printed at the end of the file) */
static double pf_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void *pf_main(void *arg)
{
    pf_worker *w = (pf_worker *)arg;
    pf_shared *sh = w->sh;
    for (;;) {
        sem_wait(&w->start);
        if (w->quit) return NULL;
        const size_t b = w->block;
        w->olen = 0; w->too_big = 0;
        const double t0 = pf_now();
        w->rc = atomic_load(&sh->stop) ? 0 : pf_load(w);
        const double t1 = pf_now();
        if (w->too_big) w->lo = w->hi = 0;
        if (w->rc || atomic_load(&sh->stop)) w->lo = w->hi = 0;
        /* (FASTA has no line numbering to keep: a line is dropped for its first byte alone, :398) */
        size_t lines = sh->fastq ? count_newlines(w->raw + w->lo, w->hi - w->lo) : 0;
        if (sh->fastq && w->hi > w->lo && w->raw[w->hi - 1] != '\n') lines++;       /* last line of the file, no newline */
        const double t2 = pf_now();
        while (!atomic_load_explicit(&sh->known[b], memory_order_acquire) && !atomic_load(&sh->stop)) sched_yield();
        const double t3 = pf_now();
        w->before = sh->lines_before[b];
        sh->lines_before[b + 1] = w->before + lines;
        atomic_store_explicit(&sh->known[b + 1], 1, memory_order_release);
        line_state st = pf_state(sh, w->before);
        w->is_packed = 0;
        if (!atomic_load(&sh->stop) && !w->rc && w->pack_mode && w->hi > w->lo) {
            const int prc = pack_whole_lines(w->raw + w->lo, w->hi - w->lo, &st, w->obuf, w->ocap, &w->packed, w->wraps);
            if (prc == 0) w->is_packed = 1;
            else if (prc == 1) w->rc = 1;
            else st = pf_state(sh, w->before);                         /* (does not pack: as ASCII lines, below) */
        }
        if (!atomic_load(&sh->stop) && !w->rc && !w->raw_mode && !w->is_packed) {
            if (w->hi - w->lo + 1 > w->ocap) w->too_big = 1;          /* more than a staging buffer holds */
            else {
                static const qm2_lines_out none = {buf_acquire, buf_commit, buf_submit, NULL, NULL, NULL};
                out_state os = {&none, w->obuf, w->ocap, 0};
                if (strip_range(w->raw + w->lo, w->hi - w->lo, &st, &os)) w->rc = 1;
                w->olen = os.len;
            }
        }
        w->st_end = st;
        w->t_read += t1 - t0; w->t_count += t2 - t1; w->t_wait += t3 - t2; w->t_pack += pf_now() - t3;
        sem_post(&w->done);
    }
}

/* exact serial state machine over file bytes [off, size) (after a desync) */
static int pf_serial_tail(const pf_shared *sh, size_t off, line_state *st, out_state *os)
{
    size_t cap = (size_t)8 << 20, have = 0;
    char *buf = malloc(cap);
    if (!buf) return 1;
    int rc = 0;
    for (;;) {
        if (cap - have < ((size_t)4 << 20)) {
            char *g = realloc(buf, cap * 2);
            if (!g) { rc = 1; break; }
            buf = g; cap *= 2;
        }
        const ssize_t r = pread_full(sh->fd, buf + have, (size_t)4 << 20, off);
        if (r < 0) { rc = 1; break; }
        off += (size_t)r;
        have += (size_t)r;
        const int eof = r == 0;
        size_t at = 0;
        while (at < have) {
            const char *nl = memchr(buf + at, '\n', have - at);
            if (!nl && !eof) break;
            const size_t body = nl ? (size_t)(nl - (buf + at)) : have - at;
            if (take_line(st, os, buf + at, body)) { rc = 1; break; }
            at += body + (nl ? 1 : 0);
        }
        if (rc || eof) break;
        memmove(buf, buf + at, have - at);
        have -= at;
    }
    free(buf);
    return rc;
}

static int stream_blocks(int fd, size_t size, const qm2_lines_out *out, int n_threads)
{
    const int raw_mode = out->commit_raw != NULL;
    pf_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.fd = fd; sh.size = size;
    /* the first line decides the format (:393-395); the file is seekable, so a FASTA first line is kept */
    {
        char head[1 << 12];
        size_t off = 0;
        ssize_t r = pread_full(fd, head, sizeof head, 0);
        if (r < 0) return 1;
        if (r > 0 && head[0] == '@') {
            sh.fastq = 1;
            for (;;) {                                               /* skip the header line, however long */
                const char *nl = memchr(head, '\n', (size_t)r);
                if (nl) { sh.start = off + (size_t)(nl - head) + 1; break; }
                off += (size_t)r;
                r = pread_full(fd, head, sizeof head, off);
                if (r <= 0) { sh.start = size; break; }
            }
        }
    }
    int rc = 0;
    out_state os = {out, NULL, 0, 0};
    const size_t body = size - sh.start;
    sh.block = qm2_reader_block_bytes();
    sh.n_blocks = (body + sh.block - 1) / sh.block;
    sh.lines_before = calloc(sh.n_blocks + 2, sizeof(size_t));
    sh.known = calloc(sh.n_blocks + 2, sizeof(atomic_int));
    pf_worker *ws = calloc((size_t)n_threads, sizeof *ws);
    pthread_t *tids = calloc((size_t)n_threads, sizeof *tids);
    if (!sh.lines_before || !sh.known || !ws || !tids) { puts("Memory allocation failed"); rc = 1; }
    int n_workers = 0;
    if (!rc) {
        atomic_store(&sh.known[0], 1);
        for (int w = 0; w < n_threads && (size_t)w < sh.n_blocks; w++) {
            ws[w].sh = &sh;
            ws[w].raw_mode = raw_mode;
            ws[w].pack_mode = !raw_mode && out->commit_packed != NULL;
            if (!raw_mode) {
                ws[w].raw_cap = QM2_BLOCK + (1u << 17);
                ws[w].raw = malloc(ws[w].raw_cap);
                if (!ws[w].raw) break;
            }
            sem_init(&ws[w].start, 0, 0);
            sem_init(&ws[w].done, 0, 0);
            if (pthread_create(&tids[w], NULL, pf_main, &ws[w])) { if (!raw_mode) free(ws[w].raw); ws[w].raw = NULL; break; }
            n_workers++;
        }
        if (!n_workers && sh.n_blocks) { puts("Memory allocation failed"); rc = 1; }
    }
    /* issue blocks round-robin, collect in order */
    size_t issued = 0, collected = 0;
    line_state tail = pf_state(&sh, 0);
    int serial_from_here = 0;
    size_t serial_lo = 0;
    while (!rc && collected < sh.n_blocks) {
        while (!serial_from_here && issued < sh.n_blocks && issued - collected < (size_t)n_workers) {
            pf_worker *w = &ws[issued % (size_t)n_workers];
            if (out->acquire(out->user, &w->obuf, &w->ocap)) { rc = 1; break; }
            w->block = issued++;
            sem_post(&w->start);
        }
        if (rc || collected >= issued) break;
        pf_worker *w = &ws[collected % (size_t)n_workers];
        sem_wait(&w->done);
        collected++;
        if (w->rc) { (void)out->commit(out->user, w->obuf, 0); rc = 1; break; }
        if (raw_mode) {
            if (w->too_big) { (void)out->commit(out->user, w->obuf, 0); rc = 2; break; }     /* the caller starts over */
            const line_state st = pf_state(&sh, w->before);
            rc = out->commit_raw(out->user, w->obuf, w->lo, w->hi - w->lo, sh.fastq, (unsigned)st.skip);   /* 0, 1 or 2 */
            continue;
        }
        if (serial_from_here) { (void)out->commit(out->user, w->obuf, 0); continue; }   /* redone below */
        if (w->too_big || w->st_end.desync) {
            /* redo this block serially from the worker's own buffer (oversize lines go through
             * out->submit) and, after a desync, everything behind it from the file */
            (void)out->commit(out->user, w->obuf, 0);
            line_state st = pf_state(&sh, w->before);
            if (strip_range(w->raw + w->lo, w->hi - w->lo, &st, &os)) rc = 1;
            if (!rc && out_flush(&os)) rc = 1;
            if (st.desync) {
                atomic_store(&sh.stop, 1);
                serial_from_here = 1;
                serial_lo = w->file_hi;
                tail = st;
            }
        } else if (w->is_packed ? out->commit_packed(out->user, w->obuf, &w->packed, w->wraps) : out->commit(out->user, w->obuf, w->olen)) rc = 1;
    }
    while (collected < issued) {                                     /* drain what is still in flight */
        pf_worker *w = &ws[collected % (size_t)n_workers];
        sem_wait(&w->done);
        collected++;
        (void)out->commit(out->user, w->obuf, 0);
    }
    if (!rc && serial_from_here) {
        tail.desync = 0;
        if (pf_serial_tail(&sh, serial_lo, &tail, &os)) rc = 1;
    }
    if (!rc) rc = out_flush(&os);
    else if (os.buf) (void)out->commit(out->user, os.buf, 0);
    double t_read = 0, t_count = 0, t_wait = 0, t_pack = 0;
    for (int w = 0; w < n_workers; w++) {
        ws[w].quit = 1;
        sem_post(&ws[w].start);
        pthread_join(tids[w], NULL);
        sem_destroy(&ws[w].start);
        sem_destroy(&ws[w].done);
        if (!raw_mode) free(ws[w].raw);
        t_read += ws[w].t_read; t_count += ws[w].t_count; t_wait += ws[w].t_wait; t_pack += ws[w].t_pack;
    }
    if (getenv("QM2_READER_TIMING"))
        printf("QM2 reader: %d workers, %zu blocks of %zu bytes; seconds summed over the workers: pread %.3f, newline count %.3f, waiting for the line count of the blocks before %.3f, %s %.3f\n",
               n_workers, sh.n_blocks, sh.block, t_read, t_count, t_wait, raw_mode ? "-" : "strip / pack", t_pack);
    free(sh.lines_before); free(sh.known); free(ws); free(tids);
    return rc;
}


/* ---- regular files, packed hand-over: small blocks that stay in the cache, no per-block hand-off ----------------- *
 * (ABI 6.)  With the packing on the host the link no longer bounds the count phase — the reader does: pread's copy out
 * of the page cache, the newline count, the line discipline and the packing all pass over the text.  Measured on the
 * GPU box's host (tools/micro/mmap_probe, reader_probe; profiles/r05_c3/micro_*.txt): pread into a private buffer and
 * packing from it runs at 80-85 GB/s of FASTQ text with 32 MiB blocks and 16 threads, at 140-146 GB/s with 4 MiB blocks
 * (the private buffer stays in the thread's L2 / L3 share: pread writes and the two passes read cache, not DRAM);
 * mapping the file costs more in page-table work than it saves.  Hence this arrangement:
 *   - blocks of 4 MiB (QM2_BLOCK_BYTES overrides), block b OWNS the lines that start inside it (pf_load, as above);
 *   - the workers take blocks from a shared counter themselves; nothing is handed over per block;
 *   - FASTQ: the line state at a block's first line follows from the line count (mod 4) of the blocks before it
 *     (pf_state), which are being read at the same moment by the other workers.  A worker does not wait for it: it packs
 *     under a guess of the state read off its block's first lines — the same single pass tells how many lines the block
 *     holds, mod 4, which is what it passes on — and packs again in the rare case the guess was wrong;
 *   - it packs what the line discipline keeps (qm2_pack_reads) BEHIND the pieces already in its own staging buffer:
 *     a piece is whole 64-position blocks, so pieces simply follow each other; only full buffers travel to the calling
 *     thread, which commits them (qm2_lines_out.commit_packed) as they come — counting does not care about the order;
 *   - a FASTQ sequence-position line that starts with '>' (QuicKmer.c:398: dropped WITHOUT the three discards) makes the
 *     line state of every later block something the line count does not give; pieces of later blocks may already have
 *     been committed by then, so the whole input is given up: return 3, and the caller starts over from zeroed counters
 *     without commit_packed (the serial-exact path above).  A block that does not pack (more run-counter wraps than a
 *     piece's list holds) goes through out->submit as ASCII lines. */
/* ---- where the file's cached pages lie ------------------------------------------------------------------------------ *
 * pread copies the text out of the page cache; when those pages sit on the other socket every byte crosses the
 * socket-to-socket links first (hg38 scale, a 30.7 GB FASTQ in /dev/shm, 16 workers: 93 GB/s = 38.7 G bases/s with the pages on
 * the other node, 48.5 with them on the workers' node; EXPERIMENTS.md, round 5).  The packed output is a fifth of the text, so
 * the workers go where the text is: 64 pages spread over the file are mapped (MAP_POPULATE: never touched from user space, so
 * a file that shrinks under us cannot raise SIGBUS) and move_pages(2) — asked, not told — says which node each lies on.  A
 * clear majority (3/4) on one node whose CPUs the process may use binds the workers there; anything else changes nothing. */
static cpu_set_t g_cpus_before;            /* the process's affinity before anybody narrowed it (qm2_reader_note_affinity) */
static int       g_cpus_before_set;

void qm2_reader_note_affinity(void)
{
    CPU_ZERO(&g_cpus_before);
    g_cpus_before_set = sched_getaffinity(0, sizeof g_cpus_before, &g_cpus_before) == 0;
}

static int file_pages_cpus(int fd, size_t size, cpu_set_t *cpus, int *node_out, int *votes_out)
{
    enum { kSamples = 64, kMaxNodes = 64 };
    if (!g_cpus_before_set || getenv("QM2_NO_BIND")) return 0;
    const long ps = sysconf(_SC_PAGESIZE);
    if (ps <= 0 || size < (size_t)ps * kSamples * 4) return 0;            /* (small files: not worth a thought) */
    int votes[kMaxNodes] = {0}, seen = 0;
    for (int i = 0; i < kSamples; i++) {
        const off_t off = (off_t)((size / kSamples * (size_t)i) & ~((size_t)ps - 1));
        void *m = mmap(NULL, (size_t)ps, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, off);
        if (m == MAP_FAILED) return 0;
        int status = -1;
        void *pages[1] = {m};
        if (syscall(SYS_move_pages, 0, 1UL, pages, NULL, &status, 0) == 0 && status >= 0 && status < kMaxNodes) { votes[status]++; seen++; }
        munmap(m, (size_t)ps);
    }
    int best = -1;
    for (int n = 0; n < kMaxNodes; n++) if (best < 0 || votes[n] > votes[best]) best = n;
    if (best < 0 || seen < kSamples / 2 || 4 * votes[best] < 3 * seen) return 0;
    char path[128], list[4096] = "";
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", best);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    if (!fgets(list, sizeof list, f)) list[0] = 0;
    fclose(f);
    CPU_ZERO(cpus);
    char *save = NULL;
    for (char *tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(NULL, ",\n", &save)) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got == 1) b = a;
        if (got < 1) continue;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &g_cpus_before)) CPU_SET(c, cpus);
    }
    if (CPU_COUNT(cpus) == 0) return 0;
    *node_out = best; *votes_out = votes[best];
    return 1;
}

typedef struct pk_buf {
    char       *buf;                       /* a staging buffer from out->acquire; NULL: `ascii` is malloc'ed lines for out->submit */
    size_t      cap;
    qm2_packed  p;                         /* the pieces in it, as one */
    uint64_t    wraps[QM2_WRAPS_CAP];
    char       *ascii;
    size_t      ascii_len;
    struct pk_buf *next;
} pk_buf;

typedef struct {
    pf_shared        sh;
    atomic_size_t    next_block;
    pthread_mutex_t  mu;
    pthread_cond_t   cv_free, cv_main;
    pk_buf          *free_q, *full_q, *full_tail;
    int              workers_active, failed, desync, closing;
    /* the line count of every block (mod 4), published by whoever packed it; sh.lines_before / sh.known are the prefix sums,
     * advanced by whoever comes by (pk_advance) */
    uint8_t         *cnt;
    atomic_int      *have;
    size_t           frontier;             /* sh.known[frontier] is set; owned by the holder of adv_busy */
    atomic_flag      adv_busy;
    int              near_file;            /* the workers run on the CPUs of the node the file's cached pages lie on */
    cpu_set_t        near_cpus;
} pk_shared;

/* a piece in the buffer in hand that was packed under a GUESSED line state which the blocks before have not confirmed yet */
typedef struct {
    size_t   block;
    int      guess, desync;
    uint64_t first_blk, n_blk, n_newlines; /* where the piece lies in the buffer, and its share of the buffer's n_newlines */
    pk_buf  *ascii;                        /* or: the block as ASCII lines (it did not pack), not queued yet */
} pk_pend;

typedef struct {
    pk_shared *ps; pf_worker w; double t_read, t_count, t_wait, t_pack, t_buf; unsigned long repacked;
    int        index, n_workers, first_buffer;
    pk_pend   *pend; int n_pend, cap_pend;
    size_t    *redo; int n_redo, cap_redo;
} pk_worker;

static pk_buf *pk_get_free(pk_shared *ps)
{
    pthread_mutex_lock(&ps->mu);
    while (!ps->free_q && !ps->closing) { pthread_cond_signal(&ps->cv_main); pthread_cond_wait(&ps->cv_free, &ps->mu); }
    pk_buf *b = ps->free_q;
    if (b) ps->free_q = b->next;
    pthread_mutex_unlock(&ps->mu);
    if (b) { b->next = NULL; memset(&b->p, 0, sizeof b->p); }
    return b;
}

static void pk_put_full(pk_shared *ps, pk_buf *b)
{
    b->next = NULL;
    pthread_mutex_lock(&ps->mu);
    if (ps->full_tail) ps->full_tail->next = b; else ps->full_q = b;
    ps->full_tail = b;
    pthread_cond_signal(&ps->cv_main);
    pthread_mutex_unlock(&ps->mu);
}

/* Which FASTQ line state does a block most likely start in?  Looks at its first lines: with `s` lines to drop before the
 * first sequence line, line i plays role (i - s) mod 4 — 0 sequence, 1 the '+' line, 2 qualities (as long as their
 * sequence), 3 the '@' header.  Only a guess (a quality line may start with '@' or '+'): pk_main packs with it and checks
 * it against the line count of the blocks before, which is what decides (:451-455). */
static int guess_fastq_skip(const char *p, size_t n)
{
    size_t start[12], len[12];
    int nl = 0;
    for (size_t at = 0; at < n && nl < 12; nl++) {
        const char *e = memchr(p + at, '\n', n - at);
        start[nl] = at;
        len[nl] = e ? (size_t)(e - (p + at)) : n - at;
        if (!e) { nl++; break; }
        at += len[nl] + 1;
    }
    for (int s = 0; s < 4; s++) {
        int ok = 1;
        for (int i = 0; i < nl && ok; i++) {
            const int role = ((i - s) % 4 + 4) % 4;
            const char c = len[i] ? p[start[i]] : 0;
            if (role == 1 && c != '+') ok = 0;
            if (role == 3 && c != '@') ok = 0;
            if (role == 2 && i >= 2 && i + 1 < nl && len[i] != len[i - 2]) ok = 0;      /* (the block's last line listed may be cut short) */
        }
        if (ok) return s;
    }
    return 0;
}

/* one block's kept lines behind the pieces already in `cur` (or, what does not pack, as ASCII lines in *ascii_out); the
 * line state *st is updated; 0, or 1 on an error */
static int pk_pack_block(pk_shared *ps, pf_worker *w, pk_buf *cur, line_state *st, qm2_packed *piece, pk_buf **ascii_out)
{
    const size_t n = w->hi - w->lo, used = (size_t)cur->p.n_blocks * QM2_PACKED_BLOCK_BYTES;
    const line_state st0 = *st;
    *ascii_out = NULL;
    memset(piece, 0, sizeof *piece);
    const int prc = qm2_packed_bound(n) <= cur->cap - used
                        ? pack_whole_lines_cap(w->raw + w->lo, n, st, cur->buf + used, cur->cap - used, piece, cur->wraps + cur->p.n_wraps,
                                               QM2_WRAPS_CAP - (size_t)cur->p.n_wraps)
                        : 2;
    if (prc == 1) return 1;
    if (prc == 2) {                                  /* does not pack (or is larger than a whole buffer): as ASCII lines */
        pk_buf *a = calloc(1, sizeof *a);
        char *lines_buf = malloc(n + 2);
        if (!a || !lines_buf) { free(a); free(lines_buf); return 1; }
        static const qm2_lines_out none = {buf_acquire, buf_commit, buf_submit, NULL, NULL, NULL};
        out_state os = {&none, lines_buf, n + 2, 0};
        *st = st0;
        if (strip_range(w->raw + w->lo, n, st, &os)) { free(a); free(lines_buf); return 1; }
        a->ascii = lines_buf; a->ascii_len = os.len;
        *ascii_out = a;
        memset(piece, 0, sizeof *piece);
    }
    (void)ps;
    return 0;
}

/* the prefix sums of the published line counts, as far as they go; whoever finds the flag free does it */
static void pk_advance(pk_shared *ps)
{
    pf_shared *sh = &ps->sh;
    if (atomic_flag_test_and_set_explicit(&ps->adv_busy, memory_order_acquire)) return;
    size_t f = ps->frontier;
    while (f < sh->n_blocks && atomic_load_explicit(&ps->have[f], memory_order_acquire)) {
        sh->lines_before[f + 1] = (sh->lines_before[f] + ps->cnt[f]) & 3;
        atomic_store_explicit(&sh->known[f + 1], 1, memory_order_release);
        f++;
    }
    ps->frontier = f;
    atomic_flag_clear_explicit(&ps->adv_busy, memory_order_release);
}

static void pk_flag_desync(pk_shared *ps)
{
    pthread_mutex_lock(&ps->mu); ps->desync = 1; pthread_mutex_unlock(&ps->mu);
    atomic_store(&ps->sh.stop, 1);
}

/* Settle the pieces of the buffer in hand whose blocks' line state is known by now (wait: all of them — the buffer is about
 * to travel).  A piece packed under the right guess stands; one packed under a wrong guess is voided where it lies — its
 * blocks become 64 resets each, which emit nothing and count as filler — and its block goes on the redo list. */
static void pk_settle(pk_worker *kw, pk_buf *cur, int wait)
{
    pk_shared *ps = kw->ps;
    pf_shared *sh = &ps->sh;
    for (int i = 0; i < kw->n_pend;) {
        pk_pend *e = &kw->pend[i];
        if (!atomic_load_explicit(&sh->known[e->block], memory_order_acquire)) {
            pk_advance(ps);
            if (!atomic_load_explicit(&sh->known[e->block], memory_order_acquire)) {
                if (!wait) { i++; continue; }
                const double t0 = pf_now();
                while (!atomic_load_explicit(&sh->known[e->block], memory_order_acquire) && !atomic_load(&sh->stop)) { sched_yield(); pk_advance(ps); }
                kw->t_wait += pf_now() - t0;
            }
        }
        if (atomic_load(&sh->stop) && !atomic_load_explicit(&sh->known[e->block], memory_order_acquire)) {      /* given up: nothing of this run is kept */
            if (e->ascii) { free(e->ascii->ascii); free(e->ascii); }
            *e = kw->pend[--kw->n_pend];
            continue;
        }
        const int correct = sh->fastq ? (int)((4 - (sh->lines_before[e->block] & 3)) & 3) : 0;
        if (e->guess == correct) {
            if (e->ascii) pk_put_full(ps, e->ascii);
            if (e->desync) pk_flag_desync(ps);           /* the blocks behind this one were given a state that is not theirs */
        } else {
            if (e->ascii) { free(e->ascii->ascii); free(e->ascii); }
            else if (e->n_blk) {
                uint64_t *blk = (uint64_t *)(void *)cur->buf + 3 * e->first_blk;
                for (uint64_t j = 0; j < e->n_blk; j++) { blk[3 * j] = 0; blk[3 * j + 1] = 0; blk[3 * j + 2] = ~0ull; }
                cur->p.n_newlines += 64 * e->n_blk - e->n_newlines;
                /* (its run-counter wraps stay listed: positions at which nothing is emitted any more) */
            }
            if (kw->n_redo < kw->cap_redo) kw->redo[kw->n_redo++] = e->block;
            else { pthread_mutex_lock(&ps->mu); ps->failed = 1; pthread_mutex_unlock(&ps->mu); atomic_store(&sh->stop, 1); }
            kw->repacked++;
        }
        *e = kw->pend[--kw->n_pend];
    }
}

/* the buffer in hand is full (or the file is at its end): settle what it holds and send it to the calling thread */
static void pk_send(pk_worker *kw, pk_buf **cur)
{
    if (!*cur) return;
    pk_settle(kw, *cur, 1);
    pk_put_full(kw->ps, *cur);
    *cur = NULL;
    kw->first_buffer = 0;
}

/* One block: read it, pack the lines that start in it behind the pieces of the buffer in hand.  exact = 0: under a guess of
 * its line state (FASTQ), its line count is published at once and the piece waits for the verdict (pk_settle); exact = 1
 * (a block whose guess was wrong, or FASTA): under the state its line count before gives.  0, or 1 = stop (error, shut down). */
static int pk_do_block(pk_worker *kw, pk_buf **curp, size_t b, int exact)
{
    pk_shared *ps = kw->ps;
    pf_shared *sh = &ps->sh;
    pf_worker *w = &kw->w;
    w->block = b;
    const double t0 = pf_now();
    int rc = atomic_load(&sh->stop) ? 0 : pf_load(w);
    if (rc || atomic_load(&sh->stop)) w->lo = w->hi = 0;
    const double t1 = pf_now();
    kw->t_read += t1 - t0;
    const size_t n = w->hi - w->lo;
    unsigned lines = 0;
    if (!rc && n) {
        pk_buf *cur = *curp;
        if (cur && cur->p.n_blocks &&                        /* (normally settled behind the block before: pk_room) */
            (cur->cap - (size_t)cur->p.n_blocks * QM2_PACKED_BLOCK_BYTES < qm2_packed_bound(n) || cur->p.n_wraps + 8 > QM2_WRAPS_CAP)) pk_send(kw, curp);
        if (!*curp) {
            const double tb = pf_now();
            *curp = pk_get_free(ps);
            kw->t_buf += pf_now() - tb;
            if (!*curp) return 1;
        }
        cur = *curp;
        const double t2 = pf_now();
        line_state st = pf_state(sh, exact ? sh->lines_before[b] : 0);
        const int guess = exact || !sh->fastq ? st.skip : guess_fastq_skip(w->raw + w->lo, n);
        st.skip = guess;
        qm2_packed piece;
        pk_buf *ascii = NULL;
        if (pk_pack_block(ps, w, cur, &st, &piece, &ascii)) rc = 1;
        kw->t_pack += pf_now() - t2;
        if (!rc) {
            lines = st.lines;
            const uint64_t first = cur->p.n_blocks;
            if (!ascii) {
                uint64_t *wr = cur->wraps + cur->p.n_wraps;
                for (uint64_t i = 0; i < piece.n_wraps; i++) wr[i] += 64 * first;
                cur->p.n_blocks += piece.n_blocks; cur->p.n_newlines += piece.n_newlines; cur->p.n_wraps += piece.n_wraps;
            }
            if (exact || !sh->fastq) {                      /* nothing to confirm */
                if (ascii) pk_put_full(ps, ascii);
                if (st.desync) pk_flag_desync(ps);
            } else {
                if (kw->n_pend == kw->cap_pend) {
                    pk_pend *g = realloc(kw->pend, (size_t)(kw->cap_pend ? 2 * kw->cap_pend : 64) * sizeof *g);
                    if (!g) { if (ascii) { free(ascii->ascii); free(ascii); } rc = 1; }
                    else { kw->pend = g; kw->cap_pend = kw->cap_pend ? 2 * kw->cap_pend : 64; }
                }
                if (!rc) {
                    const pk_pend e = {b, guess, st.desync, first, ascii ? 0 : piece.n_blocks, ascii ? 0 : piece.n_newlines, ascii};
                    kw->pend[kw->n_pend++] = e;
                }
            }
        }
    }
    if (!exact) {                                        /* the block's line count, whatever became of it: the blocks behind wait for nothing else */
        ps->cnt[b] = (uint8_t)(lines & 3u);
        atomic_store_explicit(&ps->have[b], 1, memory_order_release);
        pk_advance(ps);
    }
    /* Will the next block fit?  If not the buffer travels NOW, with this block's count already out: settling it may wait for
     * blocks before, and whoever waits for this one should not wait for those as well.  (A worker's FIRST buffer counts as
     * full at (index + 1) / n_workers of its size: all buffers fill at the same pace, and handing them over at the same moment
     * made the workers queue for the one spare buffer — 12 % of their time.) */
    if (!rc && *curp && (*curp)->p.n_blocks) {
        const pk_buf *cur = *curp;
        size_t cap = cur->cap;
        if (kw->first_buffer) cap = cap / (size_t)kw->n_workers * (size_t)(kw->index + 1);
        const size_t used = (size_t)cur->p.n_blocks * QM2_PACKED_BLOCK_BYTES, next = qm2_packed_bound(sh->block + ((size_t)1 << 16));
        if (cap < used || cap - used < next || cur->p.n_wraps + 8 > QM2_WRAPS_CAP) pk_send(kw, curp);
    }
    return rc;
}

/* A worker: blocks from the shared counter, one pass over each while it is hot in the cache.  FASTQ: packed under a GUESS of
 * the block's line state (guess_fastq_skip); the pass also yields the block's line count, which is published at once — it
 * does not depend on the guess — so that nobody ever waits for a block that is still being packed, only (when a buffer is
 * about to travel) for blocks that have not been READ yet.  The verdict on a guess comes from the line counts before it
 * (:451-455); a wrong one (a handful of blocks per file at most: quality lines that look like headers) voids the piece and
 * packs the block again under the right state.  FASTA keeps no state. */
static void *pk_main(void *arg)
{
    pk_worker *kw = arg;
    pk_shared *ps = kw->ps;
    pf_shared *sh = &ps->sh;
    pk_buf *cur = NULL;
    int failed = 0;
    kw->first_buffer = 1;
    if (ps->near_file) (void)pthread_setaffinity_np(pthread_self(), sizeof ps->near_cpus, &ps->near_cpus);      /* (a hint: failure changes nothing) */
    for (;;) {
        while (kw->n_redo && !failed && !atomic_load(&sh->stop))
            if (pk_do_block(kw, &cur, kw->redo[--kw->n_redo], 1)) failed = 1;
        if (failed || atomic_load(&sh->stop)) break;
        const size_t b = atomic_fetch_add(&ps->next_block, 1);
        if (b >= sh->n_blocks) {                         /* the end: what is in hand is settled, which may bring redos */
            if (cur) pk_settle(kw, cur, 1);
            if (kw->n_redo) continue;
            break;
        }
        if (pk_do_block(kw, &cur, b, 0)) { failed = 1; break; }
        if (cur) pk_settle(kw, cur, 0);
    }
    if (failed) atomic_store(&sh->stop, 1);
    /* (after a stop the blocks this worker drew but never packed stay unpublished: nobody waits once stop is set) */
    if (cur) pk_settle(kw, cur, 1);
    for (int i = 0; i < kw->n_pend; i++) if (kw->pend[i].ascii) { free(kw->pend[i].ascii->ascii); free(kw->pend[i].ascii); }
    kw->n_pend = 0;
    pthread_mutex_lock(&ps->mu);
    if (failed) ps->failed = 1;
    if (cur) {                                           /* what is in hand goes back, full or not (an empty one is just returned) */
        cur->next = NULL;
        if (ps->full_tail) ps->full_tail->next = cur; else ps->full_q = cur;
        ps->full_tail = cur;
    }
    ps->workers_active--;
    pthread_cond_signal(&ps->cv_main);
    pthread_mutex_unlock(&ps->mu);
    return NULL;
}

static int stream_blocks_packed(int fd, size_t size, const qm2_lines_out *out, int n_threads)
{
    pk_shared ps;
    memset(&ps, 0, sizeof ps);
    pf_shared *sh = &ps.sh;
    sh->fd = fd; sh->size = size;
    {
        char head[1 << 12];
        size_t off = 0;
        ssize_t r = pread_full(fd, head, sizeof head, 0);
        if (r < 0) return 1;
        if (r > 0 && head[0] == '@') {                       /* :393-395: FASTQ, and that line is consumed */
            sh->fastq = 1;
            for (;;) {
                const char *nl = memchr(head, '\n', (size_t)r);
                if (nl) { sh->start = off + (size_t)(nl - head) + 1; break; }
                off += (size_t)r;
                r = pread_full(fd, head, sizeof head, off);
                if (r <= 0) { sh->start = size; break; }
            }
        }
    }
    const char *env = getenv("QM2_BLOCK_BYTES");
    sh->block = env ? qm2_reader_block_bytes() : ((size_t)4 << 20);
    sh->n_blocks = (size - sh->start + sh->block - 1) / sh->block;
    sh->lines_before = calloc(sh->n_blocks + 2, sizeof(size_t));
    sh->known = calloc(sh->n_blocks + 2, sizeof(atomic_int));
    ps.cnt = calloc(sh->n_blocks + 2, 1);
    ps.have = calloc(sh->n_blocks + 2, sizeof(atomic_int));
    atomic_flag_clear(&ps.adv_busy);
    pk_worker *ws = calloc((size_t)n_threads, sizeof *ws);
    pthread_t *tids = calloc((size_t)n_threads, sizeof *tids);
    pk_buf *bufs = calloc((size_t)n_threads + 1, sizeof *bufs);
    int rc = 0;
    if (!sh->lines_before || !sh->known || !ps.cnt || !ps.have || !ws || !tids || !bufs) { puts("Memory allocation failed"); rc = 1; }
    pthread_mutex_init(&ps.mu, NULL);
    pthread_cond_init(&ps.cv_free, NULL);
    pthread_cond_init(&ps.cv_main, NULL);
    int n_workers = 0, n_bufs = 0;
    int near_node = -1, near_votes = 0;
    if (!rc) ps.near_file = file_pages_cpus(fd, size, &ps.near_cpus, &near_node, &near_votes);
    if (!rc) {
        atomic_store(&sh->known[0], 1);
        /* the buffers the workers fill: as many as there are workers + 1 (what qm2_stream_reads promises to hold at a time) */
        for (int i = 0; i < n_threads + 1 && (size_t)i < sh->n_blocks + 1; i++) {
            if (out->acquire(out->user, &bufs[i].buf, &bufs[i].cap)) { rc = 1; break; }
            n_bufs++;
            bufs[i].next = ps.free_q;
            ps.free_q = &bufs[i];
        }
        ps.workers_active = 0;
        for (int t = 0; t < n_threads && (size_t)t < sh->n_blocks && !rc; t++) {
            ws[t].ps = &ps;
            ws[t].w.sh = sh;
            ws[t].w.raw_cap = sh->block + ((size_t)1 << 20);
            ws[t].w.raw = malloc(ws[t].w.raw_cap);
            ws[t].index = t;
            ws[t].n_workers = n_threads < 1 ? 1 : n_threads;
            ws[t].cap_redo = 64;                             /* (more wrong guesses than this in one buffer: give up) */
            ws[t].redo = malloc((size_t)ws[t].cap_redo * sizeof *ws[t].redo);
            if (!ws[t].w.raw || !ws[t].redo) { free(ws[t].w.raw); free(ws[t].redo); ws[t].w.raw = NULL; ws[t].redo = NULL; break; }
            pthread_mutex_lock(&ps.mu); ps.workers_active++; pthread_mutex_unlock(&ps.mu);
            if (pthread_create(&tids[t], NULL, pk_main, &ws[t])) {
                pthread_mutex_lock(&ps.mu); ps.workers_active--; pthread_mutex_unlock(&ps.mu);
                free(ws[t].w.raw); ws[t].w.raw = NULL;
                free(ws[t].redo); ws[t].redo = NULL;
                break;
            }
            n_workers++;
        }
        if (!n_workers && sh->n_blocks) { puts("Memory allocation failed"); rc = 1; }
    }
    /* the calling thread: commit what comes back full, lend the buffer out again */
    double m_idle = 0, m_commit = 0, m_acquire = 0;
    const double m_t0 = pf_now();
    while (!rc) {
        const double mi = pf_now();
        pthread_mutex_lock(&ps.mu);
        while (!ps.full_q && ps.workers_active > 0) pthread_cond_wait(&ps.cv_main, &ps.mu);
        pk_buf *b = ps.full_q;
        if (b) { ps.full_q = b->next; if (!ps.full_q) ps.full_tail = NULL; }
        const int active = ps.workers_active;
        pthread_mutex_unlock(&ps.mu);
        m_idle += pf_now() - mi;
        if (!b) { if (active == 0) break; continue; }
        const double mc = pf_now();
        if (b->ascii) {                                      /* lines that did not pack */
            if (out->submit(out->user, b->ascii, b->ascii_len)) rc = 1;
            free(b->ascii);
            free(b);
            continue;
        }
        if (b->p.n_blocks ? out->commit_packed(out->user, b->buf, &b->p, b->wraps) : out->commit(out->user, b->buf, 0)) { rc = 1; b->buf = NULL; break; }
        b->buf = NULL;
        m_commit += pf_now() - mc;
        pthread_mutex_lock(&ps.mu);
        const int more = ps.workers_active > 0;
        pthread_mutex_unlock(&ps.mu);
        if (more) {
            const double ma = pf_now();
            const int arc = out->acquire(out->user, &b->buf, &b->cap);
            m_acquire += pf_now() - ma;
            if (arc) { rc = 1; break; }
            pthread_mutex_lock(&ps.mu);
            b->next = ps.free_q; ps.free_q = b;
            pthread_cond_signal(&ps.cv_free);
            pthread_mutex_unlock(&ps.mu);
        }
    }
    /* shut down: wake whoever waits for a buffer, join, give the buffers back */
    if (rc) atomic_store(&sh->stop, 1);
    pthread_mutex_lock(&ps.mu);
    ps.closing = 1;
    pthread_cond_broadcast(&ps.cv_free);
    pthread_mutex_unlock(&ps.mu);
    double t_read = 0, t_wait = 0, t_pack = 0, t_buf = 0;
    unsigned long repacked = 0;
    const double m_loop = pf_now() - m_t0;
    for (int t = 0; t < n_workers; t++) {
        pthread_join(tids[t], NULL);
        free(ws[t].w.raw); free(ws[t].redo); free(ws[t].pend);
        t_read += ws[t].t_read; t_wait += ws[t].t_wait; t_pack += ws[t].t_pack; t_buf += ws[t].t_buf; repacked += ws[t].repacked;
    }
    for (pk_buf *b = ps.full_q; b;) {                        /* (after an error: pieces nobody committed) */
        pk_buf *nx = b->next;
        if (b->ascii) { free(b->ascii); free(b); }
        b = nx;
    }
    for (int i = 0; i < n_bufs; i++) if (bufs[i].buf) (void)out->commit(out->user, bufs[i].buf, 0);
    if (getenv("QM2_READER_TIMING") && ps.near_file)
        printf("QM2 reader: %d of 64 sampled pages of the file lie on NUMA node %d: the workers run on its %d CPUs\n", near_votes, near_node, CPU_COUNT(&ps.near_cpus));
    if (getenv("QM2_READER_TIMING"))
        printf("QM2 reader: %d workers, %zu blocks of %zu bytes, packed; seconds summed over the workers: pread %.3f, line discipline + packing (one pass, under a guessed line state) %.3f, "
               "waiting — only when a buffer is about to travel — for the line count of the blocks before %.3f, for a free staging buffer %.3f; %lu blocks packed again (wrong guess); the calling thread's %.3f s: "
               "waiting for a full buffer %.3f, committing %.3f, acquiring the next %.3f\n",
               n_workers, sh->n_blocks, sh->block, t_read, t_pack, t_wait, t_buf, repacked, m_loop, m_idle, m_commit, m_acquire);
    if (!rc && ps.failed) rc = 1;
    if (!rc && ps.desync) rc = 3;
    pthread_mutex_destroy(&ps.mu);
    pthread_cond_destroy(&ps.cv_free);
    pthread_cond_destroy(&ps.cv_main);
    free(sh->lines_before); free(sh->known); free(ps.cnt); free(ps.have); free(ws); free(tids); free(bufs);
    return rc;
}

int qm2_stream_reads(const char *path, const qm2_lines_out *out, int n_threads)
{
    io_ring ring;
    memset(&ring, 0, sizeof ring);
    ring.fd = open(path, O_RDONLY);
    if (ring.fd < 0) {
        puts("Input open fail");                             /* :339-341 */
        return 1;
    }
    if (n_threads < 0) n_threads = 0;
    if (n_threads > QM2_MAX_WORKERS) n_threads = QM2_MAX_WORKERS;
    if (n_threads > 0 && !getenv("QM2_READER_STREAM")) {
        /* regular FASTQ file: the workers read their own blocks (measured 2.3 -> 6.0 G bases/s at C2).
         * FASTA keeps the streaming path: it has nothing to drop but headers, and one read(2)
         * thread feeding the strip workers is already the faster arrangement there (10 G bases/s). */
        struct stat sb;
        char first = 0;
        if (fstat(ring.fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && pread(ring.fd, &first, 1, 0) == 1 &&
            (first == '@' || getenv("QM2_READER_BLOCKS") || out->commit_raw || out->commit_packed)) {
            const int brc = out->commit_packed && !out->commit_raw && !getenv("QM2_READER_OLD_BLOCKS")
                                ? stream_blocks_packed(ring.fd, (size_t)sb.st_size, out, n_threads)
                                : stream_blocks(ring.fd, (size_t)sb.st_size, out, n_threads);
            close(ring.fd);
            return brc;
        }
    }
    ring.n_slots = n_threads + 3;
#ifdef F_SETPIPE_SZ
    /* a pipe (samtools ... | awk ... | quicKmer2 count ref /dev/fd/0 out, README.md:86-94): as large a kernel buffer as an
     * unprivileged process may ask for (1 MiB by default instead of 64 KiB: fewer hand-overs between the writer and this reader) */
    {
        struct stat psb;
        if (fstat(ring.fd, &psb) == 0 && S_ISFIFO(psb.st_mode)) {
            long want = 1 << 20;
            FILE *mx = fopen("/proc/sys/fs/pipe-max-size", "r");
            if (mx) { long v = 0; if (fscanf(mx, "%ld", &v) == 1 && v > 0) want = v; fclose(mx); }
            (void)fcntl(ring.fd, F_SETPIPE_SZ, (int)want);
        }
    }
#endif
    line_state ls = {1, 0, 0, lseek(ring.fd, 0, SEEK_CUR) != (off_t)-1, 0, 0};
    out_state os = {out, NULL, 0, 0};
    for (int i = 0; i < ring.n_slots; i++) {
        ring.blk[i] = malloc(QM2_RESERVE + QM2_BLOCK + 1);
        if (!ring.blk[i]) { puts("Memory allocation failed"); close(ring.fd); return 1; }
    }
    sem_init(&ring.free_slots, 0, (unsigned)ring.n_slots);
    sem_init(&ring.filled, 0, 0);
    pthread_t io;
    if (pthread_create(&io, NULL, io_main, &ring)) { puts("Thread creation failed"); close(ring.fd); return 1; }

    /* strip workers: job slot w belongs to worker w; jobs are issued and collected round-robin,
     * i.e. in block order */
    strip_job jobs[QM2_MAX_WORKERS];
    pthread_t workers[QM2_MAX_WORKERS];
    int n_workers = 0;
    for (int w = 0; w < n_threads; w++) {
        memset(&jobs[w], 0, sizeof jobs[w]);
        jobs[w].pack_mode = out->commit_packed != NULL;
        sem_init(&jobs[w].start, 0, 0);
        sem_init(&jobs[w].done, 0, 0);
        if (pthread_create(&workers[w], NULL, strip_main, &jobs[w])) break;
        n_workers++;
    }
    int parallel = n_workers > 0;
    unsigned long issued = 0, collected = 0;       /* jobs in flight = issued - collected <= n_workers */
    int release_serial = 0;                        /* a serially processed block waits to be released */

    int rc = 0;
    /* collect the oldest job: commit its output, release its block; on a desync redo what was
     * issued after it serially */
#define QM2_COLLECT()                                                                              \
    do {                                                                                           \
        strip_job *j = &jobs[collected % (unsigned long)n_workers];                                \
        sem_wait(&j->done);                                                                        \
        collected++;                                                                               \
        if (j->rc) rc = 1;                                                                         \
        if (!rc && j->is_packed ? out->commit_packed(out->user, j->obuf, &j->packed, j->wraps)     \
                                : out->commit(out->user, j->obuf, rc ? 0 : j->olen)) rc = 1;       \
        if (!rc && j->st.desync && parallel) {                                                     \
            parallel = 0;                                                                          \
            ls = j->st;                           /* exact state after this block */               \
            while (collected < issued) {          /* later blocks: their start state was a guess */ \
                strip_job *r = &jobs[collected % (unsigned long)n_workers];                        \
                sem_wait(&r->done);                                                                \
                collected++;                                                                       \
                (void)out->commit(out->user, r->obuf, 0);                                          \
                size_t at2 = 0;                                                                    \
                while (!rc && at2 < r->n) {                                                        \
                    const char *nl2 = memchr(r->data + at2, '\n', r->n - at2);                     \
                    const size_t body2 = (size_t)(nl2 - (r->data + at2));                          \
                    if (take_line(&ls, &os, r->data + at2, body2)) rc = 1;                         \
                    at2 += body2 + 1;                                                              \
                }                                                                                  \
                sem_post(&ring.free_slots);                                                        \
            }                                                                                      \
        }                                                                                          \
        sem_post(&ring.free_slots);                                                                \
    } while (0)

    char *acc = NULL;                      /* a line that outgrew the reserve */
    size_t acc_len = 0, acc_cap = 0;
    const char *left = NULL;               /* unfinished line of the previous block */
    size_t left_len = 0;
    for (int i = 0;; i = (i + 1) % ring.n_slots) {
        sem_wait(&ring.filled);
        const ssize_t got = ring.len[i];
        if (got < 0) { puts("Input read fail"); rc = 1; }
        char *data = ring.blk[i] + QM2_RESERVE;
        size_t n = got > 0 ? (size_t)got : 0;
        /* join the previous block's unfinished line in front of this block */
        if (left_len) {
            if (!acc && left_len <= QM2_RESERVE) {
                data -= left_len;
                memcpy(data, left, left_len);
                n += left_len;
            } else {
                if (acc_len + left_len > acc_cap) {
                    acc_cap = (acc_len + left_len) * 2;
                    char *na = realloc(acc, acc_cap);
                    if (!na) { puts("Memory allocation failed"); rc = 1; } else acc = na;
                }
                if (!rc) { memcpy(acc + acc_len, left, left_len); acc_len += left_len; }
            }
            left_len = 0;
        }
        /* The I/O thread refills slots in ring order, so blocks must be released in block order,
         * and only once nothing points into them: a serially processed block is released here,
         * after its unfinished line has been copied out and the jobs before it are collected. */
        if (release_serial) {
            while (collected < issued) QM2_COLLECT();
            sem_post(&ring.free_slots);
            release_serial = 0;
        }
        if (rc) break;
        size_t at = 0;
        int dispatched = 0;
        if (acc) {                                           /* finish the over-long line first */
            while (collected < issued) QM2_COLLECT();        /* (keeps the order of the serial path) */
            char *nl = n ? memchr(data, '\n', n) : NULL;
            const size_t head = nl ? (size_t)(nl - data) : n;
            if (acc_len + head + 1 > acc_cap) {
                acc_cap = (acc_len + head + 1) * 2;
                char *na = realloc(acc, acc_cap);
                if (!na) { puts("Memory allocation failed"); rc = 1; break; }
                acc = na;
            }
            memcpy(acc + acc_len, data, head);
            acc_len += head;
            if (nl || got == 0) {
                if (take_line(&ls, &os, acc, acc_len)) { rc = 1; break; }
                free(acc);
                acc = NULL;
                acc_len = acc_cap = 0;
                at = nl ? head + 1 : n;
            } else at = n;
        }
        if (ls.first_line && at < n) {                       /* the first line decides the format, serially */
            char *nl = memchr(data + at, '\n', n - at);
            if (nl) {
                if (take_line(&ls, &os, data + at, (size_t)(nl - (data + at)))) { rc = 1; break; }
                at = (size_t)(nl - data) + 1;
            }
        }
        if (parallel && !ls.first_line && !ls.desync && at < n) {
            /* hand the whole lines of this block to a worker */
            const char *last = memrchr(data + at, '\n', n - at);
            if (last) {
                if (os.buf && out_flush(&os)) { rc = 1; break; }            /* keep commits in order */
                if (issued - collected == (unsigned long)n_workers) QM2_COLLECT();
                if (rc) break;
            }
            if (last && parallel) {
                strip_job *j = &jobs[issued % (unsigned long)n_workers];
                j->data = data + at;
                j->n = (size_t)(last - (data + at)) + 1;
                j->st = ls;
                if (out->acquire(out->user, &j->obuf, &j->ocap)) { rc = 1; break; }
                if (j->ocap < j->n) {                        /* cannot happen with 64 MiB staging; be safe */
                    (void)out->commit(out->user, j->obuf, 0);
                } else {
                    /* state after this block: one step per line (:451-455) */
                    if (ls.fastq) {
                        const size_t lines = count_newlines(j->data, j->n);
                        ls.skip = (int)(((size_t)ls.skip + 4 - (lines & 3)) & 3);
                    }
                    sem_post(&j->start);
                    issued++;
                    dispatched = 1;
                    at += j->n;
                }
            }
        }
        if (!dispatched) {
            while (at < n) {
                char *nl = memchr(data + at, '\n', n - at);
                if (!nl) break;
                const size_t body = (size_t)(nl - (data + at));
                if (take_line(&ls, &os, data + at, body)) { rc = 1; break; }
                at += body + 1;
            }
        }
        if (rc) break;
        if (got == 0) {                                      /* EOF: a last line without '\n' still counts */
            while (collected < issued) QM2_COLLECT();
            if (!rc && at < n && take_line(&ls, &os, data + at, n - at)) rc = 1;
            break;
        }
        left = data + at;
        left_len = n - at;
        if (left_len > QM2_RESERVE && !acc) {                /* will not fit in front of the next block */
            acc_cap = left_len * 2;
            acc = malloc(acc_cap);
            if (!acc) { puts("Memory allocation failed"); rc = 1; break; }
            acc_len = 0;
        }
        if (!dispatched) release_serial = 1;
    }
    while (collected < issued) QM2_COLLECT();
#undef QM2_COLLECT
    if (!rc) rc = out_flush(&os);
    else if (os.buf) (void)out->commit(out->user, os.buf, 0);
    /* shut the threads down */
    for (int w = 0; w < n_workers; w++) {
        jobs[w].quit = 1;
        sem_post(&jobs[w].start);
        pthread_join(workers[w], NULL);
        sem_destroy(&jobs[w].start);
        sem_destroy(&jobs[w].done);
    }
    if (rc) close(ring.fd);
    for (int k = 0; k < ring.n_slots + 1; k++) sem_post(&ring.free_slots);
    pthread_join(io, NULL);
    if (!rc) close(ring.fd);
    free(acc);
    for (int i = 0; i < ring.n_slots; i++) free(ring.blk[i]);
    sem_destroy(&ring.free_slots);
    sem_destroy(&ring.filled);
    return rc;
}
