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
#include <fcntl.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "qm2_host.h"

#define QM2_BLOCK   ((size_t)32 << 20)     /* bytes per read(2) block                          */
#define QM2_RESERVE ((size_t)1 << 20)      /* room in front of a block for the unfinished line
                                              of the block before it                          */
#define QM2_RING    3

/* ---- the I/O thread -------------------------------------------------------- */

typedef struct {
    int     fd;
    char   *blk[QM2_RING];                 /* each QM2_RESERVE + QM2_BLOCK bytes               */
    ssize_t len[QM2_RING];                 /* bytes read into blk[i] + QM2_RESERVE; 0 = EOF, <0 = error */
    sem_t   free_slots, filled;
} io_ring;

static void *io_main(void *arg)
{
    io_ring *r = (io_ring *)arg;
    for (int i = 0;; i = (i + 1) % QM2_RING) {
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
            i = (i + 1) % QM2_RING;
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
} line_state;

static int take_line(line_state *s, out_state *o, const char *line, size_t body)
{
    if (s->first_line) {
        s->first_line = 0;
        if (body && line[0] == '@') { s->fastq = 1; return 0; }   /* :395 */
        if (!s->seekable) return 0;                               /* :396 on a pipe */
    }
    if (s->skip) { s->skip--; return 0; }                         /* :452-454 */
    if (body && line[0] == '>') return 0;                         /* :398 */
    if (out_line(o, line, body)) return 1;
    if (s->fastq) s->skip = 3;
    return 0;
}

int qm2_stream_reads(const char *path, const qm2_lines_out *out)
{
    io_ring ring;
    memset(&ring, 0, sizeof ring);
    ring.fd = open(path, O_RDONLY);
    if (ring.fd < 0) {
        puts("Input open fail");                             /* :339-341 */
        return 1;
    }
    line_state ls = {1, 0, 0, lseek(ring.fd, 0, SEEK_CUR) != (off_t)-1};
    out_state os = {out, NULL, 0, 0};
    for (int i = 0; i < QM2_RING; i++) {
        ring.blk[i] = malloc(QM2_RESERVE + QM2_BLOCK + 1);
        if (!ring.blk[i]) { puts("Memory allocation failed"); close(ring.fd); return 1; }
    }
    sem_init(&ring.free_slots, 0, QM2_RING);
    sem_init(&ring.filled, 0, 0);
    pthread_t io;
    if (pthread_create(&io, NULL, io_main, &ring)) { puts("Thread creation failed"); close(ring.fd); return 1; }

    int rc = 0;
    char *acc = NULL;                      /* a line that outgrew the reserve */
    size_t acc_len = 0, acc_cap = 0;
    const char *left = NULL;               /* unfinished line of the previous block */
    size_t left_len = 0;
    int prev = -1;
    for (int i = 0;; i = (i + 1) % QM2_RING) {
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
        if (prev >= 0) sem_post(&ring.free_slots);          /* the previous block is no longer referenced */
        prev = i;
        if (rc) break;
        size_t at = 0;
        if (acc) {                                           /* finish the over-long line first */
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
        while (at < n) {
            char *nl = memchr(data + at, '\n', n - at);
            if (!nl) break;
            const size_t body = (size_t)(nl - (data + at));
            if (take_line(&ls, &os, data + at, body)) { rc = 1; break; }
            at += body + 1;
        }
        if (rc) break;
        if (got == 0) {                                      /* EOF: a last line without '\n' still counts */
            if (at < n && take_line(&ls, &os, data + at, n - at)) rc = 1;
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
    }
    if (!rc) rc = out_flush(&os);
    else if (os.buf) (void)out->commit(out->user, os.buf, 0);
    /* let the I/O thread run out */
    if (rc) close(ring.fd);
    for (int k = 0; k < QM2_RING + 1; k++) sem_post(&ring.free_slots);
    pthread_join(io, NULL);
    if (!rc) close(ring.fd);
    free(acc);
    for (int i = 0; i < QM2_RING; i++) free(ring.blk[i]);
    sem_destroy(&ring.free_slots);
    sem_destroy(&ring.filled);
    return rc;
}
