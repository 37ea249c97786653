/*
 * qm2_reader.c — block reader that reduces a FASTA/FASTQ stream to the
 * sequence lines main_count feeds its encoder (QuicKmer.c:393-398, :451-455).
 *
 * The reference does per-base work on its single producer thread, which is
 * what caps it at ~6 threads (README.md:95).  Here the host only finds line
 * boundaries (memchr) and copies sequence lines; every per-base operation
 * happens on the GPU.
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
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "qm2_host.h"

typedef struct {
    char  *buf;
    size_t len, cap;
    size_t batch;
    qm2_lines_sink sink;
    void  *user;
} out_batch;

static int batch_flush(out_batch *o)
{
    if (o->len == 0) return 0;
    int rc = o->sink(o->user, o->buf, o->len);
    o->len = 0;
    return rc;
}

static int batch_add(out_batch *o, const char *line, size_t body)
{
    if (o->len && o->len + body + 1 > o->batch && batch_flush(o)) return 1;
    if (o->len + body + 1 > o->cap) {
        size_t cap = o->cap;
        while (o->len + body + 1 > cap) cap *= 2;
        char *nb = realloc(o->buf, cap);
        if (!nb) { puts("Memory allocation failed"); return 1; }
        o->buf = nb;
        o->cap = cap;
    }
    memcpy(o->buf + o->len, line, body);
    o->len += body;
    o->buf[o->len++] = '\n';
    return 0;
}

int qm2_stream_reads(const char *path, size_t batch_bytes, qm2_lines_sink sink, void *user)
{
    int fd = open(path, O_RDONLY);
    if (fd < 0) {
        puts("Input open fail");                             /* :339-341 */
        return 1;
    }
    const int seekable = lseek(fd, 0, SEEK_CUR) != (off_t)-1;
    size_t cap = (size_t)32 << 20, have = 0;
    char *blk = malloc(cap);
    out_batch o = {malloc(batch_bytes + 1), 0, batch_bytes + 1, batch_bytes, sink, user};
    if (!blk || !o.buf) { puts("Memory allocation failed"); free(blk); free(o.buf); close(fd); return 1; }

    int first_line = 1, fastq = 0, skip = 0, rc = 0, eof = 0;
    while (!eof && !rc) {
        if (have == cap) {                                   /* one line longer than the block */
            char *nb = realloc(blk, cap * 2);
            if (!nb) { puts("Memory allocation failed"); rc = 1; break; }
            blk = nb;
            cap *= 2;
        }
        ssize_t got = read(fd, blk + have, cap - have);
        if (got < 0) {
            if (errno == EINTR) continue;
            puts("Input read fail");
            rc = 1;
            break;
        }
        if (got == 0) {
            eof = 1;
            if (have && blk[have - 1] != '\n') blk[have++] = '\n';   /* cap > have here */
        }
        have += (size_t)got;
        size_t at = 0;
        while (at < have) {
            char *nl = memchr(blk + at, '\n', have - at);
            if (!nl) break;
            const char *line = blk + at;
            const size_t body = (size_t)(nl - line);
            at += body + 1;
            if (first_line) {
                first_line = 0;
                if (line[0] == '@') { fastq = 1; continue; }          /* :395 */
                if (!seekable) continue;                              /* :396 on a pipe */
            }
            if (skip) { skip--; continue; }                           /* :452-454 */
            if (line[0] == '>') continue;                             /* :398 */
            if (batch_add(&o, line, body)) { rc = 1; break; }
            if (fastq) skip = 3;
        }
        memmove(blk, blk + at, have - at);
        have -= at;
    }
    if (!rc) rc = batch_flush(&o);
    free(blk);
    free(o.buf);
    close(fd);
    return rc;
}
