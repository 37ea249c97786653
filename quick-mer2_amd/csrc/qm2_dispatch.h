/*
 * qm2_dispatch.h — one reader, several counting contexts (`quicKmer2 count --devices 0,1,...`).
 *
 * The reference's only parallelism knob is -t N (QuicKmer.c:304-333): one producer hands 4096-k-mer
 * batches to the first idle of N workers that share one table (:421-439).  Here the workers are whole
 * GPUs with a replica of the table each (it always fits: SURVEY section 0): the one reader's blocks go
 * to the contexts in turns of one device batch, and the counters are summed on the first device before
 * the epilogue (integer adds commute: exact).  Plain C, no GPU call in here: the sinks are function
 * pointers (the CLI binds them to the C-ABI of include/qm2.h; the CPU test suite to a harness).
 *
 * Two kinds of turn:
 *   stripped lines (commit / submit): a turn is a number of BYTES; a buffer is committed to the sink that
 *     lent it, whoever's turn it is by then (sequence lines may be counted anywhere, in any order);
 *   raw text (commit_raw, qm2_dispatch_raw_turn): a turn is a number of BUFFERS, fixed when they are
 *     acquired — buffer j of the reader belongs to sink (j / per_turn) % n — because the device line
 *     discipline numbers the lines of a batch from its first block on: a context's batch must hold
 *     consecutive blocks of the text, so a turn's blocks all go to one sink, which is flushed behind the
 *     last of them, and the next turn of that sink starts a batch with a line state of its own.
 */
#ifndef QM2_DISPATCH_H
#define QM2_DISPATCH_H
#include <stddef.h>
#include <stdint.h>

#include "../../include/qm2.h"

typedef struct {
    int (*acquire)(void *user, char **buf, size_t *capacity);
    int (*commit)(void *user, char *buf, size_t n);
    int (*submit)(void *user, const char *lines, size_t n);
    int (*flush)(void *user);          /* launch what is pending, do not wait for it */
    void *user;
    /* optional: raw text with the line state at its first byte (qm2_lines_out.commit_raw); 0, 1 or 2 */
    int (*commit_raw)(void *user, char *buf, size_t offset, size_t n, int fastq, unsigned skip);
    /* optional: a packed piece of sequence lines (qm2_lines_out.commit_packed); dealt like stripped lines, a turn
     * being counted in positions */
    int (*commit_packed)(void *user, char *buf, const qm2_packed *p, const uint64_t *wraps);
} qm2_sink;

typedef struct qm2_dispatch qm2_dispatch;

/* bytes_per_turn: how much one sink is given before the next one's turn (a little less than a device batch,
 * so that the flush at the end of the turn launches the whole batch and not what the context's own
 * batch-full flush left over) */
qm2_dispatch *qm2_dispatch_new(const qm2_sink *sinks, int n_sinks, size_t bytes_per_turn);
/* raw hand-over: turns of `buffers_per_turn` acquired buffers (see above); call before the first acquire */
void qm2_dispatch_raw_turn(qm2_dispatch *d, size_t buffers_per_turn);
void qm2_dispatch_free(qm2_dispatch *d);

/* the callbacks of qm2_lines_out (qm2_host.h), `user` = the qm2_dispatch */
int qm2_dispatch_acquire(void *d, char **buf, size_t *capacity);
int qm2_dispatch_commit(void *d, char *buf, size_t n);
int qm2_dispatch_submit(void *d, const char *lines, size_t n);
int qm2_dispatch_commit_raw(void *d, char *buf, size_t offset, size_t n, int fastq, unsigned skip);
int qm2_dispatch_commit_packed(void *d, char *buf, const qm2_packed *p, const uint64_t *wraps);
#endif
