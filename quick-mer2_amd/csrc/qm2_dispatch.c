/* qm2_dispatch.c — see qm2_dispatch.h */
#include "qm2_dispatch.h"

#include <stdio.h>
#include <stdlib.h>

#define QM2_MAX_LENT 128

struct qm2_dispatch {
    qm2_sink *sinks;
    int n, cur;
    size_t per_turn, turn_bytes;
    size_t raw_per_turn;                   /* > 0: turns are counted in acquired buffers */
    unsigned long acquired;                /* buffers handed out so far */
    struct { char *buf; int sink; unsigned long seq; } lent[QM2_MAX_LENT];
};

qm2_dispatch *qm2_dispatch_new(const qm2_sink *sinks, int n_sinks, size_t bytes_per_turn)
{
    if (n_sinks < 1) return NULL;
    qm2_dispatch *d = calloc(1, sizeof *d);
    if (!d) return NULL;
    d->sinks = malloc((size_t)n_sinks * sizeof *sinks);
    if (!d->sinks) { free(d); return NULL; }
    for (int i = 0; i < n_sinks; i++) d->sinks[i] = sinks[i];
    d->n = n_sinks;
    d->per_turn = bytes_per_turn ? bytes_per_turn : 1;
    return d;
}

void qm2_dispatch_raw_turn(qm2_dispatch *d, size_t buffers_per_turn)
{
    d->raw_per_turn = buffers_per_turn ? buffers_per_turn : 1;
}

void qm2_dispatch_free(qm2_dispatch *d)
{
    if (!d) return;
    free(d->sinks);
    free(d);
}

/* the current sink has had its share: launch it and move on */
static int turn_over(qm2_dispatch *d, size_t n)
{
    d->turn_bytes += n;
    if (d->turn_bytes < d->per_turn || d->n == 1) return 0;
    qm2_sink *s = &d->sinks[d->cur];
    if (s->flush && s->flush(s->user)) return 1;
    d->cur = (d->cur + 1) % d->n;
    d->turn_bytes = 0;
    return 0;
}

int qm2_dispatch_acquire(void *user, char **buf, size_t *capacity)
{
    qm2_dispatch *d = user;
    int slot = -1;
    for (int i = 0; i < QM2_MAX_LENT; i++) if (!d->lent[i].buf) { slot = i; break; }
    if (slot < 0) { puts("qm2: too many staging buffers held at once"); return 1; }
    const int owner = d->raw_per_turn ? (int)((d->acquired / d->raw_per_turn) % (unsigned long)d->n) : d->cur;
    qm2_sink *s = &d->sinks[owner];
    if (s->acquire(s->user, buf, capacity)) return 1;
    d->lent[slot].buf = *buf;
    d->lent[slot].sink = owner;
    d->lent[slot].seq = d->acquired++;
    return 0;
}

/* raw turns: the buffer that was acquired last in its turn closes the turn when it comes back */
static int raw_turn_over(qm2_dispatch *d, int sink, unsigned long seq)
{
    if (d->n == 1 || (seq + 1) % d->raw_per_turn != 0) return 0;
    qm2_sink *s = &d->sinks[sink];
    return s->flush ? s->flush(s->user) : 0;
}

int qm2_dispatch_commit(void *user, char *buf, size_t n)
{
    qm2_dispatch *d = user;
    for (int i = 0; i < QM2_MAX_LENT; i++) {
        if (d->lent[i].buf != buf) continue;
        const int sink = d->lent[i].sink;
        const unsigned long seq = d->lent[i].seq;
        qm2_sink *s = &d->sinks[sink];
        const int was_current = sink == d->cur;
        d->lent[i].buf = NULL;
        if (s->commit(s->user, buf, n)) return 1;
        if (d->raw_per_turn) return raw_turn_over(d, sink, seq);     /* (a buffer given back unused: n = 0) */
        return was_current ? turn_over(d, n) : 0;
    }
    puts("qm2: commit of a buffer that was not acquired");
    return 1;
}

int qm2_dispatch_commit_packed(void *user, char *buf, const qm2_packed *p, const uint64_t *wraps)
{
    qm2_dispatch *d = user;
    for (int i = 0; i < QM2_MAX_LENT; i++) {
        if (d->lent[i].buf != buf) continue;
        const int sink = d->lent[i].sink;
        qm2_sink *s = &d->sinks[sink];
        const int was_current = sink == d->cur;
        d->lent[i].buf = NULL;
        if (!s->commit_packed) { puts("qm2: packed commit to a sink that takes none"); return 1; }
        if (s->commit_packed(s->user, buf, p, wraps)) return 1;
        return was_current ? turn_over(d, (size_t)p->n_blocks * 64) : 0;
    }
    puts("qm2: commit of a buffer that was not acquired");
    return 1;
}

int qm2_dispatch_commit_raw(void *user, char *buf, size_t offset, size_t n, int fastq, unsigned skip)
{
    qm2_dispatch *d = user;
    for (int i = 0; i < QM2_MAX_LENT; i++) {
        if (d->lent[i].buf != buf) continue;
        const int sink = d->lent[i].sink;
        const unsigned long seq = d->lent[i].seq;
        qm2_sink *s = &d->sinks[sink];
        d->lent[i].buf = NULL;
        if (!s->commit_raw) { puts("qm2: raw commit to a sink that takes none"); return 1; }
        const int rc = s->commit_raw(s->user, buf, offset, n, fastq, skip);
        if (rc) return rc;
        return raw_turn_over(d, sink, seq);
    }
    puts("qm2: commit of a buffer that was not acquired");
    return 1;
}

int qm2_dispatch_submit(void *user, const char *lines, size_t n)
{
    qm2_dispatch *d = user;
    qm2_sink *s = &d->sinks[d->cur];
    if (s->submit(s->user, lines, n)) return 1;
    return turn_over(d, n);
}
