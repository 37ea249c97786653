/* qm2_dispatch.c — see qm2_dispatch.h */
#include "qm2_dispatch.h"

#include <stdio.h>
#include <stdlib.h>

#define QM2_MAX_LENT 64

struct qm2_dispatch {
    qm2_sink *sinks;
    int n, cur;
    size_t per_turn, turn_bytes;
    struct { char *buf; int sink; } lent[QM2_MAX_LENT];
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
    qm2_sink *s = &d->sinks[d->cur];
    if (s->acquire(s->user, buf, capacity)) return 1;
    d->lent[slot].buf = *buf;
    d->lent[slot].sink = d->cur;
    return 0;
}

int qm2_dispatch_commit(void *user, char *buf, size_t n)
{
    qm2_dispatch *d = user;
    for (int i = 0; i < QM2_MAX_LENT; i++) {
        if (d->lent[i].buf != buf) continue;
        qm2_sink *s = &d->sinks[d->lent[i].sink];
        const int was_current = d->lent[i].sink == d->cur;
        d->lent[i].buf = NULL;
        if (s->commit(s->user, buf, n)) return 1;
        return was_current ? turn_over(d, n) : 0;
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
