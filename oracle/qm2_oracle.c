/*
 * qm2_oracle.c — CPU restatement of QuicK-mer2's `count` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see qm2_oracle.h).  Written from the behavioural
 * specification in SURVEY.md Appendix B; each function cites the lines of the
 * reference's QuicKmer.c whose behaviour it restates.  Parity: PINNED against
 * the compiled reference (oracle/_ref) — see tests/golden/.
 */
#define _GNU_SOURCE
#include "qm2_oracle.h"

#include <math.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define QM2O_BATCH 4096u          /* FIFO_size, :12 */
#define QM2O_BIN_CHUNK (1u << 20) /* buffer_size, :11 */

/* ---- hash + probe --------------------------------------------------------- */

/* :66-76 — DJB2 over the eight little-endian bytes, 64-bit wrap-around. */
uint64_t qm2o_djb(uint64_t kmer)
{
    uint64_t h = 5381;
    for (int b = 0; b < 8; b++) {
        h = h * 33u + (kmer & 0xFFu);
        kmer >>= 8;
    }
    return h;
}

/* :90-99 — home slot from the hash, walk towards the middle of the table
 * until the slot is empty or holds the key.  No bounds check (as the
 * reference): callers guarantee an empty slot before the table edge. */
int qm2o_find(const uint64_t *table, uint64_t hash_size, uint64_t kmer,
              uint64_t *idx, uint64_t *probes)
{
    uint64_t i = qm2o_djb(kmer) & (hash_size - 1);
    const int64_t step = (i & (hash_size >> 1)) ? -1 : 1;
    uint64_t seen = 1;
    while (table[i] != 0 && table[i] != kmer) {
        i += (uint64_t)step;
        seen++;
    }
    *idx = i;
    if (probes) *probes += seen;
    return table[i] == kmer;
}

/* ---- encoder -------------------------------------------------------------- */

typedef struct {
    uint64_t fwd, rc;
    uint16_t run; /* cur_chars is a uint16_t in the reference (:402): it wraps */
} enc_state;

static inline uint64_t kmask(uint32_t k)
{
    /* :419 computes ((uint64_t)1 << (k<<1)) - 1; k==32 is undefined there. */
    return (k >= 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
}

/* One character of a sequence line, :404-420.  Returns 1 and sets *out when a
 * k-mer is emitted. */
static inline int enc_step(enc_state *s, unsigned char c, uint32_t k,
                           uint64_t mask, uint64_t *out)
{
    if (c == 'N') { /* :404-408 — uppercase N is the only reset character */
        s->fwd = 0; s->rc = 0; s->run = 0;
        return 0;
    }
    s->run++;                                   /* :410 (mod 65536) */
    uint64_t code = (c >> 1) & 3u;              /* :411 */
    s->fwd = (s->fwd << 2) | code;              /* :412-413 */
    uint64_t comp = (code - 2u) & 3u;           /* :414 */
    s->rc = (s->rc | (comp << 60)) >> 2;        /* :415-416, 60 is hard-coded */
    if (s->run < k) return 0;                   /* :418 */
    uint64_t km = s->fwd & mask;                /* :419 */
    if (km > s->rc) km = s->rc;                 /* :420 */
    *out = km;
    return 1;
}

uint64_t qm2o_extract(const char *lines, size_t n, uint32_t k,
                      uint64_t *out, uint64_t cap, uint64_t *n_bases)
{
    const uint64_t mask = kmask(k);
    uint64_t emitted = 0, bases = 0;
    enc_state s = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        unsigned char c = (unsigned char)lines[i];
        if (c == '\n') { s.fwd = 0; s.rc = 0; s.run = 0; continue; } /* :400-403 */
        bases++;
        uint64_t km;
        if (enc_step(&s, c, k, mask, &km)) {
            if (out && emitted < cap) out[emitted] = km;
            emitted++;
        }
    }
    if (n_bases) *n_bases += bases;
    return emitted;
}

/* ---- serial count --------------------------------------------------------- */

int qm2o_count_lines(const uint64_t *table, uint64_t hash_size, uint32_t k,
                     const char *lines, size_t n, uint16_t *depth,
                     qm2o_stats *st)
{
    const uint64_t mask = kmask(k);
    enc_state s = {0, 0, 0};
    uint64_t bases = 0, kmers = 0, probes = 0, hits = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char c = (unsigned char)lines[i];
        if (c == '\n') { s.fwd = 0; s.rc = 0; s.run = 0; continue; }
        bases++;
        uint64_t km, slot;
        if (!enc_step(&s, c, k, mask, &km)) continue;
        kmers++;                                        /* :445 */
        if (qm2o_find(table, hash_size, km, &slot, &probes)) { /* :442 */
            depth[slot]++;                              /* :443, u16 wrap */
            hits++;
        }
    }
    if (st) {
        st->n_bases += bases; st->n_kmers += kmers;
        st->n_probes += probes; st->n_hits += hits;
    }
    return 0;
}

/* ---- threaded count (reference model) ------------------------------------- */

typedef struct {
    uint64_t fifo[2][QM2O_BATCH];   /* FIFO0/FIFO1, :35-36 */
    sem_t feed;                     /* data_feed_sem, :37 */
    volatile uint8_t written;       /* Write_count, :38 */
    volatile uint8_t consumed;      /* Read_count, :39 */
    const uint64_t *table;
    uint64_t hash_size;
    uint16_t *depth;
    volatile int *stop;
    uint64_t hits;
} worker_t;

/* Kmer_count_TSK, :256-296 */
static void *worker_main(void *arg)
{
    worker_t *w = (worker_t *)arg;
    uint64_t hits = 0;
    for (;;) {
        sem_wait(&w->feed);                              /* :266 */
        if (*w->stop) break;                             /* :267 */
        const uint64_t *batch = w->fifo[w->consumed & 1];/* :284-285 */
        for (uint32_t i = 0; i < QM2O_BATCH; i++) {      /* :287 */
            uint64_t slot;
            if (qm2o_find(w->table, w->hash_size, batch[i], &slot, NULL)) {
                __sync_fetch_and_add(&w->depth[slot], 1);/* :291 */
                hits++;
            }
        }
        __sync_synchronize();
        w->consumed++;                                   /* :293 */
    }
    w->hits = hits;
    return NULL;
}

int qm2o_count_lines_mt(const uint64_t *table, uint64_t hash_size, uint32_t k,
                        const char *lines, size_t n, uint16_t *depth,
                        int nthreads, qm2o_stats *st)
{
    if (nthreads <= 0)
        return qm2o_count_lines(table, hash_size, k, lines, n, depth, st);

    volatile int stop = 0;
    pthread_t *tid = malloc((size_t)nthreads * sizeof *tid);
    worker_t *w = NULL;
    if (!tid || posix_memalign((void **)&w, 64, (size_t)nthreads * sizeof *w)) {
        puts("Memory allocation failed");
        free(tid);
        return 1;
    }
    for (int t = 0; t < nthreads; t++) {                 /* :376-383 */
        w[t].written = 0; w[t].consumed = 0;
        w[t].table = table; w[t].hash_size = hash_size; w[t].depth = depth;
        w[t].stop = &stop; w[t].hits = 0;
        sem_init(&w[t].feed, 0, 0);
        pthread_create(&tid[t], NULL, worker_main, &w[t]);
    }

    const uint64_t mask = kmask(k);
    enc_state s = {0, 0, 0};
    uint64_t bases = 0, kmers = 0;
    int cur = 0;
    uint32_t fill = 0;
    for (size_t i = 0; i < n; i++) {
        unsigned char c = (unsigned char)lines[i];
        if (c == '\n') { s.fwd = 0; s.rc = 0; s.run = 0; continue; }
        bases++;
        uint64_t km;
        if (!enc_step(&s, c, k, mask, &km)) continue;
        kmers++;
        w[cur].fifo[w[cur].written & 1][fill++] = km;    /* :424-427 */
        if (fill == QM2O_BATCH) {                        /* :428-438 */
            fill = 0;
            __sync_synchronize();
            w[cur].written++;
            sem_post(&w[cur].feed);
            do {
                if (++cur == nthreads) cur = 0;
            } while (w[cur].written != w[cur].consumed); /* spin for an idle worker */
        }
    }
    /* Drain, :458-479: zero-pad the last batch (k-mer 0 can only "hit" an
     * empty slot, which is never on the chain) and post it. */
    while (fill != QM2O_BATCH) w[cur].fifo[w[cur].written & 1][fill++] = 0;
    __sync_synchronize();
    w[cur].written++;
    sem_post(&w[cur].feed);
    /* The reference sleeps one second here (:469) and hopes the workers are
     * done; wait for them exactly instead. */
    for (int t = 0; t < nthreads; t++)
        while (w[t].written != w[t].consumed) sched_yield();
    stop = 1;                                            /* :470 */
    uint64_t hits = 0;
    for (int t = 0; t < nthreads; t++) {                 /* :471-476 */
        sem_post(&w[t].feed);
        pthread_join(tid[t], NULL);
        sem_destroy(&w[t].feed);
        hits += w[t].hits;
    }
    free(tid);
    free(w);
    if (st) {
        st->n_bases += bases; st->n_kmers += kmers; st->n_hits += hits;
        /* hits include zero-padding matches on an empty slot; n_probes not
         * collected in this mode. */
    }
    return 0;
}

/* ---- .qm dictionary ------------------------------------------------------- */

int qm2o_dict_load(const char *qm_path, qm2o_dict *d)
{
    memset(d, 0, sizeof *d);
    FILE *f = fopen(qm_path, "rb");
    if (!f) { printf("Cannot open %s\n", qm_path); return 1; }
    unsigned char hdr[24];
    if (fread(hdr, 1, 24, f) != 24) { fclose(f); puts("Short .qm header"); return 1; }
    d->k = hdr[4];                                       /* :345-347 */
    memcpy(d->hdr_bytes, hdr + 4, 4);
    memcpy(&d->hash_size, hdr + 8, 8);                   /* :348-349 */
    memcpy(&d->first_idx, hdr + 16, 8);                  /* :350-351 */
    uint64_t H = d->hash_size;
    d->keys = malloc(H * sizeof(uint64_t));              /* :353 */
    d->chain = malloc(H * sizeof(uint32_t));
    if (!d->keys || !d->chain) {
        puts("Memory allocation failed");                /* :355 */
        fclose(f); qm2o_dict_free(d); return 1;
    }
    if (fread(d->keys, 8, H, f) != H || fread(d->chain, 4, H, f) != H) {
        puts("Short .qm body");
        fclose(f); qm2o_dict_free(d); return 1;
    }
    fclose(f);
    return 0;
}

void qm2o_dict_free(qm2o_dict *d)
{
    free(d->keys); free(d->chain);
    d->keys = NULL; d->chain = NULL;
}

/* ---- input line discipline ------------------------------------------------ */

/* :393-398, :451-455.  The reference reads with fgets(line, 100000): a line of
 * >= 99 999 characters, or a last line with no '\n', makes its loop at :403 run
 * past the terminator (undefined behaviour).  Here every emitted line is
 * '\n'-terminated and lines are unbounded; that divergence is documented. */
char *qm2o_strip_reads(const char *path, int seekable, size_t *n_out)
{
    FILE *f = fopen(path, "rb");
    if (!f) { puts("Input open fail"); return NULL; }    /* :339-341 */
    size_t cap = 1 << 20, len = 0;
    char *out = malloc(cap);
    char *line = NULL;
    size_t lcap = 0;
    ssize_t got;
    int fastq = 0;
    got = getline(&line, &lcap, f);                      /* :393 */
    if (got > 0 && line[0] == '@') fastq = 1;            /* :395 */
    else if (seekable) rewind(f);                        /* :396 */
    while ((got = getline(&line, &lcap, f)) > 0) {       /* :397 */
        if (line[0] == '>') continue;                    /* :398 */
        size_t body = (size_t)got;
        if (line[body - 1] == '\n') body--;
        if (len + body + 1 > cap) {
            while (len + body + 1 > cap) cap *= 2;
            out = realloc(out, cap);
        }
        memcpy(out + len, line, body);
        len += body;
        out[len++] = '\n';
        if (fastq)                                       /* :451-455 */
            for (int skip = 0; skip < 3; skip++)
                if (getline(&line, &lcap, f) <= 0) break;
    }
    free(line);
    fclose(f);
    *n_out = len;
    return out;
}

/* ---- epilogue ------------------------------------------------------------- */

uint64_t qm2o_epilogue(const qm2o_dict *d, const uint16_t *depth,
                       const char *qgc_path, const char *out_prefix,
                       double *mean_depth)
{
    char path[4096];
    snprintf(path, sizeof path, "%s.bin", out_prefix);
    FILE *bin = fopen(path, "wb");
    if (!bin) { printf("Cannot open %s\n", path); return (uint64_t)-1; }
    FILE *qgc = qgc_path ? fopen(qgc_path, "rb") : NULL; /* :486-488 */

    uint16_t *buf = malloc(QM2O_BIN_CHUNK * sizeof(uint16_t));
    uint16_t *gc = malloc(QM2O_BIN_CHUNK * sizeof(uint16_t));
    double sum[401] = {0}, sumsq[401] = {0};             /* :495-496 */
    uint32_t cnt[401] = {0};                             /* :497 */
    uint32_t fill = 0;
    uint64_t total = 0;
    const uint32_t first = (uint32_t)d->first_idx;       /* :494 */
    uint32_t at = first;
    do {                                                 /* :498-516 */
        buf[fill] = depth[at];
        at = d->chain[at];
        if (qgc) {
            if (fill == 0) {                             /* :502-503 */
                size_t r = fread(gc, 2, QM2O_BIN_CHUNK, qgc);
                (void)r;
            }
            uint16_t g = gc[fill];
            if (g & 0x8000) {                            /* :504-508 */
                uint32_t b = g & 0x1FF;
                if (b < 401) {
                    sum[b] += buf[fill];
                    cnt[b]++;
                    sumsq[b] += (int)buf[fill] * (int)buf[fill];
                }
            }
        }
        fill++; total++;
        if (fill == QM2O_BIN_CHUNK) {                    /* :511-514 */
            fwrite(buf, 2, fill, bin);
            fill = 0;
        }
    } while (at != first);
    fwrite(buf, 2, fill, bin);                           /* :517 */
    fclose(bin);
    free(buf); free(gc);

    if (mean_depth) *mean_depth = NAN;
    if (qgc) {                                           /* :522-542 */
        fclose(qgc);
        snprintf(path, sizeof path, "%s.txt", out_prefix);
        FILE *txt = fopen(path, "w");
        if (!txt) { printf("Cannot open %s\n", path); return (uint64_t)-1; }
        double tot_depth = 0;
        uint64_t tot_cnt = 0;
        for (int i = 0; i < 401; i++) {
            tot_cnt += cnt[i];
            tot_depth += sum[i];
            double mean = sum[i], var = sumsq[i];
            if (cnt[i]) {
                mean = sum[i] / cnt[i];
                var = sumsq[i] / cnt[i] - mean * mean;
            }
            fprintf(txt, "%.2f\t%f\t%i\t%f\n", i / 4.0, mean, (int)cnt[i], var);
        }
        fclose(txt);
        if (mean_depth) *mean_depth = tot_depth / (double)tot_cnt;
    }
    return total;
}

/* ---- whole command -------------------------------------------------------- */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int qm2o_count_main(const char *ref_prefix, const char *reads_path,
                    const char *out_prefix, int nthreads, qm2o_stats *st,
                    double *count_seconds)
{
    char path[4096];
    qm2o_dict d;
    snprintf(path, sizeof path, "%s.qm", ref_prefix);    /* :335-337 */
    if (qm2o_dict_load(path, &d)) return 1;
    uint16_t *depth = calloc(d.hash_size, sizeof(uint16_t)); /* :360 */
    size_t n = 0;
    char *lines = qm2o_strip_reads(reads_path, 1, &n);
    if (!depth || !lines) {
        puts("Memory allocation failed");
        free(depth); free(lines); qm2o_dict_free(&d);
        return 1;
    }
    qm2o_stats local = {0, 0, 0, 0};
    double t0 = now_s();
    int rc = qm2o_count_lines_mt(d.keys, d.hash_size, d.k, lines, n, depth,
                                 nthreads, &local);
    double t1 = now_s();
    free(lines);
    if (count_seconds) *count_seconds = t1 - t0;
    if (st) *st = local;
    if (rc == 0) {
        snprintf(path, sizeof path, "%s.qgc", ref_prefix);/* :484-486 */
        double mean;
        if (qm2o_epilogue(&d, depth, path, out_prefix, &mean) == (uint64_t)-1)
            rc = 1;
    }
    free(depth);
    qm2o_dict_free(&d);
    return rc;
}
