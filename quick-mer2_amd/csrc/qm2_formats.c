/*
 * qm2_formats.c — .qm reader and .bin / .txt writers of `quicKmer2 count`.
 * Formats per SURVEY Appendix A; behaviour per QuicKmer.c:334-366 and :483-545.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "qm2_host.h"

#ifndef QM2_PIECE_ENTRIES                /* (tests build small pieces: many of them on a small dictionary) */
#define QM2_PIECE_ENTRIES (1u << 22)    /* (the reference flushes .bin every 2^20 entries, :11, :511; here 8 MiB pieces, in parallel) */
#endif
#define QM2_OUT_THREADS 12

/* The loop of :498-516 over the counts in chain order, split in two.  Depth to .bin: pieces that worker threads take
 * in turn (fetch from the device, pwrite).  The sums of the GC curve (:504-509): on the device, where the counts
 * are (qm2_chain_gc_sums_fd: the .qgc goes up like the chain did, 4.5 GB at hg38 scale), while the workers write.
 * (Round 2 did the sums in the workers: 5.2 s of thread time of the 6.3 s the whole loop took at hg38 scale, 1.2 - 1.4 s
 * on 12 threads; the serial loop of the reference's shape: 7 s.)  The sums are exact 64-bit integers, converted at the
 * end: the reference adds into doubles, which is the same number as long as every partial sum is below 2^53 (SURVEY
 * Appendix B.9: depths would have to average > 4 x 10^6 over a GC bin for that to fail). */
#define QM2_GC_BLOCK (1u << 20)          /* entries the reference freads from the .qgc at a time (buffer_size, :11, :502-503) */
enum { QM2_OUT_OK = 0, QM2_OUT_ALLOC, QM2_OUT_FETCH, QM2_OUT_QGC, QM2_OUT_WRITE };

typedef struct {
    qm2_ctx   *ctx;
    int        bin_fd;
    uint64_t   n_entries;
    uint64_t   next;                            /* next piece to hand out (under `lock`) */
    int        failed;                          /* QM2_OUT_*: what went wrong first */
    pthread_mutex_t lock;
    double     t_fetch, t_write;                /* seconds, summed over the workers (QM2_TIMING) */
} out_job;

static double out_now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int pread_all(int fd, void *buf, size_t n, uint64_t off, size_t *got_out)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t r = pread(fd, (char *)buf + got, n - got, (off_t)(off + got));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return 1;
        if (r == 0) break;
        got += (size_t)r;
    }
    *got_out = got;
    return 0;
}

/* A .qgc shorter than the dictionary.  The reference freads 2^20 entries into gc_control[] whenever its buffer index
 * is 0 and never looks at what fread returned (:502-503): behind the end of the file the array simply keeps what the
 * earlier reads left in it.  With n entries in the file, F = n / 2^20 full blocks and r = n % 2^20: entry e >= n reads
 * position p = e % 2^20 of that array, which holds file entry F * 2^20 + p if p < r (the last, partial read), else
 * file entry (F - 1) * 2^20 + p (the last full block).  With F = 0 the rest of the array is the reference's
 * uninitialised stack: taken as zeros here (the one deliberate difference, on an input the reference leaves undefined).
 * -> the 2^20 values, or NULL (*bad says why). */
static uint16_t *stale_block(int qgc_fd, uint64_t n_qgc, int *bad)
{
    const uint64_t B = QM2_GC_BLOCK, F = n_qgc / B, r = n_qgc % B;
    uint16_t *stale = calloc(B, sizeof(uint16_t));
    if (!stale) { *bad = QM2_OUT_ALLOC; return NULL; }
    size_t got = 0;
    if ((F && pread_all(qgc_fd, stale, B * sizeof(uint16_t), (F - 1) * B * sizeof(uint16_t), &got)) ||
        (r && pread_all(qgc_fd, stale, r * sizeof(uint16_t), F * B * sizeof(uint16_t), &got))) {
        free(stale);
        *bad = QM2_OUT_QGC;
        return NULL;
    }
    return stale;
}

static void *out_worker(void *arg)
{
    out_job *j = arg;
    uint16_t *out = malloc(QM2_PIECE_ENTRIES * sizeof(uint16_t));
    int bad = !out ? QM2_OUT_ALLOC : QM2_OUT_OK;
    double t_fetch = 0, t_write = 0;
    while (!bad) {
        pthread_mutex_lock(&j->lock);
        const uint64_t at = j->next;
        const int stop = j->failed || at >= j->n_entries;
        if (!stop) j->next = at + QM2_PIECE_ENTRIES;
        pthread_mutex_unlock(&j->lock);
        if (stop) break;
        const uint64_t m = j->n_entries - at < QM2_PIECE_ENTRIES ? j->n_entries - at : QM2_PIECE_ENTRIES;
        const double t0 = out_now();
        if (qm2_chain_fetch(j->ctx, out, at, m)) { bad = QM2_OUT_FETCH; break; }
        const double t1 = out_now();
        t_fetch += t1 - t0;
        size_t put = 0;                                                        /* :512, :517 */
        while (put < m * sizeof(uint16_t)) {
            const ssize_t n = pwrite(j->bin_fd, (const char *)out + put, m * sizeof(uint16_t) - put, (off_t)(at * sizeof(uint16_t) + put));
            if (n < 0 && errno == EINTR) continue;
            if (n <= 0) { bad = QM2_OUT_WRITE; break; }
            put += (size_t)n;
        }
        t_write += out_now() - t1;
    }
    pthread_mutex_lock(&j->lock);
    j->t_fetch += t_fetch; j->t_write += t_write;
    if (bad && !j->failed) j->failed = bad;
    pthread_mutex_unlock(&j->lock);
    free(out);
    return NULL;
}

int qm2_dict_open(const char *ref_prefix, qm2_dict_file *d)
{
    memset(d, 0, sizeof *d);
    size_t len = strlen(ref_prefix);
    char *path = malloc(len + 8);
    if (!path) { puts("Memory allocation failed"); return 1; }
    memcpy(path, ref_prefix, len);
    memcpy(path + len, ".qm", 4);
    d->fp = fopen(path, "rb");                               /* :337 */
    if (!d->fp) {
        /* the reference does not check this fopen and crashes; say why instead */
        printf("Cannot open dictionary %s\n", path);
        free(path);
        return 1;
    }
    free(path);
    unsigned char hdr[24];
    if (fread(hdr, 1, sizeof hdr, d->fp) != sizeof hdr) {
        puts("Dictionary file is too short");
        qm2_dict_close(d);
        return 1;
    }
    d->k = hdr[4];                                           /* :345-347 */
    memcpy(&d->hash_size, hdr + 8, 8);                       /* :348-349 */
    memcpy(&d->first_idx, hdr + 16, 8);                      /* :350-351 */
    printf("Hash Size: 0x%lX\nFirst location: 0x%lX\n",      /* :352 */
           (unsigned long)d->hash_size, (unsigned long)d->first_idx);
    const uint64_t H = d->hash_size;
    if (H < 2 || (H & (H - 1)) || H > ((uint64_t)1 << 32)) {
        puts("Dictionary header holds an impossible hash size");
        qm2_dict_close(d);
        return 1;
    }
    return 0;
}

int qm2_dict_load_keys_quiet(qm2_dict_file *d, qm2_ctx *ctx, int print)
{
    /* :353-359 without the host array: file -> pinned pieces -> device */
    if (qm2_keys_from_fd(ctx, fileno(d->fp), 24)) return 1;                      /* says "... truncated" itself */
    if (print) printf("Read 0x%lX hash\n", (unsigned long)d->hash_size);         /* :359 */
    return 0;
}

int qm2_dict_load_keys(qm2_dict_file *d, qm2_ctx *ctx) { return qm2_dict_load_keys_quiet(d, ctx, 1); }

void qm2_dict_close(qm2_dict_file *d)
{
    if (d->fp) fclose(d->fp);
    d->fp = NULL;
}

int qm2_write_outputs(qm2_dict_file *d, qm2_ctx *ctx, const char *ref_prefix, const char *out_prefix)
{
    const uint64_t H = d->hash_size;
    /* :483 prints how many chain entries the file gave; the read itself happens inside
     * qm2_chain_gather_fd, so the count comes from the file size */
    uint64_t got = H;
    struct stat sb;
    if (fstat(fileno(d->fp), &sb) == 0 && S_ISREG(sb.st_mode)) {
        const uint64_t chain_at = 24 + H * sizeof(uint64_t);
        const uint64_t have = (uint64_t)sb.st_size > chain_at ? ((uint64_t)sb.st_size - chain_at) / sizeof(uint32_t) : 0;
        if (have < got) got = have;
    }
    printf("Pileup finish\nRead chain file %lu entries\n", (unsigned long)got);   /* :483 */
    if (got != H) { puts("Dictionary chain is truncated"); return 1; }

    size_t plen = strlen(ref_prefix), olen = strlen(out_prefix);
    char *path = malloc((plen > olen ? plen : olen) + 8);
    if (!path) { puts("Memory allocation failed"); return 1; }

    memcpy(path, ref_prefix, plen);
    memcpy(path + plen, ".qgc", 5);                          /* :484-486 */
    FILE *qgc = fopen(path, "rb");
    if (!qgc) printf("GC control file %s absent. Continue without GC correction!\n", path); /* :488 */

    /* the chain goes from the file to where the dead key table was on the device (:483, :490);
     * the walk of :494-516 is a parallel list ranking there */
    uint64_t n_entries = 0;
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (qm2_chain_gather_fd(ctx, fileno(d->fp), 24 + H * sizeof(uint64_t), d->first_idx, &n_entries)) {
        if (qgc) fclose(qgc);
        free(path);
        return 1;
    }

    clock_gettime(CLOCK_MONOTONIC, &t1);
    memcpy(path, out_prefix, olen);
    memcpy(path + olen, ".bin", 5);                          /* :342-344 */
    FILE *bin = fopen(path, "wb");
    if (!bin) {
        printf("Cannot create %s\n", path);
        if (qgc) fclose(qgc);
        free(path);
        return 1;
    }

    double sum[401], sumsq[401];
    uint32_t cnt[401];
    memset(sum, 0, sizeof sum); memset(sumsq, 0, sizeof sumsq); memset(cnt, 0, sizeof cnt);
    int rc = 0, n_th = 0;
    double t_sums = 0;
    {
        out_job job;
        memset(&job, 0, sizeof job);
        job.ctx = ctx;
        job.bin_fd = fileno(bin);
        job.n_entries = n_entries;
        /* (4.5 GB at hg38 scale: 1.2 s, the rate at which ONE file takes buffered writes — twelve threads calling pwrite
         * take turns at its inode lock; mapping the file and fetching straight into the mapping was measured the same,
         * 1.12 s: the page cache of one file fills at ~4 GB/s either way) */
        pthread_mutex_init(&job.lock, NULL);
        pthread_t th[QM2_OUT_THREADS];
        const uint64_t pieces = (n_entries + QM2_PIECE_ENTRIES - 1) / QM2_PIECE_ENTRIES;
        while (n_th < QM2_OUT_THREADS && (uint64_t)n_th < pieces && pthread_create(&th[n_th], NULL, out_worker, &job) == 0) n_th++;
        /* the GC sums meanwhile, on the device */
        int gc_bad = QM2_OUT_OK;
        if (qgc) {
            const double t_a = out_now();
            const int qgc_fd = fileno(qgc);
            uint64_t n_qgc = n_entries;
            uint16_t *stale = NULL;
            struct stat gsb;
            if (fstat(qgc_fd, &gsb) == 0 && S_ISREG(gsb.st_mode) && (uint64_t)gsb.st_size / sizeof(uint16_t) < n_entries) {
                n_qgc = (uint64_t)gsb.st_size / sizeof(uint16_t);
                stale = stale_block(qgc_fd, n_qgc, &gc_bad);
            }
            if (!gc_bad) {
                int64_t isum[401], isq[401];
                uint64_t icnt[401];
                /* (a failure has printed its reason itself: qm2_last_error) */
                if (qm2_chain_gc_sums_fd(ctx, qgc_fd, n_qgc, stale, isum, isq, icnt)) gc_bad = QM2_OUT_QGC;
                else for (int b = 0; b < 401; b++) { sum[b] = (double)isum[b]; sumsq[b] = (double)isq[b]; cnt[b] = (uint32_t)icnt[b]; }
            }
            free(stale);
            t_sums = out_now() - t_a;
        }
        if (n_th == 0) out_worker(&job);                                        /* no thread to be had: this one */
        for (int t = 0; t < n_th; t++) pthread_join(th[t], NULL);
        pthread_mutex_destroy(&job.lock);
        if (!job.failed) job.failed = gc_bad;
        if (job.failed) {
            /* (a failed qm2_chain_fetch has printed the HIP error itself: qm2_last_error) */
            puts(job.failed == QM2_OUT_ALLOC ? "Memory allocation failed" : job.failed == QM2_OUT_FETCH ? "Fetching the counts from the device failed"
                 : job.failed == QM2_OUT_QGC ? "Reading the GC control file failed" : "Write to .bin failed");
            rc = 1;
        }
        if (getenv("QM2_TIMING"))
            printf("QM2 epilogue workers (seconds summed over %d threads): fetch %.3f, .bin write %.3f; GC sums on the device beside them %.3f s\n",
                   n_th ? n_th : 1, job.t_fetch, job.t_write, t_sums);
    }
    if (fclose(bin) != 0 && !rc) { puts("Write to .bin failed"); rc = 1; }
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (getenv("QM2_TIMING"))
        printf("QM2 epilogue: chain file -> device + list ranking %.3f s, .bin%s on %d threads %.3f s\n",
               (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), qgc ? " (+ GC sums on the device)" : "", n_th ? n_th : 1,
               (t2.tv_sec - t1.tv_sec) + 1e-9 * (t2.tv_nsec - t1.tv_nsec));
    if (rc) { if (qgc) fclose(qgc); free(path); return 1; }

    if (qgc) {                                               /* :522-542 */
        fclose(qgc);
        memcpy(path + olen, ".txt", 5);
        FILE *txt = fopen(path, "w");
        if (!txt) { printf("Cannot create %s\n", path); free(path); return 1; }
        double all_sum = 0;
        uint64_t all_cnt = 0;
        for (int i = 0; i < 401; i++) {
            all_cnt += cnt[i];
            all_sum += sum[i];
            double mean = sum[i], var = sumsq[i];
            if (cnt[i]) {
                mean = sum[i] / cnt[i];
                var = sumsq[i] / cnt[i] - mean * mean;
            }
            fprintf(txt, "%.2f\t%f\t%i\t%f\n", i / 4.0, mean, (int)cnt[i], var);   /* :536 */
        }
        fclose(txt);
        printf("Mean sequencing depth: %.2f\n", all_sum / (double)all_cnt);        /* :540 */
    }
    free(path);
    return 0;
}
