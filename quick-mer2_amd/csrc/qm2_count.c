/*
 * qm2_count.c — `quicKmer2 count [-h] [-t N] ref.fa sample.fast[a/q] Out_prefix`
 * with the counting phase on the MI355X behind include/qm2.h.
 * Follows main_count, QuicKmer.c:304-545: same positional arguments taken from
 * the end (:335, :338, :342), same stdout lines, same output files.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <getopt.h>
#include <libgen.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "../../include/qm2_shards.h"
#include "qm2_dispatch.h"
#include "qm2_host.h"

#define QM2_MAX_DEVICES 16

static void help_count(void)
{
    puts("\nquicKmer2 count [Options] ref.fa sample.fast[a/q] Out_prefix\n\nOptions:");   /* :299 */
    puts("-h\t\tShow this help information");
    puts("-t [num]\tNumber of threads");
    puts("--devices a,b,..\tGPUs to count on (a replica of the dictionary on each, the reads dealt to them a device");
    puts("\t\tbatch at a time, the counts summed on the first); default: the current device");
    puts("--shards a,b,..\tGPUs (a power of two of them) that each hold a hash range of the dictionary; the k-mers");
    puts("\t\tof every batch are routed to their owners by RCCL (needs libqm2_shards.so beside this binary).");
    puts("\t\tEXPERIMENTAL on more than one GPU (not yet run on two physical GPUs): set QM2_SHARDS_EXPERIMENTAL=1");
}

/* The reference prints `Read %liG kmers` whenever its k-mer count reaches a multiple of 2^30 (:446).
 * Here the count advances a device batch at a time: every multiple crossed since the last look is
 * printed (same lines, same order, in front of "Counting elapse"), from the backend's non-blocking
 * snapshot while the stream runs and from the final totals at the end. */
static uint64_t progress_printed_g;
static void progress_upto(uint64_t n_kmers)
{
    while (progress_printed_g < (n_kmers >> 30)) printf("Read %liG kmers\n", (long)++progress_printed_g);
}
static qm2_ctx *progress_ctx[QM2_MAX_DEVICES];
static int progress_n;
static void progress_peek(qm2_ctx *ctx)
{
    (void)ctx;
    uint64_t total = 0;
    for (int i = 0; i < progress_n; i++) {
        qm2_stats st;
        if (qm2_stats_peek(progress_ctx[i], &st) == 0) total += st.n_kmers;
    }
    progress_upto(total);
}
static int out_flush(void *user) { return qm2_flush((qm2_ctx *)user); }

static int out_acquire(void *user, char **buf, size_t *cap) { return qm2_staging_acquire((qm2_ctx *)user, buf, cap); }
static int out_commit(void *user, char *buf, size_t n)
{
    progress_peek((qm2_ctx *)user);
    return qm2_staging_commit((qm2_ctx *)user, buf, n);
}
static int out_submit(void *user, const char *lines, size_t n)
{
    progress_peek((qm2_ctx *)user);
    return qm2_submit((qm2_ctx *)user, lines, n);
}
static int out_commit_packed(void *user, char *buf, const qm2_packed *p, const uint64_t *wraps)
{
    progress_peek((qm2_ctx *)user);
    return qm2_staging_commit_packed((qm2_ctx *)user, buf, p, wraps);
}
static int out_commit_raw(void *user, char *buf, size_t offset, size_t n, int fastq, unsigned skip)
{
    progress_peek((qm2_ctx *)user);
    if (qm2_staging_commit_raw((qm2_ctx *)user, buf, offset, n, (uint32_t)fastq, skip)) return 1;
    return qm2_raw_verdict((qm2_ctx *)user) ? 2 : 0;       /* refused: no point in reading on */
}

/* The CLI's device batch: 1.5 bytes per table slot, 1..2 GiB.  (The library's own default goes up to 6 GiB, which is what
 * the kernels like best: 95 instead of 50-75 G bases/s at hg38 scale.  But the CLI is fed over PCIe at 23 G bases/s of FASTQ or
 * 45 of FASTA, which 2 GiB batches still outrun, and every byte of batch costs ~11-20 bytes of partition scratch:
 * with 6 GiB batches a run takes 170 GB of device memory and the driver stalls 3-4 s in one of the allocations in four
 * runs out of five (load phase 2.1-4.9 s); with 2 GiB it takes 125 GB and did not on two boxes (1.0-1.2 s; 1.1-2.5 s on a third) — whole run at hg38 scale,
 * 100 M reads: 6.5-8.1 s -> 3.5-4.1 s.  profiles/r04_c3/cli_batch_bytes.txt) */
static uint64_t default_batch_bytes(uint64_t hash_size)
{
    uint64_t bb = hash_size + hash_size / 2;
    if (bb < (1ull << 30)) bb = 1ull << 30;
    if (bb > (2ull << 30)) bb = 2ull << 30;
    return bb;
}

/* How the input reaches the device (QM2_READER = packed | raw | lines; the older QM2_READER_RAW = 1 / 0 = raw / lines):
 *   packed (default)  the reader threads run the line discipline and pack what it keeps — 2-bit codes and reset flags,
 *                     0.375 bytes per base over the link (include/qm2.h, "packed sequence"); files and pipes alike
 *   raw               regular files: the threads only pread, the device does the line discipline (2.05 bytes per base
 *                     of FASTQ, 1.0x of FASTA); pipes: lines
 *   lines             the threads strip to ASCII sequence lines (1 byte per base) */
enum { READER_PACKED = 0, READER_RAW = 1, READER_LINES = 2 };
static int reader_mode(void)
{
    const char *m = getenv("QM2_READER"), *raw = getenv("QM2_READER_RAW");
    if (m && !strcmp(m, "raw")) return READER_RAW;
    if (m && !strcmp(m, "lines")) return READER_LINES;
    if (m && !strcmp(m, "packed")) return READER_PACKED;
    if (raw) return strcmp(raw, "0") == 0 ? READER_LINES : READER_RAW;
    return READER_PACKED;
}

/* What the reader will do with `reads`: a regular file goes to the device as raw text (unless QM2_READER_RAW=0) and
 * its first byte says FASTQ or FASTA (:393-395); *size = its length, 0 for pipes. */
static void sniff_input(const char *reads, int *regular, int *fastq, uint64_t *size)
{
    *regular = 0; *fastq = 0; *size = 0;
    struct stat sb;
    if (stat(reads, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= 0) return;
    *regular = 1;
    *size = (uint64_t)sb.st_size;
    const int fd = open(reads, O_RDONLY);
    if (fd < 0) return;
    char first = 0;
    if (pread(fd, &first, 1, 0) == 1 && first == '@') *fastq = 1;
    close(fd);
}

/* k-mers per 1024 bytes of sequence lines, from the head of a regular input file (qm2_opts.kmers_per_kib): a line of
 * L bases and its newline yield at most L - k + 1 k-mers, so 150-base reads fill 0.8 of what their bytes would allow
 * and the partition buffers can be that much smaller.  An estimate, 5 % on top; reads that turn out longer further
 * down only cost speed (spill list / recount), never counts.  0 = no hint. */
static uint32_t estimate_kmers_per_kib(const char *reads, uint32_t k, int fastq)
{
    const int fd = open(reads, O_RDONLY);
    if (fd < 0) return 0;
    static char head[1 << 20];
    const ssize_t got = pread(fd, head, sizeof head, 0);
    close(fd);
    if (got <= 0) return 0;
    uint64_t bytes = 0, kmers = 0, line_no = 0;
    ssize_t at = 0;
    while (at < got) {
        const char *nl = memchr(head + at, '\n', (size_t)(got - at));
        if (!nl) break;                                        /* an unfinished line: not counted */
        const uint64_t len = (uint64_t)(nl - (head + at));
        const int is_seq = fastq ? (line_no & 3u) == 1u : head[at] != '>';
        if (is_seq) {
            bytes += len + 1;
            if (len >= k) kmers += len - k + 1;
        }
        line_no++;
        at += (ssize_t)len + 1;
    }
    if (bytes < 4096) return 0;                                /* (one huge line, or hardly any text: no hint) */
    const double per_kib = 1024.0 * (double)kmers / (double)bytes * 1.05 + 8.0;
    return per_kib >= 1024.0 ? 0u : (uint32_t)per_kib;
}

/* The partition buffers (tens of GB of hipMalloc: seconds of driver time at hg38 scale) are allocated on a thread of
 * their own while the calling thread streams the dictionary from the file to the device. */
typedef struct { qm2_ctx *ctx; uint32_t what; int rc; } prepare_job;
static void *prepare_main(void *arg)
{
    prepare_job *j = arg;
    j->rc = qm2_prepare(j->ctx, 0, j->what);
    return NULL;
}

/* qm2_prepare's partition part beside qm2_keys_from_fd, then the look at the key table's runs of occupied slots */
static int load_keys_and_prepare(qm2_dict_file *d, qm2_ctx *ctx, uint32_t what, int print)
{
    /* the thread makes the partition buffers and nothing else: the staging buffers and device batches (made before, by
     * the QM2_PREPARE_NO_PART call) are what qm2_keys_from_fd streams the dictionary through at this very moment */
    prepare_job job = {ctx, what & QM2_PREPARE_FASTQ, 0};
    pthread_t th;
    const int threaded = pthread_create(&th, NULL, prepare_main, &job) == 0;
    int rc = qm2_dict_load_keys_quiet(d, ctx, print);
    if (threaded) pthread_join(th, NULL);
    else job.rc = qm2_prepare(ctx, 0, job.what);
    if (rc || job.rc) return 1;
    return qm2_prepare(ctx, 0, what);          /* (everything is there: only the one-off look at the runs is left) */
}

typedef struct {
    qm2_dict_file *d;
    int device, n_strip;
    uint32_t kmers_per_kib, what;
    int print;
    qm2_ctx *ctx;
    int rc;
} device_load;

static void *device_load_main(void *arg)
{
    device_load *j = arg;
    qm2_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.device = j->device;
    o.flags = QM2_FLAG_KEYS_LATER | QM2_FLAG_RAW_STRICT;
    o.n_staging = (uint32_t)j->n_strip + 3;
    o.kmers_per_kib = j->kmers_per_kib;
    o.batch_bytes = default_batch_bytes(j->d->hash_size);
    if (getenv("QM2_BATCH_BYTES")) o.batch_bytes = strtoull(getenv("QM2_BATCH_BYTES"), NULL, 0);   /* (experiments) */
    j->rc = 1;
    if (qm2_open(&j->ctx, NULL, j->d->hash_size, j->d->k, &o)) return NULL;
    if (qm2_prepare(j->ctx, 0, j->what | QM2_PREPARE_NO_PART)) return NULL;
    if (load_keys_and_prepare(j->d, j->ctx, j->what, j->print)) return NULL;
    j->rc = 0;
    return NULL;
}

/* --devices a,b,...: a replica of the dictionary on every listed GPU (the same ordinal may be listed
 * twice: two contexts on one GPU), one reader, the input dealt to the contexts a device batch at a time
 * (qm2_dispatch.c), the counters summed into the first context before the epilogue.  Regular files go to the
 * devices as raw text, like in the single-context path: a turn is a run of consecutive blocks with the line
 * state at its first byte, so every batch of every context is consecutive text and the device line discipline
 * applies unchanged.  The one thing a context cannot know is what the batches it did NOT see did to the line
 * numbering — a FASTQ sequence-position line that starts with '>' (:398) moves all later sequence lines on by
 * one — so the contexts refuse any such line (QM2_FLAG_RAW_STRICT) and the whole input then goes through the
 * host line discipline instead, from zeroed counters: exact either way.  On return ctxs[0] holds the job's
 * counters and *st its totals. */
static int count_on_devices(qm2_dict_file *d, const char *reads, int n_strip, const int *devices, int n_dev,
                            qm2_ctx **ctxs, qm2_stats *st, int timing)
{
    qm2_sink sinks[QM2_MAX_DEVICES];
    struct timespec ta, tb;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    int regular, fastq;
    uint64_t in_size;
    sniff_input(reads, &regular, &fastq, &in_size);
    int want_raw = regular && n_strip > 0 && reader_mode() == READER_RAW;
    int want_packed = n_strip > 0 && reader_mode() == READER_PACKED;
    const uint32_t what = QM2_PREPARE_HOST | (want_raw ? QM2_PREPARE_RAW : 0u) | (want_raw && fastq ? QM2_PREPARE_FASTQ : 0u);
    /* one loader thread per replica: each reads the .qm through the page cache and feeds its own GPU over its own
     * PCIe link, so N replicas load in about the time of one (the contexts share nothing but the file descriptor,
     * read with pread) */
    device_load jobs[QM2_MAX_DEVICES];
    pthread_t th[QM2_MAX_DEVICES];
    int started[QM2_MAX_DEVICES];
    const uint32_t kpk = regular ? estimate_kmers_per_kib(reads, d->k, fastq) : 0;
    int distinct = 1;                                          /* (two contexts on ONE GPU load one after the other: the
                                                                * first takes the partition buffers it needs, the second
                                                                * what is left, rather than both failing half-way) */
    for (int i = 0; i < n_dev; i++)
        for (int j = 0; j < i; j++) distinct &= devices[i] != devices[j];
    const char *lt = getenv("QM2_LOAD_THREADS");               /* (tests: the threaded loaders on a one-GPU box) */
    if (lt) distinct = strcmp(lt, "0") != 0;
    for (int i = 0; i < n_dev; i++) {
        jobs[i] = (device_load){d, devices[i], n_strip, kpk, what, i == 0, NULL, 0};
        started[i] = distinct && i > 0 && pthread_create(&th[i], NULL, device_load_main, &jobs[i]) == 0;
    }
    for (int i = 0; i < n_dev; i++) {
        if (started[i]) pthread_join(th[i], NULL);
        else device_load_main(&jobs[i]);                       /* (the first one, and any whose thread did not start) */
    }
    int load_rc = 0;
    for (int i = 0; i < n_dev; i++) {
        ctxs[i] = jobs[i].ctx;                                 /* (the caller closes what was opened, failed or not) */
        load_rc |= jobs[i].rc;
    }
    if (load_rc) return 1;
    for (int i = 0; i < n_dev; i++) {
        sinks[i].acquire = out_acquire; sinks[i].commit = out_commit; sinks[i].submit = out_submit;
        sinks[i].flush = out_flush; sinks[i].user = ctxs[i]; sinks[i].commit_raw = out_commit_raw; sinks[i].commit_packed = out_commit_packed;
        progress_ctx[i] = ctxs[i];
    }
    progress_n = n_dev;
    clock_gettime(CLOCK_MONOTONIC, &tb);
    if (timing) printf("QM2 phase load-dictionary %.3f s (file -> %d devices)\n", (tb.tv_sec - ta.tv_sec) + 1e-9 * (tb.tv_nsec - ta.tv_nsec), n_dev);
    /* a turn = one device batch, a little less: the flush that ends the turn then launches the whole batch (a turn
     * of exactly a batch is flushed by the context itself when it fills, and the turn's flush would launch the few
     * bytes that were left over as a batch of their own) */
    const uint64_t bb = default_batch_bytes(d->hash_size);
    const char *env = getenv("QM2_TURN_BYTES");               /* (tests: small turns on small inputs) */
    const size_t turn = env ? (size_t)strtoull(env, NULL, 10) : (size_t)(bb - bb / 16);
    int rc = 0;
    for (int pass = 0; pass < 2; pass++) {
        qm2_dispatch *disp = qm2_dispatch_new(sinks, n_dev, turn);
        if (!disp) { puts("Memory allocation failed"); return 1; }
        if (want_raw) {
            /* raw blocks: whole blocks of the reader, as many as fit a batch with room for the line that ends each */
            const size_t per_block = qm2_reader_block_bytes() + (1u << 20);
            const char *tb_env = getenv("QM2_TURN_BLOCKS");
            size_t blocks = tb_env ? (size_t)strtoull(tb_env, NULL, 10) : (size_t)(bb / per_block);
            qm2_dispatch_raw_turn(disp, blocks ? blocks : 1);
        }
        qm2_lines_out out = {qm2_dispatch_acquire, qm2_dispatch_commit, qm2_dispatch_submit, disp, want_raw ? qm2_dispatch_commit_raw : NULL,
                             want_packed && !want_raw ? qm2_dispatch_commit_packed : NULL};
        rc = qm2_stream_reads(reads, &out, n_strip);
        qm2_dispatch_free(disp);
        memset(st, 0, sizeof *st);
        uint32_t verdict = 0;
        for (int i = 0; i < n_dev && rc != 1; i++) {
            qm2_stats s1;
            if (qm2_finish(ctxs[i], NULL, &s1)) { rc = 1; break; }
            verdict |= qm2_raw_verdict(ctxs[i]);
            st->n_bases += s1.n_bases; st->n_kmers += s1.n_kmers; st->n_probes += s1.n_probes; st->n_hits += s1.n_hits;
        }
        if (rc == 3 && want_packed) {                          /* ('>' at a FASTQ sequence position: qm2_host.h) */
            if (timing) printf("QM2 packed input given up ('>' at a FASTQ sequence position): host line discipline in order\n");
            want_packed = 0;
            rc = 0;
            for (int i = 0; i < n_dev && !rc; i++) rc = qm2_reset(ctxs[i]);
            if (rc) break;
            pass = 0;                                          /* (one more go) */
            continue;
        }
        if (rc == 1 || !want_raw || (rc == 0 && verdict == 0)) break;
        /* a raw block was refused (or did not fit a buffer): everything again through the host line discipline */
        if (timing) printf("QM2 raw input refused (verdict %u): host line discipline\n", verdict);
        want_raw = 0;
        rc = 0;
        for (int i = 0; i < n_dev && !rc; i++) rc = qm2_reset(ctxs[i]);
        if (rc) break;
    }
    struct timespec tc;
    clock_gettime(CLOCK_MONOTONIC, &tc);
    if (timing && !rc) {
        const double sec = (tc.tv_sec - tb.tv_sec) + 1e-9 * (tc.tv_nsec - tb.tv_nsec);
        printf("QM2 phase count %.3f s (%.2f G bases/s) on %d contexts, %s\n", sec, st->n_bases / sec / 1e9, n_dev,
               want_raw ? "raw blocks in turns" : want_packed ? "packed on the host" : "host line discipline");
    }
    for (int i = 1; i < n_dev && !rc; i++) rc = qm2_merge_counters(ctxs[0], ctxs[i]);
    if (timing && !rc) {
        struct timespec tm;
        clock_gettime(CLOCK_MONOTONIC, &tm);
        printf("QM2 phase merge-counters %.3f s\n", (tm.tv_sec - tc.tv_sec) + 1e-9 * (tm.tv_nsec - tc.tv_nsec));
    }
    return rc;
}

/* --shards a,b,...: BASELINE config 4 (include/qm2_shards.h).  The driver lives in libqm2_shards.so, which brings
 * RCCL with it: loaded here, on demand, so that nothing else of the CLI depends on it. */
typedef struct {
    void *lib;
    int (*open)(qm2_shards **, int, uint64_t, uint32_t, const int *, int, uint64_t, uint32_t);
    int (*acquire)(qm2_shards *, char **, size_t *);
    int (*commit)(qm2_shards *, char *, size_t);
    int (*submit)(qm2_shards *, const char *, size_t);
    int (*finish)(qm2_shards *, qm2_stats *);
    qm2_ctx *(*result)(qm2_shards *);
    int (*get_info)(qm2_shards *, qm2_shards_info *);
    void (*close)(qm2_shards *);
} shards_api;

static int shards_load(shards_api *a)
{
    char exe[4096], path[4200];
    ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (n <= 0) { puts("Cannot locate libqm2_shards.so"); return 1; }
    exe[n] = 0;
    snprintf(path, sizeof path, "%s/libqm2_shards.so", dirname(exe));
    a->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!a->lib) { printf("Cannot load %s: %s\n", path, dlerror()); return 1; }
#define SHARDS_SYM(field, name) do { *(void **)&a->field = dlsym(a->lib, name); if (!a->field) { printf("%s lacks %s\n", path, name); return 1; } } while (0)
    SHARDS_SYM(open, "qm2_shards_open"); SHARDS_SYM(acquire, "qm2_shards_acquire"); SHARDS_SYM(commit, "qm2_shards_commit");
    SHARDS_SYM(submit, "qm2_shards_submit"); SHARDS_SYM(finish, "qm2_shards_finish"); SHARDS_SYM(result, "qm2_shards_result");
    SHARDS_SYM(get_info, "qm2_shards_get_info"); SHARDS_SYM(close, "qm2_shards_close");
#undef SHARDS_SYM
    return 0;
}

static shards_api g_shards;
static int sh_acquire(void *u, char **b, size_t *c) { return g_shards.acquire((qm2_shards *)u, b, c); }
static int sh_commit(void *u, char *b, size_t n) { return g_shards.commit((qm2_shards *)u, b, n); }
static int sh_submit(void *u, const char *p, size_t n) { return g_shards.submit((qm2_shards *)u, p, n); }

static int count_on_shards(qm2_dict_file *d, const char *reads, int n_strip, const int *devices, int n_dev,
                           const char *ref_prefix, const char *out_prefix, int timing)
{
    if (shards_load(&g_shards)) return 1;
    struct timespec ta, tb, tc;
    clock_gettime(CLOCK_MONOTONIC, &ta);
    qm2_shards *S = NULL;
    const char *bb_env = getenv("QM2_SHARDS_BATCH_BYTES");            /* (tests: small rounds on small inputs) */
    int rc = g_shards.open(&S, fileno(d->fp), d->hash_size, d->k, devices, n_dev, bb_env ? strtoull(bb_env, NULL, 10) : 0,
                           (uint32_t)n_strip + 3);
    if (!rc) printf("Read 0x%lX hash\n", (unsigned long)d->hash_size);           /* :359 */
    clock_gettime(CLOCK_MONOTONIC, &tb);
    if (timing && !rc) printf("QM2 phase load-dictionary %.3f s (file -> %d hash-range shards)\n", (tb.tv_sec - ta.tv_sec) + 1e-9 * (tb.tv_nsec - ta.tv_nsec), n_dev);
    qm2_stats st;
    memset(&st, 0, sizeof st);
    if (!rc) {
        qm2_lines_out out = {sh_acquire, sh_commit, sh_submit, S, NULL, NULL};       /* sequence lines: the host line discipline */
        rc = qm2_stream_reads(reads, &out, n_strip);
        if (!rc) rc = g_shards.finish(S, &st);
    }
    clock_gettime(CLOCK_MONOTONIC, &tc);
    if (!rc) {
        progress_printed_g = 0;
        progress_upto(st.n_kmers);
        printf("Counting elapse %u sec, total %lu kmers\n", (unsigned)(tc.tv_sec - tb.tv_sec), (unsigned long)st.n_kmers);   /* :481 */
        if (timing) {
            qm2_shards_info info;
            memset(&info, 0, sizeof info);
            (void)g_shards.get_info(S, &info);
            const double sec = (tc.tv_sec - tb.tv_sec) + 1e-9 * (tc.tv_nsec - tb.tv_nsec);
            printf("QM2 phase count %.3f s (%.2f G bases/s) on %d hash-range shards: %llu rounds, %llu cut and routed again, %.2f GB exchanged; %s, %u pieces fell back to one slab per owner\n",
                   sec, st.n_bases / sec / 1e9, n_dev, (unsigned long long)info.rounds, (unsigned long long)info.rerouted, info.exchanged_bytes / 1e9,
                   info.ten_bit_route ? "10-bit route (whole sub-slab blocks, one split behind the wire)" : "one slab per owner (filled parts, two splits behind the wire)",
                   info.flat_fallbacks);
        }
        rc = qm2_write_outputs(d, g_shards.result(S), ref_prefix, out_prefix);
    }
    if (S) g_shards.close(S);
    return rc;
}

int qm2_main_count(int argc, char **argv)
{
    unsigned thread_count = 0;
    int getopt_seen_t = 0;
    if (argc < 2) { help_count(); return 1; }                /* :309-312 */
    int opt;
    optind = 1;
    int devices[QM2_MAX_DEVICES], n_dev = 0;
    int shards = 0;
    static const struct option long_opts[] = {{"devices", required_argument, NULL, 'D'}, {"shards", required_argument, NULL, 'S'}, {NULL, 0, NULL, 0}};
    while ((opt = getopt_long(argc, argv, "ht:", long_opts, NULL)) != -1) {        /* :314 (+ --devices) */
        switch (opt) {
        case 'S': shards = 1; /* fall through */
        case 'D':
            for (char *tok = strtok(optarg, ","); tok; tok = strtok(NULL, ",")) {
                if (n_dev == QM2_MAX_DEVICES) { printf("At most %d devices\n", QM2_MAX_DEVICES); return 1; }
                char *end = NULL;
                const long v = strtol(tok, &end, 10);
                if (end == tok || *end != '\0' || v < 0 || v > 1023) {
                    /* (an ordinal the machine does not have is refused by qm2_open: "device N out of range") */
                    printf("--devices / --shards: '%s' is not a device ordinal\n", tok);
                    return 1;
                }
                devices[n_dev++] = (int)v;
            }
            break;
        case 'h': help_count(); return 1;
        case 't':
            thread_count = (unsigned)atoi(optarg) & 0xFFu;   /* uint8_t in the reference (:306) */
            getopt_seen_t = 1;
            printf("[Option] Set %u threads\n", thread_count);
            break;
        case '?':
            puts("Option error, check help");
            help_count();
            return 1;
        default: return 1;
        }
    }
    if (argc < 4) { help_count(); return 1; }
    const char *ref_prefix = argv[argc - 3], *reads = argv[argc - 2], *out_prefix = argv[argc - 1];
    /* -t sized the reference's probing workers (:372-384); those are GPU waves now.  Here it sizes
     * the host threads that strip FASTA/FASTQ blocks into pinned staging (default 4). */
    /* default: 12 where the machine has the cores, else 8 or 4.  Since the readers count newlines at memory speed 8 of them
     * fill the PCIe link (hg38 scale, 30.7 GB of FASTQ from /dev/shm, a 16-core share of a 256-thread host: -t 8 / 12 / 16 / 24
     * -> 19.0 / 22.9 / 19.3 / 20.9 G bases/s; chr22 scale 20.4 / 23.5 / 21.6); before that it was 12.6 / 20.8 / 19.2 at 12 / 16 / 24 */
    const long cores = sysconf(_SC_NPROCESSORS_ONLN);
    /* (round 5: with the packing on the reader threads the link is no longer the bound, the threads are — hg38 scale, a 16-CPU share:
     *  -t 12 / 16 -> 41 / 50 G bases/s of FASTQ, profiles/r05_c3/cli_end_to_end_100M_reads.txt) */
    int n_strip = getopt_seen_t ? (int)thread_count : (cores >= 32 ? 16 : cores >= 24 ? 12 : cores >= 16 ? 8 : 4);
    if (n_strip > 32) n_strip = 32;

    const int timing = getenv("QM2_TIMING") != NULL;   /* extra phase lines; off by default so that
                                                          stdout stays the reference's */
    /* the reader threads, the pinned buffers they fill and the .qm's way to the device on the socket the (first) GPU hangs off */
    qm2_reader_note_affinity();
    if (n_dev <= 1) (void)qm2_bind_near_device(n_dev == 1 ? devices[0] : -1, NULL);      /* (several GPUs: they may sit on both sockets) */
    struct timespec ta, tb;
    qm2_dict_file d;
    if (qm2_dict_open(ref_prefix, &d)) return 1;
    if (d.k < 1 || d.k > 31) {
        printf("Dictionary k=%u is not supported (1..31)\n", d.k);
        qm2_dict_close(&d);
        return 1;
    }
    /* (the reference's Kmer_hash and Kmer_depth, :353 and :360, live on the device: qm2_open allocates
     * both and zeroes the counters; the keys are streamed from the file, never held on the host) */
    qm2_ctx *ctx = NULL;
    if (shards) {
        const int rc = count_on_shards(&d, reads, n_strip, devices, n_dev, ref_prefix, out_prefix, timing);
        qm2_dict_close(&d);
        if (rc) return 1;
        puts("Exit quicK-mer2 count");                        /* :543 */
        return 0;
    }
    if (n_dev > 1) {
        qm2_ctx *ctxs[QM2_MAX_DEVICES] = {NULL};
        qm2_stats st;
        struct timespec t0, t1;
        progress_printed_g = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int rc = count_on_devices(&d, reads, n_strip, devices, n_dev, ctxs, &st, timing);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (!rc) {
            progress_upto(st.n_kmers);
            printf("Counting elapse %u sec, total %lu kmers\n", (unsigned)(t1.tv_sec - t0.tv_sec), (unsigned long)st.n_kmers);   /* :481 */
            if (timing) printf("QM2 phase load+count on %d devices %.3f s\n", n_dev, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
            rc = qm2_write_outputs(&d, ctxs[0], ref_prefix, out_prefix);
        }
        for (int i = 0; i < n_dev; i++) qm2_close(ctxs[i]);
        qm2_dict_close(&d);
        if (rc) return 1;
        puts("Exit quicK-mer2 count");                        /* :543 */
        return 0;
    }
    qm2_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o;
    o.device = n_dev == 1 ? devices[0] : -1;
    o.flags = QM2_FLAG_KEYS_LATER;
    o.n_staging = (uint32_t)n_strip + 3;
    int regular, fastq;
    uint64_t in_size;
    sniff_input(reads, &regular, &fastq, &in_size);
    if (regular) o.kmers_per_kib = estimate_kmers_per_kib(reads, d.k, fastq);
    /* The CLI's device batch (default_batch_bytes: 1.5 bytes per table slot, 1 to 2 GiB).  A regular file smaller
     * than that gets one batch of its size: qm2_prepare then allocates (and the driver clears) a fraction of the
     * partition scratch — 22 bytes per batch byte (11 for FASTQ text). */
    o.batch_bytes = default_batch_bytes(d.hash_size);
    if (regular) {
        uint64_t need = (in_size + (256ull << 20)) & ~((256ull << 20) - 1);
        if (need < (1ull << 30)) need = 1ull << 30;
        if (need < o.batch_bytes) o.batch_bytes = need;
    }
    if (getenv("QM2_BATCH_BYTES")) o.batch_bytes = strtoull(getenv("QM2_BATCH_BYTES"), NULL, 0);   /* (experiments) */
    clock_gettime(CLOCK_MONOTONIC, &ta);
    const int want_raw = reader_mode() == READER_RAW;
    const int want_packed = reader_mode() == READER_PACKED && n_strip > 0;
    struct timespec t_open, t_prep;
    if (qm2_open(&ctx, NULL, d.hash_size, d.k, &o)) { qm2_dict_close(&d); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t_open);
    /* every buffer of the counting phase is allocated here, like the reference's mallocs and thread start-up
     * ahead of its timed loop (:353-384): nothing but copies and kernels between the first block and the drain.
     * The partition buffers — the slow part — are made while the dictionary streams in. */
    const uint32_t what = QM2_PREPARE_HOST | (want_raw ? QM2_PREPARE_RAW : 0u) | (want_raw && regular && fastq ? QM2_PREPARE_FASTQ : 0u);
    if (qm2_prepare(ctx, 0, what | QM2_PREPARE_NO_PART)) { qm2_close(ctx); qm2_dict_close(&d); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &t_prep);
    if (load_keys_and_prepare(&d, ctx, what, 1)) { qm2_close(ctx); qm2_dict_close(&d); return 1; }
    clock_gettime(CLOCK_MONOTONIC, &tb);
    if (timing)
        printf("QM2 phase load-dictionary %.3f s (file -> device; open %.3f + staging and batches %.3f + keys beside the partition buffers %.3f)\n",
               (tb.tv_sec - ta.tv_sec) + 1e-9 * (tb.tv_nsec - ta.tv_nsec), (t_open.tv_sec - ta.tv_sec) + 1e-9 * (t_open.tv_nsec - ta.tv_nsec),
               (t_prep.tv_sec - t_open.tv_sec) + 1e-9 * (t_prep.tv_nsec - t_open.tv_nsec), (tb.tv_sec - t_prep.tv_sec) + 1e-9 * (tb.tv_nsec - t_prep.tv_nsec));

    uint64_t allocs_before = 0;
    if (timing) {
        qm2_profile pr;
        memset(&pr, 0, sizeof pr);
        (void)qm2_get_profile(ctx, &pr);
        allocs_before = pr.allocations;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);                      /* :386-387 */
    /* Regular files are not stripped on the host: the workers only pread and the device does the line
     * discipline (measured 0.25 -> 0.19 s for 3 GB of FASTQ), the reference's corner
     * cases included (uint16 run-counter wraps of long lines, '>' at FASTQ sequence positions).  Only
     * absurd amounts of those are refused (known at the end of the device batch that holds them):
     * the whole input then goes through the host path again, from zeroed counters — exact either
     * way, nothing extra on stdout.  QM2_READER_RAW=0 goes straight to the host path. */
    progress_printed_g = 0;
    progress_ctx[0] = ctx;
    progress_n = 1;
    qm2_lines_out out = {out_acquire, out_commit, out_submit, ctx, want_raw ? out_commit_raw : NULL, want_packed ? out_commit_packed : NULL};
    int rc = qm2_stream_reads(reads, &out, n_strip);
    qm2_stats st;
    memset(&st, 0, sizeof st);
    if (rc == 0) rc = qm2_finish(ctx, NULL, &st);             /* :458-479; counts stay on the device */
    if (rc == 3 && out.commit_packed) {
        /* packed pieces are committed in any order; a FASTQ sequence-position line that starts with '>' (:398) changes what the
         * lines BEHIND it are, which only a pass in order can follow: everything again, from zeroed counters, as ASCII lines */
        if (timing) printf("QM2 packed input given up ('>' at a FASTQ sequence position): host line discipline in order\n");
        out.commit_packed = NULL;
        rc = qm2_reset(ctx);
        if (!rc) rc = qm2_stream_reads(reads, &out, n_strip);
        memset(&st, 0, sizeof st);
        if (!rc) rc = qm2_finish(ctx, NULL, &st);
    }
    if (want_raw && (rc == 2 || (rc == 0 && qm2_raw_verdict(ctx) != 0))) {
        if (timing) printf("QM2 raw input refused (verdict %u): host line discipline\n", qm2_raw_verdict(ctx));
        out.commit_raw = NULL;
        /* (progress lines already printed stay printed: the host pass prints only what lies beyond them) */
        rc = qm2_reset(ctx);
        if (!rc) rc = qm2_stream_reads(reads, &out, n_strip);
        memset(&st, 0, sizeof st);
        if (!rc) rc = qm2_finish(ctx, NULL, &st);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) { qm2_close(ctx); qm2_dict_close(&d); return 1; }
    progress_upto(st.n_kmers);
    printf("Counting elapse %u sec, total %lu kmers\n",       /* :481 */
           (unsigned)(t1.tv_sec - t0.tv_sec), (unsigned long)st.n_kmers);

    if (timing) {
        qm2_profile pr;
        memset(&pr, 0, sizeof pr);
        (void)qm2_get_profile(ctx, &pr);
        printf("QM2 phase count %.3f s (%.2f G bases/s); allocations inside it: %llu; input %s\n",
               (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec),
               st.n_bases / ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / 1e9,
               (unsigned long long)(pr.allocations - allocs_before),
               out.commit_raw ? "raw text, line discipline on the device" : out.commit_packed ? "packed on the host" : "ASCII lines from the host");
    }
    clock_gettime(CLOCK_MONOTONIC, &ta);
    rc = qm2_write_outputs(&d, ctx, ref_prefix, out_prefix);
    clock_gettime(CLOCK_MONOTONIC, &tb);
    if (timing) printf("QM2 phase epilogue %.3f s\n", (tb.tv_sec - ta.tv_sec) + 1e-9 * (tb.tv_nsec - ta.tv_nsec));
    qm2_close(ctx);
    qm2_dict_close(&d);
    if (rc) return 1;
    puts("Exit quicK-mer2 count");                            /* :543 */
    return 0;
}
