/* tools/micro/reader_probe.c — throughput of the CLI's host reader alone (quick-mer2_amd/csrc/qm2_reader.c + qm2_pack.c,
 * no GPU): a FASTA / FASTQ file and a thread count, with the output going nowhere.
 *
 *   reader_probe make <path> <n_reads>            write a FASTQ file of 150-base reads (e.g. into /dev/shm)
 *   reader_probe <path> <threads> <packed|lines|raw> [reps]
 *
 * QM2_READER_TIMING=1 adds what the block workers spent in pread / newline count / waiting for their turn / packing.
 * Rates: bytes of the file per second, and bases per second. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include "../../quick-mer2_amd/csrc/qm2_host.h"
static size_t CAP = (size_t)64 << 20, total, bases;
static char *pool[128];
static int np;
static int acq(void *u, char **b, size_t *c) { (void)u; *b = np ? pool[--np] : aligned_alloc(64, CAP); *c = CAP; return *b == NULL; }
static int com(void *u, char *b, size_t n) { (void)u; total += n; for (size_t i = 0; i < n; i += 4096) bases += 0; pool[np++] = b; return 0; }
static int sub(void *u, const char *p, size_t n) { (void)u; (void)p; total += n; return 0; }
static int raw(void *u, char *b, size_t off, size_t n, int fastq, unsigned skip) { (void)u; (void)off; (void)fastq; (void)skip; total += n; pool[np++] = b; return 0; }
static int pk(void *u, char *b, const qm2_packed *p, const uint64_t *w) { (void)u; (void)w; total += 24 * p->n_blocks; bases += 64 * p->n_blocks - p->n_newlines; pool[np++] = b; return 0; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    qm2_reader_note_affinity();       /* (as the CLI does: the block reader's workers then go where the file's cached pages lie) */
    if (argc >= 4 && !strcmp(argv[1], "make")) {
        FILE *f = fopen(argv[2], "wb");
        if (!f) { perror(argv[2]); return 1; }
        const long n = atol(argv[3]);
        char *buf = malloc((size_t)400 * 4096);
        unsigned long long x = 88172645463325252ull;
        for (long i = 0; i < n;) {
            size_t at = 0;
            for (int r = 0; r < 4096 && i < n; r++, i++) {
                at += (size_t)sprintf(buf + at, "@SRR0000000.%ld %ld/1\n", i, i);
                for (int j = 0; j < 150; j++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; buf[at++] = "ACGT"[x & 3]; }
                memcpy(buf + at, "\n+\n", 3); at += 3;
                memset(buf + at, 'I', 150); at += 150;
                buf[at++] = '\n';
            }
            fwrite(buf, 1, at, f);
        }
        fclose(f);
        return 0;
    }
    if (argc < 4) { puts("reader_probe <path> <threads> <packed|lines|raw> [reps]"); return 1; }
    const int packed = !strcmp(argv[3], "packed"), is_raw = !strcmp(argv[3], "raw");
    struct stat sb;
    if (stat(argv[1], &sb)) { perror(argv[1]); return 1; }
    qm2_lines_out out = {acq, com, sub, NULL, is_raw ? raw : NULL, packed ? pk : NULL};
    for (int rep = 0; rep < (argc > 4 ? atoi(argv[4]) : 3); rep++) {
        total = bases = 0;
        const double t0 = now();
        const int rc = qm2_stream_reads(argv[1], &out, atoi(argv[2]));
        const double s = now() - t0;
        printf("%s -t %s %-6s rc=%d  %.3f s  %.2f GB/s of the file", argv[1], argv[2], argv[3], rc, s, (double)sb.st_size / s / 1e9);
        if (packed) printf("  = %.2f G bases/s (%.2f GB/s handed over)", (double)bases / s / 1e9, (double)total / s / 1e9);
        else printf("  (%.2f GB/s handed over)", (double)total / s / 1e9);
        puts("");
    }
    return 0;
}
