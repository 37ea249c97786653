/* tools/micro/pack_probe.c — the host packers of quick-mer2_amd/csrc/qm2_pack.c alone: bytes of text per second and thread.
 *
 *   pack_probe [threads=1,2,4,8,12,16] [MiB of text per thread=256] [isa=auto|avx512vbmi2|avx512|avx2|portable] [kinds=0123] [reps=1]
 *
 * Every thread packs its own slice of synthetic text from pageable memory into its own output buffer (no GPU, no I/O):
 *   lines   150-base sequence lines                                    -> qm2_pack_lines
 *   fastq   "@r..\n" + 150 bases + "\n+\n" + 150 qualities + "\n"      -> qm2_pack_reads (line discipline + packing)
 *   fasta   ">r..\n" + 150 bases + "\n"                                -> qm2_pack_reads
 * beside memcpy of the same text (what the ASCII path's staging copy costs).  Rates are text bytes in, GB/s; kinds picks the
 * rows (0 lines, 1 fastq, 2 fasta, 3 memcpy), reps > 1 prints the best of so many runs over the same text. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/qm2.h"

typedef struct { const char *text; size_t n; char *out; size_t out_cap; int kind; double sec; uint64_t bases; int rc; } job;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void *run(void *arg)
{
    job *j = arg;
    const double t0 = now();
    uint64_t wraps[64];
    qm2_packed p = {0, 0, 0};
    const size_t piece = (size_t)16 << 20;                /* what a reader thread handles at a time */
    for (size_t off = 0; off < j->n;) {
        size_t take = j->n - off < piece ? j->n - off : piece;
        if (off + take < j->n) {                          /* whole lines (records start at '@' / '>' in this text) */
            const char *cut = memrchr(j->text + off, j->kind == 1 ? '@' : j->kind == 2 ? '>' : '\n', take);
            take = (size_t)(cut - (j->text + off)) + (j->kind ? 0 : 1);
        }
        if (j->kind == 3) memcpy(j->out, j->text + off, take);
        else if (j->kind == 0) j->rc |= qm2_pack_lines(j->text + off, take, j->out, j->out_cap, wraps, 64, &p);
        else {
            qm2_line_state st = {j->kind == 1, j->kind == 1, 0, 0};      /* a piece starts at a header: FASTQ drops one line first */
            j->rc |= qm2_pack_reads(j->text + off, take, &st, j->out, j->out_cap, wraps, 64, &p);
        }
        j->bases += 64 * p.n_blocks - p.n_newlines;
        off += take;
    }
    j->sec = now() - t0;
    return NULL;
}

int main(int argc, char **argv)
{
    const char *tl = argc > 1 ? argv[1] : "1,2,4,8,12,16";
    const size_t mib = argc > 2 ? (size_t)atol(argv[2]) : 256;
    printf("packers: %s\n", qm2_pack_isa_select(argc > 3 && strcmp(argv[3], "auto") ? argv[3] : NULL));
    const char *kinds = argc > 4 ? argv[4] : "0123";
    const int reps = argc > 5 && atoi(argv[5]) > 0 ? atoi(argv[5]) : 1;
    static const char *names[4] = {"lines", "fastq", "fasta", "memcpy (lines)"};
    char list[256];
    snprintf(list, sizeof list, "%s", tl);
    for (char *tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
        const int nt = atoi(tok);
        if (nt < 1 || nt > 256) continue;
        for (int kind = 0; kind < 4; kind++) {
            if (!strchr(kinds, '0' + kind)) continue;
            job *jobs = calloc((size_t)nt, sizeof *jobs);
            pthread_t *th = calloc((size_t)nt, sizeof *th);
            for (int t = 0; t < nt; t++) {
                const size_t n = mib << 20;
                char *text = malloc(n + 512);
                size_t at = 0;
                uint64_t x = 88172645463325252ull + (uint64_t)t;
                while (at + 400 < n) {
                    if (kind == 1 || kind == 2) at += (size_t)sprintf(text + at, "%cr%u\n", kind == 1 ? '@' : '>', (unsigned)(at & 0xFFFFF));
                    for (int i = 0; i < 150; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; text[at++] = "ACGT"[x & 3]; }
                    text[at++] = '\n';
                    if (kind == 1) { text[at++] = '+'; text[at++] = '\n'; memset(text + at, 'I', 150); at += 150; text[at++] = '\n'; }
                }
                jobs[t].text = text; jobs[t].n = at; jobs[t].kind = kind;
                jobs[t].out_cap = kind == 3 ? ((size_t)16 << 20) + 512 : qm2_packed_bound((size_t)16 << 20) + 64;
                jobs[t].out = aligned_alloc(64, (jobs[t].out_cap + 63) & ~(size_t)63);
                memset(jobs[t].out, 0, jobs[t].out_cap);
            }
            double wall = 1e30, bytes = 0, bases = 0;
            int rc = 0;
            for (int rep = 0; rep < reps; rep++) {
                const double t0 = now();
                bytes = bases = 0;
                for (int t = 0; t < nt; t++) { jobs[t].bases = 0; pthread_create(&th[t], NULL, run, &jobs[t]); }
                for (int t = 0; t < nt; t++) {
                    pthread_join(th[t], NULL);
                    bytes += (double)jobs[t].n; bases += (double)jobs[t].bases; rc |= jobs[t].rc;
                }
                if (now() - t0 < wall) wall = now() - t0;
            }
            char rate[64] = "";
            if (kind != 3) snprintf(rate, sizeof rate, "  = %.2f G bases/s", bases / wall / 1e9);
            printf("%2d threads  %-15s %7.2f GB/s of text (%6.2f per thread)%s\n", nt, names[kind], bytes / wall / 1e9, bytes / wall / 1e9 / nt, rate);
            if (rc) { puts("a packer refused its input"); return 1; }
            for (int t = 0; t < nt; t++) { free((void *)jobs[t].text); free(jobs[t].out); }
            free(jobs); free(th);
        }
    }
    return 0;
}
