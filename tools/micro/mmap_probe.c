/* tools/micro/mmap_probe.c — how should the reader threads get at a FASTQ file that sits in the page cache (/dev/shm)?
 *
 *   mmap_probe <path> <threads> <block MiB>
 *
 * Every thread takes blocks round-robin and runs the packer (qm2_pack_reads; the line state is made up — this is a timing
 * probe, not a count) over each, reaching the bytes three ways:
 *   pread    into a private buffer of one block (what qm2_reader.c does), then pack from it
 *   mmap     the file mapped once (MAP_SHARED), packed straight from the mapping, pages faulted in as they are touched
 *   mmap+populate   the same with MAP_POPULATE (the time of the mmap call is counted)
 *   mmap, per block the worker itself maps a block in (MADV_POPULATE_READ), packs from it and lets it go again (MADV_DONTNEED):
 *                   the page-table work of the two modes above, spread over the threads
 * Rates: bytes of the file per second. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "../../include/qm2.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { int fd, id, nt, mode; size_t size, block; const char *map; uint64_t bases; } job;
static void *run(void *arg)
{
    job *j = arg;
    char *buf = j->mode == 0 ? malloc(j->block + 4096) : NULL;
    char *out = aligned_alloc(64, qm2_packed_bound(j->block + 4096) + 64);
    uint64_t wraps[64];
    for (size_t b = (size_t)j->id; b * j->block < j->size; b += (size_t)j->nt) {
        const size_t off = b * j->block, n = j->size - off < j->block ? j->size - off : j->block;
        const char *p;
        if (j->mode == 0) { ssize_t r = pread(j->fd, buf, n, (off_t)off); if (r != (ssize_t)n) { perror("pread"); exit(1); } p = buf; }
        else p = j->map + off;
#ifdef MADV_POPULATE_READ
        if (j->mode == 3 && madvise((void *)(j->map + off), n, MADV_POPULATE_READ)) { perror("MADV_POPULATE_READ"); exit(1); }
#endif
        const char *first = memchr(p, '\n', n);                       /* from a line start on, whole lines */
        const char *last = memrchr(p, '\n', n);
        if (!first || last <= first) continue;
        qm2_line_state st = {1, 0, 0, 0};
        qm2_packed pk;
        if (qm2_pack_reads(first + 1, (size_t)(last - first), &st, out, qm2_packed_bound(j->block + 4096), wraps, 64, &pk)) { puts("pack failed"); exit(1); }
        j->bases += 64 * pk.n_blocks - pk.n_newlines;
        if (j->mode == 3) madvise((void *)(j->map + off), n, MADV_DONTNEED);
    }
    free(buf); free(out);
    return NULL;
}
int main(int argc, char **argv)
{
    if (argc < 4) { puts("mmap_probe <path> <threads> <block MiB>"); return 1; }
    const int nt = atoi(argv[2]);
    const size_t block = (size_t)atol(argv[3]) << 20;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) { perror(argv[1]); return 1; }
    static const char *names[4] = {"pread", "mmap", "mmap+populate", "mmap, per block"};
    for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 4; mode++) {
        const double t0 = now();
        const char *map = NULL;
        if (mode) {
            map = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
            if (map == MAP_FAILED) { perror("mmap"); return 1; }
        }
        const double t1 = now();
        job *jobs = calloc((size_t)nt, sizeof *jobs);
        pthread_t *th = calloc((size_t)nt, sizeof *th);
        for (int t = 0; t < nt; t++) { jobs[t] = (job){fd, t, nt, mode, (size_t)sb.st_size, block, map, 0}; pthread_create(&th[t], NULL, run, &jobs[t]); }
        uint64_t bases = 0;
        for (int t = 0; t < nt; t++) { pthread_join(th[t], NULL); bases += jobs[t].bases; }
        const double t2 = now();
        if (mode) munmap((void *)map, (size_t)sb.st_size);
        const double t3 = now();
        printf("%2d threads, %3zu MiB blocks  %-14s %.3f s (map %.3f + pack %.3f + unmap %.3f)  %.2f GB/s of the file\n", nt, block >> 20, names[mode],
               t3 - t0, t1 - t0, t2 - t1, t3 - t2, (double)sb.st_size / (t3 - t0) / 1e9);
        free(jobs); free(th);
        (void)bases;
    }
    return 0;
}
