/* tools/reader_bench.c — throughput of the host reader alone (qm2_reader.c, no GPU): stripped bytes per second
 * for a FASTA/FASTQ file and a thread count.  gcc -O3 -pthread -I quick-mer2_amd/csrc tools/reader_bench.c
 * quick-mer2_amd/csrc/qm2_reader.c; QM2_READER_STREAM / QM2_READER_BLOCKS pick the arrangement. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "qm2_host.h"
static size_t CAP; static size_t total;
static char *pool[64]; static int np;
static int acq(void *u, char **b, size_t *c) { (void)u; *b = np ? pool[--np] : malloc(CAP); *c = CAP; return *b == NULL; }
static int com(void *u, char *b, size_t n) { (void)u; total += n; pool[np++] = b; return 0; }
static int sub(void *u, const char *p, size_t n) { (void)u; (void)p; total += n; return 0; }
int main(int argc, char **argv) { CAP = 64u << 20;
  qm2_lines_out out = {acq, com, sub, NULL};
  for (int rep = 0; rep < 3; rep++) { total = 0; struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
  int rc = qm2_stream_reads(argv[1], &out, atoi(argv[2])); clock_gettime(CLOCK_MONOTONIC, &b);
  double s = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
  printf("rc=%d out=%zu bytes  %.3f s  %.2f GB/s stripped\n", rc, total, s, total / s / 1e9); }
  return 0; }
