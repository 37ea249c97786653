/*
 * qm2_oracle.h — CPU restatement of QuicK-mer2's `count` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (quick-mer2_amd/) never links or calls it and has no CPU
 * fallback.
 *
 * Parity status: PINNED.  Every function below was checked bit-for-bit in the
 * build container against the real reference binary (oracle/_ref/quicKmer2,
 * compiled by oracle/Makefile from /root/reference/QuicKmer.c where it lies)
 * on config C1 and on the edge known-answer set; the resulting vectors are
 * committed under tests/golden/ (made by tests/golden/make_golden.py).
 *
 * All file:line citations are into the reference's QuicKmer.c.
 */
#ifndef QM2_ORACLE_H
#define QM2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t n_bases;   /* sequence characters seen (incl. 'N'), newlines excluded */
    uint64_t n_kmers;   /* k-mers emitted == reference's process_kmers (:445)      */
    uint64_t n_probes;  /* table slots inspected by Find_hash (:96-98)             */
    uint64_t n_hits;    /* counter increments (:291 / :443)                        */
} qm2o_stats;

/* DJBHash_encode, :66-76 */
uint64_t qm2o_djb(uint64_t kmer);

/* Find_hash, :90-99.  Returns 1 on match; *idx = final slot; *probes (may be
 * NULL) += slots inspected. */
int qm2o_find(const uint64_t *table, uint64_t hash_size, uint64_t kmer,
              uint64_t *idx, uint64_t *probes);

/* Rolling encoder + canonicaliser, :399-420, over a buffer of whole sequence
 * lines, every line '\n'-terminated.  Writes up to cap k-mers to out (out may
 * be NULL to count only).  Returns the number of k-mers the stream emits. */
uint64_t qm2o_extract(const char *lines, size_t n, uint32_t k,
                      uint64_t *out, uint64_t cap, uint64_t *n_bases);

/* Serial counting (:440-445) of a buffer of sequence lines into depth[H]
 * (uint16, wraps mod 65536).  st accumulates (caller zeroes it). */
int qm2o_count_lines(const uint64_t *table, uint64_t hash_size, uint32_t k,
                     const char *lines, size_t n, uint16_t *depth,
                     qm2o_stats *st);

/* The reference's threading model (:256-296, :368-384, :421-439, :458-479):
 * one producer (the caller) encoding, N workers probing 4096-k-mer batches,
 * atomic 16-bit increments.  nthreads == 0 falls through to the serial path.
 * The reference's fixed sleep(1) (:469) is replaced by an exact drain.
 * st->n_probes is not collected here (workers stay as lean as :287-292). */
int qm2o_count_lines_mt(const uint64_t *table, uint64_t hash_size, uint32_t k,
                        const char *lines, size_t n, uint16_t *depth,
                        int nthreads, qm2o_stats *st);

/* ---- file level (the rest of main_count) --------------------------------- */

typedef struct {
    uint32_t  k;            /* byte @4   (:345-347) */
    uint8_t   hdr_bytes[4]; /* bytes 4..7 as stored */
    uint64_t  hash_size;    /* u64 @8    (:348-349) */
    uint64_t  first_idx;    /* u64 @16   (:350-351) */
    uint64_t *keys;         /* H x u64 @24 (:353-359) */
    uint32_t *chain;        /* H x u32 @24+8H (:483, :490) */
} qm2o_dict;

int  qm2o_dict_load(const char *qm_path, qm2o_dict *d);
void qm2o_dict_free(qm2o_dict *d);

/* FASTA/FASTQ line discipline of :393-398 and :451-455: returns a malloc'd
 * buffer holding only the sequence lines (each '\n'-terminated) that
 * main_count would feed to the encoder.  `seekable` = 0 reproduces the
 * lost-first-line behaviour on pipes (:396 fseek fails). */
char *qm2o_strip_reads(const char *path, int seekable, size_t *n_out);

/* Epilogue, :483-542: walk the chain from first_idx, emit depth in genome
 * order to <out_prefix>.bin; if qgc_path is readable also write
 * <out_prefix>.txt and return mean depth through *mean_depth (NaN if none).
 * Returns number of .bin entries, or (uint64_t)-1 on error. */
uint64_t qm2o_epilogue(const qm2o_dict *d, const uint16_t *depth,
                       const char *qgc_path, const char *out_prefix,
                       double *mean_depth);

/* Whole `quicKmer2 count [-t N] ref.fa reads out_prefix` on the CPU. */
int qm2o_count_main(const char *ref_prefix, const char *reads_path,
                    const char *out_prefix, int nthreads, qm2o_stats *st,
                    double *count_seconds);

#ifdef __cplusplus
}
#endif
#endif
