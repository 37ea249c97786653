/*
 * qm2_tools.h — synthetic-workload tooling that runs on the device.
 *
 * NOT part of the `count` path and not a replacement for anything in the
 * reference: it exists so that bench.py and the full-size GPU tests can make
 * dictionaries and read streams of the shapes BASELINE.json names (chr22-like
 * H=2^28, hg38-like H=2^32, HiFi reads) on a box that has no hg38 and no
 * network, in seconds instead of the hours the reference's `search` would take.
 * The dictionaries it builds obey the reference's table rules (home slot =
 * DJB & (H-1), walk towards the middle, QuicKmer.c:866-876; unique k-mers only,
 * :1218-1231; k-mer 0 never stored, :864) so the count path cannot tell them
 * from a `search`-built one.  Everything is a pure function of (seed, index):
 * the CPU baseline consumes byte-identical reads by copying a slice back.
 */
#ifndef QM2_TOOLS_H
#define QM2_TOOLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t start, len; } qm2t_range;          /* an 'N' block        */
typedef struct { uint64_t dst, src, len; } qm2t_dup;         /* dst copies src      */

/* Fill d_genome[0..n) with ACGT (and 'N' inside n_blocks).  GC content is drawn
 * per 10 kb segment from U(gc_lo, gc_hi).  dups[] (host array) make dst a copy
 * of src (src must not itself lie in a dup's dst). */
int qm2t_genome_generate(void *d_genome, uint64_t n, uint64_t seed, float gc_lo, float gc_hi,
                         const qm2t_range *n_blocks, uint32_t n_n_blocks,
                         const qm2t_dup *dups, uint32_t n_dups, void *stream);

/* Build the table of the UNIQUE canonical k-mers of d_genome (one record, no
 * newlines; 'N' resets) into d_keys[H].  d_scratch is H x u64 of scratch.
 * Outputs (host pointers, may be NULL): distinct and unique k-mer counts. */
int qm2t_dict_build(const void *d_genome, uint64_t n, uint32_t k, uint64_t *d_keys, uint64_t H,
                    uint64_t *d_scratch, uint64_t *n_distinct, uint64_t *n_unique, void *stream);

/* Write n_reads sequence lines into d_out: read r (global index first_read + r)
 * starts at a uniform genome position, has length U[len_lo, len_hi], is reverse-
 * complemented with p = 0.5, gets substitutions at sub_rate per base and, with
 * probability n_read_frac, one 'N'.  from[] (host array, may be empty) restricts the start
 * positions to those regions (real reads do not come out of assembly gaps).  d_offsets: device array of n_reads+1 u64
 * byte offsets (from qm2t_read_offsets).  Each line is '\n'-terminated. */
int qm2t_reads_generate(const void *d_genome, uint64_t n, void *d_out, const uint64_t *d_offsets,
                        uint64_t n_reads, uint64_t first_read, uint32_t len_lo, uint32_t len_hi,
                        uint64_t seed, float sub_rate, float n_read_frac,
                        const qm2t_range *from, uint32_t n_from, void *stream);

/* Host helper: offsets[0..n_reads] for the same (seed, first_read, lens). */
void qm2t_read_offsets(uint64_t *offsets, uint64_t n_reads, uint64_t first_read,
                       uint32_t len_lo, uint32_t len_hi, uint64_t seed);

/* Random-access microbenchmarks that give the roofline denominator its
 * empirical ceiling.  Each touches n_ops uniformly random
 * slots of a table of `slots` entries; returns the kernel time in ms.
 *   mode 0: 8-byte loads        mode 1: 4-byte no-return atomic adds
 *   mode 2: 8-byte load, then an atomic add on a second array at the same index */
int qm2t_microbench(uint64_t slots, uint64_t n_ops, int mode, int repeats, double *ms_out);

/* Parity tooling: out[0] = sum over i of c[i] * (splitmix64(i) | 1) mod 2^64 of the n u32 counters at
 * d_counters (16-byte aligned; position-weighted, so a count that lands on another slot changes it),
 * out[1] = plain sum, out[2] = non-zero slots.  Two runs over a 2^32-slot table are compared with it
 * without holding two 16 GiB arrays (bench.py: partitioned pipeline vs the direct kernel at the batch
 * geometry the number is measured on). */
int qm2t_checksum_u32(const void *d_counters, uint64_t n, uint64_t out[3], void *stream);

#ifdef __cplusplus
}
#endif
#endif
