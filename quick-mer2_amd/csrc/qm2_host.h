/*
 * qm2_host.h — host-side (plain C) pieces of `quicKmer2 count` around the
 * C-ABI of include/qm2.h: on-disk formats, the input line discipline and the
 * command itself.  Mirrors the reference's main_count (QuicKmer.c:304-545)
 * region by region; nothing here touches the GPU directly.
 */
#ifndef QM2_HOST_H
#define QM2_HOST_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qm2.h"

/* ---- .qm dictionary ---------------------------------- */
typedef struct {
    FILE     *fp;          /* left open: keys and chain are streamed from it later  */
    uint32_t  k;           /* byte @4  (:345-347)                                */
    uint64_t  hash_size;   /* u64 @8   (:348-349)                                */
    uint64_t  first_idx;   /* u64 @16  (:350-351)                                */
} qm2_dict_file;

/* Header only.  The reference mallocs H x 8 bytes and freads the keys here (:353-359), later
 * the chain over them (:483, :490); this build streams both from the file straight to the
 * device (qm2_dict_load_keys -> qm2_keys_from_fd, qm2_write_outputs -> qm2_chain_gather_fd), so
 * the host never holds the 32 GiB array. */
int  qm2_dict_open(const char *ref_prefix, qm2_dict_file *d);
int  qm2_dict_load_keys(qm2_dict_file *d, qm2_ctx *ctx);      /* prints the reference's "Read 0x.. hash" (:359) */
int  qm2_dict_load_keys_quiet(qm2_dict_file *d, qm2_ctx *ctx, int print);   /* the same for a further replica: one line in all */
void qm2_dict_close(qm2_dict_file *d);

/* Epilogue (:483-542): counts in chain order -> <out>.bin, optional .qgc stats ->
 * <out>.txt.  The chain walk itself (:494-516) runs on the device (qm2_chain_gather_fd reads
 * the chain from the .qm); the result is streamed out.  Prints the reference's stdout lines. */
int qm2_write_outputs(qm2_dict_file *d, qm2_ctx *ctx, const char *ref_prefix, const char *out_prefix);

/* ---- input line discipline (:393-398, :451-455) ---------------------------- */
/* Where the stripped sequence lines go: `acquire` hands out a buffer to fill (the CLI: a
 * pinned staging buffer of the backend, qm2_staging_acquire), `commit` takes its first n
 * bytes (whole '\n'-terminated lines; n may be 0), `submit` takes a pageable buffer and is
 * only used for a single line longer than an acquired buffer. */
typedef struct {
    int (*acquire)(void *user, char **buf, size_t *capacity);
    int (*commit)(void *user, char *buf, size_t n);
    int (*submit)(void *user, const char *lines, size_t n);
    void *user;
    /* optional: when set, regular files are NOT stripped on the host; the workers only
     * pread their blocks into acquired buffers and the n raw bytes at `offset` are handed over with
     * the line state at their first byte (fastq, lines still to drop), in file order.  n = 0 just
     * returns the buffer.  Returns 0, 1 (error, message printed) or 2 (give up on raw hand-over:
     * qm2_stream_reads returns 2). */
    int (*commit_raw)(void *user, char *buf, size_t offset, size_t n, int fastq, unsigned skip);
    /* optional (ABI 6, include/qm2.h "packed sequence"): when set (and commit_raw is not), the workers run the line
     * discipline AND pack what it keeps — qm2_pack_reads, straight into the acquired buffer: 2-bit codes and reset flags,
     * 0.375 bytes per base — and hand over the piece here, in input order.  What does not pack (a piece with more
     * run-counter wraps than the list holds) and what the serial parts of the reader produce still comes through
     * commit / submit as ASCII lines. */
    int (*commit_packed)(void *user, char *buf, const qm2_packed *p, const uint64_t *wraps);
} qm2_lines_out;

/* Streams `path` (file, pipe or /dev/fd/N) with a read-ahead thread, strips it to sequence
 * lines on n_threads worker threads (0 = on the calling thread) and hands them to `out`, which
 * is only ever called from the calling thread and must be able to lend n_threads + 1 buffers at
 * a time.  Returns 0, or 1 after printing a message, or 2 when raw hand-over (commit_raw) met
 * something it cannot pass on (a line longer than a buffer): nothing is printed, the caller
 * starts over without commit_raw; or 3 when packed hand-over (commit_packed) of a regular file met a
 * FASTQ sequence-position line that starts with '>' (its pieces are committed in any order, so what
 * such a line does to the lines behind it cannot be followed): nothing is printed, the caller starts
 * over — from zeroed counters — without commit_packed. */
int qm2_stream_reads(const char *path, const qm2_lines_out *out, int n_threads);
/* Call BEFORE narrowing the process's CPU affinity (qm2_bind_near_device): remembers the CPUs the process may use, so that
 * the block reader can put its workers on the NUMA node a regular file's cached pages lie on (qm2_reader.c, file_pages_cpus).
 * Never called: the workers stay where the process is. */
void qm2_reader_note_affinity(void);
/* bytes of text per block (= per acquired buffer) when a regular file is read block-wise */
size_t qm2_reader_block_bytes(void);

/* ---- the command ----------------------------------------------------------- */
int qm2_main_count(int argc, char **argv);
int qm2_main_est(int argc, char **argv);      /* qm2_est.c: QuicKmer.c:555-685 */

#endif
