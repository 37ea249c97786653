/*
 * qm2.h — C-ABI of the MI355X-native `quicKmer2 count` counting phase.
 *
 * The reference (KiddLab/QuicK-mer2, one C file) has no plugin or FFI seam:
 * main_count is monolithic and its workers reach the table through globals
 * (QuicKmer.c:14-23).  This header DEFINES the seam the replacement binds at:
 * the whole counting phase, between "dictionary loaded" (QuicKmer.c:366) and
 * "dump count file" (QuicKmer.c:482).  Inputs are the loaded key array and the
 * stream of sequence lines; the output is the filled Kmer_depth[H].
 *
 *   reference region replaced                          entry point
 *   -------------------------------------------------  -------------------------
 *   Kmer_hash/Kmer_depth alloc, QuicKmer.c:353-366      qm2_open
 *     fread of the key table, :353-359                    (qm2_open's keys, or qm2_keys_from_fd)
 *   thread pool start-up, :368-384                      qm2_open
 *   format sniff + line discipline, :393-398, :451-455  caller (qm2_submit takes sequence lines), or
 *                                                       qm2_staging_commit_raw / qm2_submit_raw_device
 *   per-line encode + dispatch loop, :397-456           qm2_submit / qm2_staging_commit / qm2_submit_device
 *     reset test + 2-bit code of a character, :403-411    (optionally on the host: qm2_pack_lines / qm2_pack_reads ->
 *                                                          qm2_staging_commit_packed / qm2_submit_packed_device)
 *     rolling 2-bit encoder + canonicaliser, :399-420
 *     Kmer_count_TSK batch worker, :256-296
 *     DJBHash_encode :66-76, Find_hash :90-99
 *     counter increment, :291 / :443
 *   drain + join, :458-479                              qm2_finish
 *   "total %lu kmers" bookkeeping, :445, :481           qm2_stats
 *   progress line every 2^30 k-mers, :446               qm2_stats_peek
 *   chain read + walk -> depth in genome order, :483-516  qm2_chain_gather(_fd) / qm2_chain_fetch
 *   GC-curve sums over the control k-mers, :501-509     qm2_chain_gc_sums_fd
 *   free, :519-520                                      qm2_close
 *
 * Plain C types only: no torch / HIP types cross this boundary (device
 * pointers and streams travel as void*).  The library that implements it is
 * quick-mer2_amd/libqm2_hip.so (hand-written gfx950 kernels).  There is no CPU
 * implementation of this ABI in the product: if no HIP device is usable every
 * entry point fails loudly (non-zero + message), it never falls back.
 *
 * Error convention (mirrors QuicKmer.c:354-358, `puts(msg); return 1;`):
 * 0 on success; non-zero after printing one line to stdout.  The same line is
 * available from qm2_last_error().  No exceptions, no abort().
 *
 * Threading: one caller thread per context.  Work is asynchronous inside the
 * backend; results are independent of batch boundaries, batch sizes and
 * scheduling (integer adds commute).  qm2_finish/qm2_sync are the only points
 * where the caller observes completion.
 */
#ifndef QM2_H
#define QM2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QM2_ABI_VERSION 6

typedef struct qm2_ctx qm2_ctx;

/* flags for qm2_opts.flags */
#define QM2_FLAG_TIMING      0x1u  /* bracket every kernel launch with HIP events
                                      on its own stream (qm2_get_profile)       */
#define QM2_FLAG_NO_STATS    0x2u  /* do not collect n_probes / n_hits          */
#define QM2_FLAG_DIRECT      0x4u  /* always use the direct kernel (random probes +
                                      global atomics), never the partitioned pipeline */
#define QM2_FLAG_PARTITION   0x8u  /* always use the partitioned pipeline (radix-
                                      partition by home slot, probe + count in LDS);
                                      default: chosen per batch from its size vs. H   */
#define QM2_FLAG_KEYS_LATER  0x10u /* qm2_open(keys = NULL): allocate the device key
                                      table but fill it later with qm2_keys_from_fd   */
#define QM2_FLAG_RAW_STRICT  0x20u /* raw text (qm2_staging_commit_raw / qm2_submit_raw_device): refuse a batch with ANY
                                      FASTQ sequence-position line that starts with '>' (QM2_RAW_DESYNC) instead of
                                      following up to 64 of them.  Such a line moves every later sequence line of the
                                      TEXT on by one; a context that sees only some of the text's batches (one of
                                      several replicas fed in turns, `--devices`) cannot know about those in the
                                      batches it did not see, so there the caller takes the host path for such input */
#define QM2_FLAG_SUBMIT_ASCII 0x40u /* qm2_submit copies the caller's ASCII lines to the device as they are instead of
                                      packing them into the pinned buffers (qm2_pack_lines): 1 byte per base over the link */

typedef struct {
    uint32_t struct_size;     /* = sizeof(qm2_opts); 0-filled fields = defaults  */
    int32_t  device;          /* HIP device ordinal; <0 = current device         */
    uint32_t flags;           /* QM2_FLAG_*                                      */
    uint32_t n_staging;       /* pinned host staging buffers for qm2_submit /
                                 qm2_staging_acquire (default 4)                 */
    uint64_t staging_bytes;   /* size of each staging buffer (default 64 MiB)    */
    uint64_t batch_bytes;     /* device batch that qm2_submit accumulates before it
                                 launches (default: 1.5 bytes per table slot,
                                 clamped to 1..6 GiB; two are kept).  The
                                 partitioned pipeline streams the whole table once
                                 per batch, so batches should be large against it */
    uint32_t part_max_bits;   /* partitioned pipeline: radix bits per level, 1..10
                                 (default 10); smaller values only make sense in tests */
    uint32_t kmers_per_kib;   /* what the caller expects of its input: k-mers per 1024 bytes of sequence lines
                                 (150-base reads at k = 30: 1024 x 121 / 151 = 820), 0 = 1024 (every byte may end a
                                 k-mer).  Only sizes the partition buffers (22 bytes of scratch per expected k-mer
                                 byte); input that yields more overflows into the spill list or is recounted by the
                                 direct kernel — exact either way */
    void    *ext_keys;        /* optional: device pointer to H x u64 keys that are
                                 ALREADY resident (then `keys` may be NULL).  They
                                 must not change while the context lives: the
                                 context keeps a digest of the table (look at the
                                 runs of occupied slots, table image) made once     */
    void    *ext_counters;    /* optional: device pointer to H x u32 counters
                                 owned by the caller (e.g. a tensor handed to
                                 RCCL for the replica reduce); zeroed by caller.
                                 COMPLETE ONLY BEHIND qm2_finish (or a call of
                                 qm2_counters_device / qm2_merge_counters): between
                                 batches part of the counts sits in a 2-byte array
                                 of the context's own (the private bucket interiors
                                 of the partitioned pipeline) and is added in there */
    void    *stream;          /* optional hipStream_t used by qm2_submit_device /
                                 qm2_encode_device / qm2_probe_device; NULL =
                                 a stream owned by the context                   */
    uint64_t span_lo;         /* hash-range shard: the slots      */
    uint64_t span_hi;         /* [span_lo, span_hi) of the table that are resident
                                 on this device; `keys`/ext_keys/ext_counters then
                                 hold span_hi - span_lo entries, slot span_lo first.
                                 Both 0 = the whole table.  Only k-mers whose probe
                                 walk stays inside the span may be fed (see
                                 qm2_shard_span for how the span is chosen).     */
    uint64_t own_lo;          /* with a span: the range [own_lo, own_hi) of HOME slots this   */
    uint64_t own_hi;          /* shard is fed (qm2_shard_span's own range: a power of two of
                                 slots, aligned to its size).  Given, qm2_probe_slabs_device
                                 runs the partitioned pipeline on the span; both 0: the
                                 shard is probed directly (random gathers + atomics).           */
} qm2_opts;

typedef struct {
    uint64_t n_bases;   /* sequence characters consumed (incl. 'N'), no newlines */
    uint64_t n_kmers;   /* k-mers emitted == the reference's process_kmers (:445) */
    uint64_t n_probes;  /* table slots inspected (Find_hash loop, :96-98)         */
    uint64_t n_hits;    /* counter increments (:291 / :443)                       */
} qm2_stats;

/* kernel classes timed separately under QM2_FLAG_TIMING */
enum {
    QM2_K_COUNT_DIRECT = 0,   /* count_kernel: fused encode + probe + atomics     */
    QM2_K_SPLIT_TEXT   = 1,   /* split_text_kernel: encode + level-1 partition    */
    QM2_K_SPLIT_KMER   = 2,   /* split_kmer_kernel: deeper partition levels       */
    QM2_K_BUCKET_COUNT = 3,   /* bucket_count_kernel: probe + count in LDS        */
    QM2_K_PROBE        = 4,   /* probe_kernel: spill list / qm2_probe_device      */
    QM2_K_ENCODE       = 5,   /* encode_kernel (qm2_encode_device)                */
    QM2_K_LINES        = 6,   /* raw_tile / raw_scan / raw_apply: line discipline   */
    QM2_K_CLASSES      = 8
};

typedef struct {
    uint64_t n_launches;      /* batches (qm2_submit_device calls / staged batches) timed */
    double   kernel_ms;       /* sum of all HIP-event kernel durations            */
    double   last_kernel_ms;  /* kernel time of the most recent batch             */
    uint64_t class_launches[QM2_K_CLASSES];
    double   class_ms[QM2_K_CLASSES];
    uint64_t partitioned_batches, direct_batches;
    uint64_t fallback_batches; /* partitioned batches whose spill list overflowed: recounted by the
                                  direct kernel (decided on the device; read at qm2_get_profile,
                                  which therefore waits for the work submitted so far) */
    uint64_t allocations;      /* device / pinned-host allocations the counting pipeline made since
                                  qm2_open or the last qm2_reset (0 between the first submit and
                                  qm2_finish once qm2_prepare has run)                          */
    uint64_t spilled_kmers;    /* partitioned batches: k-mers that did not fit their bucket and were probed from the
                                  spill list by the direct path (skewed input; 0 on uniform reads)            */
    uint64_t staged_bytes;     /* bytes the host pipeline has copied to the device (qm2_submit*, qm2_staging_commit*): what
                                  crossed the link — packed pieces count their blocks (ABI 6)                  */
    uint64_t refused_batches;  /* batches that were large enough for the partitioned pipeline but took the direct kernel because its
                                  buffers (partition scratch, interior counters, table image) did not fit in device memory;
                                  qm2_last_error() holds the allocation's message                               */
    uint64_t replans;          /* times the partition plan was torn down and rebuilt because the caller alternated between
                                  qm2_probe_subslabs_device and the other entry points (each costs a stream sync + the
                                  re-allocation of the partition buffers)                                      */
} qm2_profile;

/* Create a counting context for a dictionary of `hash_size` slots (power of
 * two, <= 2^32: the on-disk chain is uint32, QuicKmer.c:20) and k-mer size k
 * (1..31; k = 32 is undefined behaviour in the reference, :419, and rejected).
 * `keys` is the host array read from the .qm at offset 24 (QuicKmer.c:353-359);
 * it is uploaded and the caller may reuse the buffer at once (the reference
 * re-uses it for the chain, :483).  Counters start at zero (calloc, :360). */
int qm2_open(qm2_ctx **out, const uint64_t *keys, uint64_t hash_size,
             uint32_t k, const qm2_opts *opts);

/* Feed sequence lines from HOST memory: `n` bytes made of whole lines, each
 * '\n'-terminated, headers and quality lines already removed (what
 * QuicKmer.c:397-398/:451-455 hands to the loop at :403).  Any byte other than
 * '\n' and 'N' is a base, exactly as in the reference (:411).  Copies into
 * pinned staging and returns; the caller's buffer is free on return.  Lines of
 * any length are accepted (the reference's uint16 run counter, :402, is
 * reproduced: SURVEY B.3). */
int qm2_submit(qm2_ctx *c, const char *seq_bytes, size_t n);

/* Zero-copy variant for callers that produce the sequence lines themselves (the CLI's
 * reader strips FASTA/FASTQ straight into pinned memory): qm2_staging_acquire hands out one
 * of the context's pinned staging buffers (blocking until its previous copy has finished);
 * qm2_staging_commit submits the first n bytes of it under qm2_submit's contract (whole
 * '\n'-terminated lines) without the host memcpy and returns the buffer to the pool.  At most
 * n_staging - 1 buffers may be held at a time. */
int qm2_staging_acquire(qm2_ctx *c, char **buf, size_t *capacity);
int qm2_staging_commit(qm2_ctx *c, char *buf, size_t n);

/* Same, for bytes ALREADY in device memory (any alignment).  Asynchronous on
 * the context's stream; the buffer must stay valid until qm2_sync/qm2_finish.
 * `suppress`/`n_suppress` (device pointer to sorted u64 byte offsets, or
 * NULL/0) carries the positions whose windows the reference's uint16 run
 * counter drops; callers whose lines are all < 65536 bases pass NULL. */
int qm2_submit_device(qm2_ctx *c, const void *d_seq_bytes, size_t n,
                      const uint64_t *d_suppress, uint64_t n_suppress);

/* Launch the batch that is being filled (qm2_submit / qm2_staging_commit*) without waiting for it: a
 * caller that feeds several contexts in turns (quicKmer2 count --devices) hands a context its share
 * and moves on.  qm2_merge_counters adds the counters of `src` — a replica of the same table, or a
 * hash-range shard whose span lies inside dst's (its own slots and the overhang past its middle-facing
 * edge alike: slot for slot), on the same or on another device — into `dst` (both are synchronised first;
 * src keeps its own counts). */
int qm2_flush(qm2_ctx *c);
int qm2_merge_counters(qm2_ctx *dst, qm2_ctx *src);

/* Block until everything submitted so far has been counted. */
int qm2_sync(qm2_ctx *c);

/* Drain, then write the H counters truncated to uint16 (the reference's
 * counters wrap mod 65536, :23/:291; 32-bit accumulation then truncation is
 * exact) into depth_out (may be NULL) and the totals into st (may be NULL).
 * The context stays usable; counts keep accumulating until qm2_reset. */
int qm2_finish(qm2_ctx *c, uint16_t *depth_out, qm2_stats *st);

/* The totals as of the last batch whose kernels have finished — never blocks, may lag the stream
 * (the snapshot is copied out by the stream itself behind every batch).  The CLI prints the
 * reference's progress line `Read %liG kmers` (QuicKmer.c:446) from it, at batch granularity. */
int qm2_stats_peek(qm2_ctx *c, qm2_stats *st);

/* Allocate ahead of time what the counting phase needs for device batches of `batch_bytes` (0 = the
 * context's batch size): QM2_PREPARE_HOST the pinned staging buffers and the two device batches of
 * the qm2_submit / qm2_staging_* pipeline, QM2_PREPARE_RAW the scratch of the device line discipline;
 * always the partition buffers and (once the key table is there) the one-off look at its runs of
 * occupied slots.  Optional: without it all of that happens inside the first batches (the reference
 * does its allocations up front too, QuicKmer.c:353-384). */
#define QM2_PREPARE_HOST 1u
#define QM2_PREPARE_RAW  2u
/* the raw batches will be FASTQ text: at most half of a batch's bytes are sequence, so the partition buffers are
 * sized for half a batch (22 -> 11 bytes of scratch per batch byte: the largest share of the load phase at hg38
 * scale is allocating it).  Only a bound on memory: text that breaks it (qualities shorter than their sequences)
 * overflows into the spill list or is recounted by the direct kernel, exact either way. */
#define QM2_PREPARE_FASTQ 4u
/* everything BUT the partition buffers; a second call without this flag makes them — e.g. from another thread while
 * qm2_keys_from_fd streams the dictionary in (the two may run concurrently; nothing else of a context may) */
#define QM2_PREPARE_NO_PART 8u
/* with QM2_PREPARE_HOST: the device batches will only ever hold PACKED pieces (qm2_staging_commit_packed, qm2_submit's
 * default): they are made 3/8 the size (24 bytes per 64 positions).  Text that arrives all the same (a line too long
 * for a piece, a caller's ASCII) makes its batch grow then — an allocation inside the counting phase, nothing else. */
#define QM2_PREPARE_PACKED 16u
int qm2_prepare(qm2_ctx *c, uint64_t batch_bytes, uint32_t what);

/* Zero counters and statistics (a fresh calloc, :360). */
int qm2_reset(qm2_ctx *c);

/* The caller has rewritten the key table behind qm2_opts.ext_keys (or is about to count against other keys in the same
 * device array): everything the context derived from the keys — the look at the runs of occupied slots, the table image
 * bucket_count streams — is made again in front of the next batch.  Counters and statistics are not touched.  Without this
 * call a context goes on matching against the image of the OLD keys (the image is made once per context: include/qm2.h,
 * qm2_opts.ext_keys).  Waits for the work submitted so far.  ABI 6. */
int qm2_keys_changed(qm2_ctx *c);

void qm2_close(qm2_ctx *c);

/* Device pointers, for callers that reduce counters across GPUs with RCCL or
 * keep inputs resident: H x u32 counters, H x u64 keys. */
void *qm2_counters_device(qm2_ctx *c);
void *qm2_keys_device(qm2_ctx *c);

int qm2_get_profile(qm2_ctx *c, qm2_profile *p);

/* ---- split path for hash-range sharding (multi-GPU) ------- *
 * The fused kernel of qm2_submit does encode+probe in one launch.  When the
 * table is range-partitioned over G devices the two halves run on different
 * devices with an all-to-all in between, so they are also exported alone.     */

/* Encode device-resident sequence lines into canonical k-mers (same rules as
 * qm2_submit), bucketed by owner = home_slot >> (log2(H) - log2(n_buckets)).
 * d_out has n_buckets regions of `bucket_cap` u64 each; d_counts[n_buckets]
 * (u64, device) receives the fill of each and must be zero on entry.
 * k-mers beyond bucket_cap are dropped and reported by a non-zero return from
 * the next qm2_sync (size bucket_cap for the worst case: n k-mers). */
int qm2_encode_device(qm2_ctx *c, const void *d_seq_bytes, size_t n,
                      const uint64_t *d_suppress, uint64_t n_suppress,
                      uint64_t *d_out, uint64_t bucket_cap, uint32_t n_buckets,
                      uint64_t *d_counts);

/* The same two legs in the shape the multi-GPU driver wants them (quick-mer2_amd/multi.py):
 *
 * qm2_route_device    encode + bucket by owner at the speed of the partitioned pipeline's text level:
 *                     d_slabs has n_owners slabs of slab_cap u64 (slab_cap even, d_slabs 16-byte
 *                     aligned), d_fill[n_owners] (u32, zero on entry) receives how many k-mers were
 *                     ROUTED to each owner — min(d_fill[g], slab_cap) of them are in the slab, the
 *                     rest did not fit and are counted in d_call[2]; d_call = {bases, k-mers, k-mers
 *                     that did not fit} of this call (u64 x 3, zero on entry).  The context's own
 *                     statistics are not touched: the caller adds the totals of the calls it keeps
 *                     (a call with d_call[2] != 0 is repeated in smaller pieces, by every rank).
 * qm2_probe_slabs_device  the k-mers of n_slabs received slabs (slab s: the first
 *                     min(d_fill[s], slab_cap) entries at d_slabs + s * slab_cap) against this shard's
 *                     span.  With an own range in qm2_opts: partition by home slot + probe / count in
 *                     LDS, like a whole table; without: direct probes.  Adds to n_probes / n_hits. */
int qm2_route_device(qm2_ctx *c, const void *d_seq_bytes, size_t n, const uint64_t *d_suppress, uint64_t n_suppress,
                     uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t *d_fill, uint64_t *d_call);
int qm2_probe_slabs_device(qm2_ctx *c, const uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_slabs,
                           const uint32_t *d_fill);
/* qm2_route_device for text that starts in the MIDDLE of a line (a line longer than the caller's batch, cut by the
 * caller): the first `carry` bytes (<= 32; at least max(k, 31) - 1 of the line's bytes in front of the cut, or all of
 * them if the line is younger) are look-back only — windows that end inside them emit nothing, they were routed with
 * the previous piece — exactly what qm2_submit does with a line it has to cut (QuicKmer.c:415-416: the forward window
 * and the 30-base reverse-complement register both reach back).  Positions in d_suppress count from d_seq_bytes.  ABI 5. */
int qm2_route_device_carry(qm2_ctx *c, const void *d_seq_bytes, size_t n, uint32_t carry, const uint64_t *d_suppress,
                           uint64_t n_suppress, uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t *d_fill,
                           uint64_t *d_call);

/* The same when the senders have already partitioned what they route: qm2_route_device called with
 * n_owners x n_subs "owners" (<= 1024 in all) cuts every owner's slab into n_subs sub-slabs by the next
 * log2(n_subs) bits of the home slot, and what a shard then receives — d_slabs = [n_peers][n_subs] sub-slabs of
 * sub_cap k-mers (peer p's sub-slab 0 lies peer_stride regions behind peer p - 1's; 0 = n_subs: side by side), d_fill
 * their fills in the same order — IS the first level of its own partition (the sender takes
 * the place the XCD has in the text level's sub-regions), so the slab split of qm2_probe_slabs_device falls away:
 * text -> 10-bit route -> exchange -> ONE split -> count, as many passes as an unsharded table takes.
 * kmers_bound: an upper estimate of the k-mers in the call (sizes the partition buffers; 0 = every sub-slab full).
 * Needs a context with an own range at least two levels deep behind the sub-slabs; otherwise an error message. */
/* (A context may mix this entry point with the others, but its partition plan comes in two kinds — level 0 of its own
 *  making, or given by the sub-slabs — and every change of kind tears the plan down and rebuilds it: a stream
 *  synchronisation and the re-allocation of the partition buffers, tens of GB at hg38 scale; qm2_profile.replans counts
 *  them.  A caller that alternates per batch pays that per batch: keep one context per kind.) */
int qm2_probe_subslabs_device(qm2_ctx *c, const uint64_t *d_slabs, uint64_t sub_cap, uint32_t n_peers,
                              uint32_t n_subs, uint32_t peer_stride, const uint32_t *d_fill, uint64_t kmers_bound);

/* qm2_route_device with the text level's per-XCD sub-regions (EXPERIMENTS.md section 4.1): every one of the n_regions
 * (<= 1024) regions is cut into 8, one per XCD, so that a 128-byte line of the output only ever passes through one L2
 * (with hundreds of regions the plain layout writes a third of its lines as two masked pieces: 8.9 - 11.9 ms instead
 * of 7.4 per 20 M reads at C3).  Layout x-major: sub-region x of region r = region x * n_regions + r of d_slabs, each of
 * region_cap k-mers; d_fill has 8 * n_regions entries in the same order.  A receiver takes owner g's part of it as
 * qm2_probe_subslabs_device(ctx, d_slabs + g * n_subs * region_cap, region_cap, 8 * n_peers, n_subs,
 * peer_stride = n_regions, d_fill + g * n_subs, ...): the 8 sub-regions of a sender count as 8 peers. */
int qm2_route_xcd_device(qm2_ctx *c, const void *d_seq_bytes, size_t n, const uint64_t *d_suppress, uint64_t n_suppress,
                         uint64_t *d_slabs, uint64_t region_cap, uint32_t n_regions, uint32_t *d_fill, uint64_t *d_call);
/* ... for text that starts in the middle of a line (qm2_route_device_carry's contract).  ABI 6. */
int qm2_route_xcd_device_carry(qm2_ctx *c, const void *d_seq_bytes, size_t n, uint32_t carry, const uint64_t *d_suppress,
                               uint64_t n_suppress, uint64_t *d_slabs, uint64_t region_cap, uint32_t n_regions, uint32_t *d_fill,
                               uint64_t *d_call);

/* Probe + increment for an array of canonical k-mers already on this device
 * (QuicKmer.c:283-293: the worker's batch loop).  d_n (device u64, may be
 * NULL) overrides n with a device-side count (avoids a host round trip after
 * the exchange). */
int qm2_probe_device(qm2_ctx *c, const uint64_t *d_kmers, uint64_t n,
                     const uint64_t *d_n);

/* ---- epilogue on the device -------------------------------- *
 * QuicKmer.c:494-516 walks the genome-order chain (uint32 next[H], cyclic from
 * first_idx) one dependent load at a time: ~0.45 us per dictionary k-mer, ~15
 * minutes for hg38, long after the counting itself has become seconds.  Here the
 * walk is a parallel list ranking: every slot whose index is a multiple of R (and
 * first_idx) is a splitter; one thread per splitter walks to the next splitter and
 * records the length; the short list of splitters is ranked on the host; a second
 * walk writes counters[slot] (truncated to uint16) at its rank.
 *
 * qm2_chain_gather uploads `chain` (H entries, the .qm at offset 24 + 8H,
 * QuicKmer.c:483) and leaves the result on the device; qm2_chain_fetch copies a
 * window of it out (so a caller can stream <Out_prefix>.bin in pieces).  As in
 * the reference (:483, :490) the chain overwrites the context's key table when the
 * context owns it: after qm2_chain_gather the context cannot count any more. */
int qm2_chain_gather(qm2_ctx *c, const uint32_t *chain, uint64_t first_idx, uint64_t *n_out);
int qm2_chain_fetch(qm2_ctx *c, uint16_t *out, uint64_t offset, uint64_t count);

/* The input line discipline on the device (QuicKmer.c:393-398, :451-455):
 * `d_raw` = n bytes of RAW FASTA / FASTQ text on the device, starting at a line start,
 * 16-byte aligned.  fastq = the format the first line of the file decided (:395);
 * skip (FASTQ only, 0..3) = lines to drop before the first sequence line of this block
 * (1 for a block that starts at an '@' header).  Headers, '+' lines and qualities are
 * blanked on the device and the rest is counted like qm2_submit_device.  The reference's corner
 * cases are reproduced on the device too: the uint16 run-counter wrap of long lines (:402) and the
 * FASTQ sequence-position line that starts with '>' (:398 skips it without the three discards, so
 * every later sequence line comes one line later).  Only absurd amounts of either are reported in
 * *verdict instead (nothing of the block is counted then; the caller strips it on the host and
 * uses qm2_submit):
 *   QM2_RAW_DESYNC     more than 64 such '>' lines in one block
 *   QM2_RAW_MANY_WRAPS more than 2^20 places in one block where the run counter wraps
 * The call waits for the classification (not for the counting). */
#define QM2_RAW_DESYNC    1u
#define QM2_RAW_MANY_WRAPS 2u
int qm2_submit_raw_device(qm2_ctx *c, const void *d_raw, size_t n, uint32_t fastq, uint32_t skip,
                          uint32_t *verdict);

/* The same for host input, through the pinned staging buffers: the caller fills a buffer from
 * qm2_staging_acquire with raw text (the CLI: pread, nothing else) and commits the n bytes at
 * `offset` in it — whole lines, in file order, with the line state at their first byte.  Blocks
 * collect in a device batch; the line state of a batch is that of its first block (the device
 * numbers the lines itself), so consecutive blocks must be consecutive text.  A refusal
 * (see qm2_submit_raw_device) is sticky: qm2_raw_verdict() != 0 after qm2_sync / qm2_finish
 * means that raw batch and everything committed raw after it were NOT counted — the caller
 * starts over (qm2_reset) on the host path.  Do not mix with qm2_submit inside one batch. */
int qm2_staging_commit_raw(qm2_ctx *c, char *buf, size_t offset, size_t n, uint32_t fastq, uint32_t skip);
uint32_t qm2_raw_verdict(qm2_ctx *c);

/* ---- packed sequence (ABI 6): 3 bits a base over the PCIe link instead of 8 ---------------------------------- *
 * The counting kernels turn every 16 ASCII bytes into a u32 of 2-bit codes and a 16-bit reset mask before anything
 * else (QuicKmer.c:403-411: `c == 'N'`, the line end, `(c >> 1) & 3`).  The host may do exactly that — and no more —
 * in the pass it makes over the text anyway (the memcpy into a pinned buffer, the reader threads' block pass), so
 * that 0.375 bytes per base cross the link instead of 1 (FASTQ: instead of 2.05).  Rolling the k-mer, the reverse
 * complement, the canonical choice, hashing, probing and counting stay on the device.
 *
 * Format: a stream of POSITIONS; every sequence character and every line end is one position, in text order (the
 * text being what qm2_submit takes: whole '\n'-terminated sequence lines).  64 positions make one block of
 * QM2_PACKED_BLOCK_BYTES = 24 bytes, three little-endian u64:
 *     word 0, 1   the codes of positions 0..31 / 32..63: position q of the word has its code's HIGH bit
 *                 ((c >> 2) & 1) in bit 2q and its LOW bit ((c >> 1) & 1) in bit 2q + 1 (one v_bfrev_b32 on the
 *                 device puts a lane's 16 positions first-base-first)
 *     word 2      bit p set: position p is a reset — 'N', a line end, or filler behind the last line
 * A piece (one call) is whole blocks; its last block is filled up with resets, which behave like empty lines
 * (zero iterations of :403).  n_newlines = line ends + filler positions: n_bases = 64 n_blocks - n_newlines.
 * wraps: positions (from the piece's first) where the reference's uint16 run counter wraps (:402; lines of
 * >= 65536 bases only), ascending.
 *
 * qm2_pack_lines   host only, no device: n bytes of sequence lines (qm2_submit's contract) -> blocks at dst
 *                  (8-byte aligned, qm2_packed_bound(n) bytes suffice).  Thread-safe; callers pack pieces on as many
 *                  threads as they like and commit them in order.
 * qm2_pack_reads   the same with the reference's FASTA/FASTQ line discipline in front (:393-398, :451-455): `text`
 *                  is whole RAW lines (a last one without '\n' gets one), *st the state at its first line —
 *                  fastq (the first line of the file started with '@', :395; that line is NOT part of text),
 *                  skip (lines still to drop, 0..3), desync (out: set when a FASTQ sequence-position line started
 *                  with '>': :398 skips it without the three discards, so the state of the blocks behind it is
 *                  no longer a function of their line count) — updated to the state behind the last line.
 * Both return 0, 1 (bad arguments; message printed) or 2 (dst or wraps too small: nothing is printed, the caller
 * takes the ASCII path for this piece).
 * qm2_staging_commit_packed  commits the p->n_blocks blocks at the start of a buffer from qm2_staging_acquire
 *                  (wraps: host array of p->n_wraps positions or NULL) under qm2_staging_commit's rules; pieces
 *                  collect in a device batch of up to batch_bytes positions; a piece never straddles batches.
 * qm2_submit_packed_device   the same for blocks already in device memory (16-byte aligned; d_wraps on the device).
 * qm2_submit itself packs into the pinned buffers it copies through (QM2_FLAG_SUBMIT_ASCII turns that off). */
#define QM2_PACKED_BLOCK_BYTES 24
typedef struct {
    uint64_t n_blocks;
    uint64_t n_newlines;
    uint64_t n_wraps;
} qm2_packed;
typedef struct {
    uint32_t fastq, skip, desync;
    uint32_t lines;                  /* out: grows by the number of lines of the text (mod 2^32) — whatever `skip` was on entry, so a
                                        reader that packs a block under a GUESSED state still learns the state of the block behind it */
} qm2_line_state;
size_t qm2_packed_bound(size_t n_text);
int qm2_pack_lines(const char *lines, size_t n, void *dst, size_t dst_bytes, uint64_t *wraps, size_t wraps_cap,
                   qm2_packed *out);
int qm2_pack_reads(const char *text, size_t n, qm2_line_state *st, void *dst, size_t dst_bytes, uint64_t *wraps,
                   size_t wraps_cap, qm2_packed *out);
const char *qm2_pack_isa(void);      /* "avx512vbmi2" / "avx512" / "avx2" / "portable": the build of the packers this CPU runs */
const char *qm2_pack_isa_select(const char *name);   /* force a build (tests run all of them; NULL = the widest again); returns what runs */
int qm2_staging_commit_packed(qm2_ctx *c, char *buf, const qm2_packed *p, const uint64_t *wraps);
/* qm2_submit with the packing spread over n_threads host threads (0 = qm2_submit): pieces of whole lines, packed into
 * the pinned staging buffers by the threads, committed in order by the calling thread.  Same contract, same result. */
int qm2_submit_threads(qm2_ctx *c, const char *seq_bytes, size_t n, uint32_t n_threads);
int qm2_submit_packed_device(qm2_ctx *c, const void *d_blocks, const qm2_packed *p, const uint64_t *d_wraps);

/* Straight from the .qm file to the device (QuicKmer.c:353-359 and :483 without the 32 GiB
 * host array the reference mallocs at :353): the key table / the chain are read from `fd` at
 * `file_offset` by several threads into pinned buffers and copied up as they arrive.
 *   qm2_keys_from_fd     fills the key table of a context opened with QM2_FLAG_KEYS_LATER
 *                        (H x u64, the .qm at offset 24); counting before it is an error.
 *   qm2_chain_gather_fd  = qm2_chain_gather with the chain (H x u32, offset 24 + 8H) read
 *                        from the file.
 * A file that ends early is an error ("... truncated"). */
int qm2_keys_from_fd(qm2_ctx *c, int fd, uint64_t file_offset);
int qm2_chain_gather_fd(qm2_ctx *c, int fd, uint64_t file_offset, uint64_t first_idx, uint64_t *n_out);

/* The sums of the GC curve (QuicKmer.c:504-509) over the gathered counts, on the device: the .qgc
 * (one u16 per dictionary k-mer in chain order: bit 15 = control region, bits 0..8 = GC bin) is read
 * from `fd` (offset 0) like the two arrays above, and for every control-region entry e with bin b < 401
 *   cnt[b] += 1,  sum[b] += depth[e],  sumsq[b] += (int32)(depth[e] * depth[e])
 * — exact 64-bit integers (the reference adds the same numbers into doubles; equal while every partial
 * sum stays below 2^53), the square wrapped to int32 as the reference's `int` product does (:507).
 * n_file = entries the file holds; when it is shorter than the gathered counts, entry e >= n_file takes
 * stale[e mod 2^20] (2^20 values: what the reference's buffer still holds behind the end of the file,
 * :502-503 never looks at fread's return; the caller works them out).  sum, sumsq, cnt: 401 each.
 * At hg38 scale (2.27 G entries) this replaces 5 s of host thread time by a 4.5 GB upload. */
#define QM2_GC_BINS 401
int qm2_chain_gc_sums_fd(qm2_ctx *c, int fd, uint64_t n_file, const uint16_t *stale,
                         int64_t *sum, int64_t *sumsq, uint64_t *cnt);

/* Pure host helper (no device needed): the slot range shard g of n_shards
 * OWNS, [*own_lo, *own_hi) = [g*H/G, (g+1)*H/G), and the range it must hold,
 * [*span_lo, *span_hi): the owned range extended on its middle-facing side
 * (probes walk towards the middle of the table, QuicKmer.c:93-95) up to and
 * including the first empty slot, so that every walk that starts in the owned
 * range ends inside the span. */
int qm2_shard_span(const uint64_t *keys, uint64_t hash_size, uint32_t n_shards,
                   uint32_t g, uint64_t *own_lo, uint64_t *own_hi,
                   uint64_t *span_lo, uint64_t *span_hi);

/* Restrict the CALLING thread (and every thread it starts afterwards) to the CPUs of the NUMA node that HIP device
 * `device` hangs off (its PCI function's numa_node in sysfs): the threads that read, pack and stage the input then touch
 * memory — the page cache copy, the pinned staging buffers — on the socket whose PCIe root the DMA leaves through.  A
 * host-side placement hint and nothing else: returns 0 and changes nothing where the topology cannot be read (no sysfs,
 * one node, a device without affinity); QM2_NO_BIND=1 in the environment turns it into a no-op.  *n_cpus (may be NULL)
 * receives how many CPUs the thread may run on afterwards.  ABI 6. */
int qm2_bind_near_device(int device, uint32_t *n_cpus);

const char *qm2_last_error(void);
uint32_t    qm2_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QM2_H */
