/*
 * qm2_shards.h — BASELINE config 4 from the C host: the dictionary range-partitioned by the high bits of
 * the home slot over the GPUs of one node, the k-mers of every batch routed to their owners by an RCCL
 * all-to-all over xGMI, overlapped with the encode of the next batch (libqm2_shards.so = this driver +
 * librccl, on top of libqm2_hip.so; `quicKmer2 count --shards a,b,..` loads it on demand, so the plain CLI
 * does not depend on RCCL).
 *
 * What it replaces in the reference: the same region as include/qm2.h — main_count's counting phase
 * (QuicKmer.c:368-481) — with the reference's ONE shared table (Kmer_hash / Kmer_depth globals, :14-23, probed
 * by N worker threads, :256-296) cut into G = 2^g slices of consecutive slots.  Owner of a k-mer = the top g bits of
 * its home slot DJB & (H-1) (:92).  Probes walk towards the middle of the table (:93-95), so a shard also holds the
 * run of occupied slots past its middle-facing edge (qm2_shard_span); what is counted in that overhang is added
 * to the owning neighbour's slots at the end (integer adds commute: exact).
 *
 * One process, one host thread, one RCCL communicator per GPU (ncclCommInitAll): the single-process
 * multi-GPU form.  Per round, for every rank g (each on its own pair of HIP streams):
 *     route     qm2_route_device: text -> canonical k-mers -> G slabs by owner        (router stream)
 *     exchange  ncclGroupStart; ncclSend / ncclRecv of every slab's FILLED part and of the fills;
 *               ncclGroupEnd                                                         (router stream)
 *     probe     qm2_probe_slabs_device on what arrived: the shard's own partitioned pipeline (prober stream)
 * with two sets of exchange buffers used in turn, so that round i + 1 is encoded and sent while round i is probed.
 * The host looks at one word per rank and round (did every slab fit: skewed input such as poly-A sends everything
 * to one owner); a round that does not fit is cut at a line end and routed again — exact for any input.
 *
 * QM2_SHARDS_ROUTE = xcd | flat picks the exchange layout.  flat: the above.  xcd (the default for tables of 2^23 slots
 * and more): the senders route by 10 bits — G owners x 1024 / G sub-slabs, each in 8 per-XCD sub-regions of fixed
 * capacity (qm2_route_xcd_device) — an owner's part travels as 8 whole blocks per sender with the fills alongside, and
 * what a shard receives is the first level of its own partition (qm2_probe_subslabs_device: one split, not two).
 *
 * The text is fed like qm2_staging_acquire / qm2_staging_commit / qm2_submit: whole '\n'-terminated sequence lines
 * (the CLI's host line discipline), dealt to the ranks a batch at a time.
 */
#ifndef QM2_SHARDS_H
#define QM2_SHARDS_H

#include <stddef.h>
#include <stdint.h>

#include "qm2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qm2_shards qm2_shards;

/* dict_fd: the open .qm (header at 0, keys at 24: each rank streams its own span with qm2_keys_from_fd).
 * devices[n_devices]: HIP ordinals, n_devices a power of two; the same ordinal may only be listed twice with
 * QM2_SHARDS_FAKE_EXCHANGE=1 in the environment (tests on one GPU: the exchange is then device-to-device copies
 * instead of RCCL, everything else is the real path).  batch_bytes: text per rank and round (0 = 1.5 bytes per
 * slot of a shard, 1 .. 6 GiB). */
int qm2_shards_open(qm2_shards **out, int dict_fd, uint64_t hash_size, uint32_t k, const int *devices,
                    int n_devices, uint64_t batch_bytes, uint32_t n_staging);

int qm2_shards_acquire(qm2_shards *s, char **buf, size_t *capacity);
int qm2_shards_commit(qm2_shards *s, char *buf, size_t n);
int qm2_shards_submit(qm2_shards *s, const char *lines, size_t n);

/* Drain, then fold every shard's counters (own slots + overhang) into a whole-table context on devices[0]: the
 * job's counters, ready for qm2_finish / qm2_chain_gather_fd.  st (may be NULL): the job's totals. */
int qm2_shards_finish(qm2_shards *s, qm2_stats *st);
qm2_ctx *qm2_shards_result(qm2_shards *s);

/* what the run did: rounds, rounds that had to be cut and routed again, bytes that went over the wire */
typedef struct { uint64_t rounds, rerouted, exchanged_bytes; uint32_t ten_bit_route; uint32_t flat_fallbacks; } qm2_shards_info;
/* (ten_bit_route: 1 = QM2_SHARDS_ROUTE=xcd is in force; flat_fallbacks: pieces that overflowed its sub-regions however they
 *  were cut and went round with one slab per owner) */
int qm2_shards_get_info(qm2_shards *s, qm2_shards_info *info);

void qm2_shards_close(qm2_shards *s);

#ifdef __cplusplus
}
#endif
#endif
