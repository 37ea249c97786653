/*
 * qm2_partition.hip.h — the partitioned count pipeline (gfx950).
 *
 * Why: the direct kernel (qm2_kernels.hip.h) is bound by one memory-side atomic
 * per hit (~20 G/s measured on MI355X for scattered 4-byte atomics) and one
 * 64-byte sector per 8-byte probe.  Here the k-mers of a batch are first
 * radix-partitioned by the high bits of their HOME SLOT (DJB & (H-1),
 * QuicKmer.c:92) into buckets of S = 2^lgS consecutive table slots; then one
 * workgroup per bucket streams that slice of the table (keys + a zeroed counter
 * tile) into LDS, probes and counts its k-mers entirely in LDS (Find_hash,
 * :90-99, and the increment, :291), and flushes the tile with coalesced adds.
 * Random HBM gathers and per-hit global atomics become sequential streams:
 *
 *   split_text_kernel   text -> canonical k-mers, scattered into 2^b1 level-1 buckets
 *   split_kmer_kernel   level l-1 bucket -> its 2^bl children          (0..2 times)
 *   bucket_count_kernel finest bucket + table slice -> counters
 *   probe_kernel        (direct path) for the rare spill: bucket overflow (skewed
 *                       input, e.g. poly-A) or a probe walk longer than the halo
 *
 * Exactness: every emitted k-mer lands in exactly one finest bucket or in the
 * spill list, and is probed exactly once under the reference's rule; integer
 * adds commute, so the order inside buckets is irrelevant.
 */
#pragma once
#include "qm2_device.hip.h"

namespace qm2 {

constexpr int kSplitThreads  = 1024;                 /* 16 waves, one workgroup per CU          */
constexpr int kSplitWaves    = kSplitThreads / kWave;
constexpr int kSplitSlots    = kSplitThreads * 16;   /* k-mers staged in LDS per chunk (128 KiB) */
constexpr int kSplitTextBytes = kSplitWaves * kTileBytes;   /* 15872 sequence bytes per chunk   */
constexpr int kMaxFan        = 256;                  /* children per split                      */
constexpr int kHalo          = 256;                  /* table slots staged past a bucket's
                                                        middle-facing edge                      */
constexpr int kCountThreads  = 512;

struct SplitDst {
    uint64_t           *kmers;      /* child c occupies [c*cap, (c+1)*cap)                      */
    uint32_t           *cursors;    /* per child: k-mers reserved so far (may exceed cap)        */
    uint32_t            cap;
    uint64_t           *spill;      /* k-mers that did not fit: probed by the direct kernel      */
    unsigned long long *spill_count;
    uint64_t            spill_cap;
};

struct SplitLds {
    uint64_t sorted[kSplitSlots];
    uint32_t hist[kMaxFan];
    uint32_t off[kMaxFan];          /* exclusive prefix of hist                                  */
    uint32_t gbase[kMaxFan];        /* reserved start in the child's region                      */
    uint32_t fit[kMaxFan];          /* how many of this chunk's k-mers fit the child             */
    unsigned long long sbase[kMaxFan];   /* reserved start in the spill list for the rest        */
    uint32_t wsum[kMaxFan / kWave];
};

/* Steps 2-5 of a split chunk: every thread holds up to 16 k-mers km[j] with
 * tag[j] = (child << 16) | rank_in_child (rank from an LDS atomic on hist), or
 * 0xFFFFFFFF when slot j is empty.  Sorts the chunk by child in LDS and copies
 * each child's run to its region with full-wave contiguous stores. */
__device__ __forceinline__ void split_commit(SplitLds &L, const uint64_t (&km)[16], const uint32_t (&tag)[16],
                                             uint64_t child0, uint32_t fan, const SplitDst &d,
                                             unsigned long long &lost)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    __syncthreads();                                           /* hist complete */
    /* exclusive scan of hist[0..fan) by the first fan threads (fan <= 256 = 4 waves) */
    uint32_t mine = 0, incl = 0;
    if (tid < kMaxFan) {
        mine = tid < (int)fan ? L.hist[tid] : 0;
        incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) L.wsum[wave] = incl;
    }
    __syncthreads();
    if (tid < kMaxFan) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += L.wsum[w];
        L.off[tid] = before + incl - mine;
        if (tid < (int)fan && mine) {
            const uint64_t child = child0 + (uint64_t)tid;
            const uint32_t g = atomicAdd(&d.cursors[child], mine);
            const uint32_t f = g >= d.cap ? 0u : (d.cap - g < mine ? d.cap - g : mine);
            L.gbase[tid] = g;
            L.fit[tid] = f;
            if (f < mine) L.sbase[tid] = atomicAdd(d.spill_count, (unsigned long long)(mine - f));
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (tag[j] != 0xFFFFFFFFu) L.sorted[L.off[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = km[j];
    __syncthreads();
    /* run copy: wave w takes children w, w+16, ... */
    for (uint32_t b = wave; b < fan; b += kSplitWaves) {
        const uint32_t n = L.hist[b];
        if (n == 0) continue;
        const uint32_t o = L.off[b], f = L.fit[b];
        uint64_t *dst = d.kmers + (child0 + b) * (uint64_t)d.cap + L.gbase[b];
        for (uint32_t i = lane; i < f; i += kWave) dst[i] = L.sorted[o + i];
        if (f < n) {
            const unsigned long long sb = L.sbase[b];
            for (uint32_t i = f + lane; i < n; i += kWave) {
                const unsigned long long at = sb + (i - f);
                if (at < d.spill_cap) d.spill[at] = L.sorted[o + i];
                else lost++;
            }
        }
    }
    __syncthreads();                                           /* LDS reusable */
}

/* Level 1: sequence text -> canonical k-mers -> 2^b1 buckets by the top bits of
 * the home slot.  One chunk = 16 tiles (one per wave). */
template <bool K30>
__global__ __launch_bounds__(kSplitThreads) void split_text_kernel(SeqView sv, uint32_t k, uint32_t hmask,
                                                                   uint32_t shift, uint32_t fan,
                                                                   uint64_t n_chunks, SplitDst d, DevStats *stats)
{
    __shared__ SplitLds L;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    unsigned long long n_bases = 0, n_kmers = 0, lost = 0;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        if (tid < kMaxFan) L.hist[tid] = 0;
        __syncthreads();
        const uint64_t tile = chunk * kSplitWaves + wave;
        const Window W = make_window<K30>(sv, tile * kTileBytes, lane, k);
        n_bases += W.n_bases;
        n_kmers += __builtin_popcount(W.emit);
        uint64_t km[16];
        uint32_t tag[16];
#define QM2_T(J)                                                                   \
        {                                                                          \
            km[J] = kmer_at<K30, J>(W, k);                                         \
            tag[J] = 0xFFFFFFFFu;                                                  \
            if (W.emit & (0x8000u >> J)) {                                         \
                const uint32_t b = fan > 1 ? (djb32(km[J]) & hmask) >> shift : 0u; \
                tag[J] = (b << 16) | atomicAdd(&L.hist[b], 1u);                    \
            }                                                                      \
        }
        QM2_T(0) QM2_T(1) QM2_T(2) QM2_T(3) QM2_T(4) QM2_T(5) QM2_T(6) QM2_T(7)
        QM2_T(8) QM2_T(9) QM2_T(10) QM2_T(11) QM2_T(12) QM2_T(13) QM2_T(14) QM2_T(15)
#undef QM2_T
        split_commit(L, km, tag, 0, fan, d, lost);
    }
    n_bases = wave_sum(n_bases); n_kmers = wave_sum(n_kmers); lost = wave_sum(lost);
    if (lane == 0) {
        atomicAdd(&stats->n_bases, n_bases);
        atomicAdd(&stats->n_kmers, n_kmers);
        if (lost) atomicAdd(&stats->part_lost, lost);
    }
}

/* Level l >= 2: split every level l-1 bucket into 2^bl children.
 * grid.y = parent, grid.x = chunk of kSplitSlots k-mers inside the parent. */
__global__ __launch_bounds__(kSplitThreads) void split_kmer_kernel(const uint64_t *src, const uint32_t *src_cursors,
                                                                   uint32_t src_cap, uint32_t hmask, uint32_t shift,
                                                                   uint32_t fan, SplitDst d, DevStats *stats)
{
    __shared__ SplitLds L;
    const uint64_t parent = blockIdx.y;
    uint32_t n = src_cursors[parent];
    if (n > src_cap) n = src_cap;
    const uint64_t first = (uint64_t)blockIdx.x * kSplitSlots;
    if (first >= n) return;                                    /* whole workgroup leaves together */
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    unsigned long long lost = 0;
    if (tid < kMaxFan) L.hist[tid] = 0;
    __syncthreads();
    const uint64_t *in = src + parent * (uint64_t)src_cap;
    uint64_t km[16];
    uint32_t tag[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t at = first + (uint64_t)j * kSplitThreads + tid;
        km[j] = 0;
        tag[j] = 0xFFFFFFFFu;
        if (at < n) {
            km[j] = in[at];
            const uint32_t b = ((djb32(km[j]) & hmask) >> shift) & (fan - 1);
            tag[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
        }
    }
    split_commit(L, km, tag, parent * fan, fan, d, lost);
    lost = wave_sum(lost);
    if (lane == 0 && lost) atomicAdd(&stats->part_lost, lost);
}

/* Finest level: one workgroup per bucket of S = 2^LG_S table slots. */
template <int LG_S>
__global__ __launch_bounds__(kCountThreads) void bucket_count_kernel(const uint64_t *kmers, const uint32_t *cursors,
                                                                     uint32_t cap, TableView tv, uint64_t H,
                                                                     uint32_t n_buckets, DevStats *stats)
{
    if (stats->part_lost != 0) return;       /* this batch is recounted by the direct kernel */
    constexpr int S = 1 << LG_S;
    __shared__ uint64_t skey[S + kHalo];
    __shared__ uint32_t scnt[S + kHalo];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    unsigned long long n_probes = 0, n_hits = 0;

    for (uint32_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        uint32_t n = cursors[b];
        if (n > cap) n = cap;
        if (n == 0) continue;                                  /* uniform for the workgroup */
        /* the slice of the table this bucket can touch: its own S slots plus kHalo
         * slots on the side its probe walks leave through (towards the middle, :94-95) */
        const uint64_t own_lo = (uint64_t)b << LG_S;
        uint64_t own_hi = own_lo + S;
        if (own_hi > H) own_hi = H;
        const bool down = (own_lo & (H >> 1)) != 0;            /* upper half walks down */
        uint64_t seg_lo = own_lo, seg_hi = own_hi;
        if (H > (uint64_t)S) {
            if (down) seg_lo = own_lo >= (uint64_t)kHalo ? own_lo - kHalo : 0;
            else      seg_hi = own_hi + kHalo <= H ? own_hi + kHalo : H;
        }
        const uint32_t seg = (uint32_t)(seg_hi - seg_lo);
        for (uint32_t i = tid; i < seg; i += kCountThreads) {
            skey[i] = tv.keys[seg_lo + i];
            scnt[i] = 0;
        }
        __syncthreads();
        const uint64_t *in = kmers + (uint64_t)b * cap;
        for (uint32_t i = tid; i < n; i += kCountThreads) {
            const uint64_t km = in[i];
            const uint32_t home = djb32(km) & tv.hmask;                    /* :92 */
            /* H <= S: the single bucket holds both halves, direction per k-mer */
            const bool dn = H > (uint64_t)S ? down : (home & tv.hhalf) != 0;
            int32_t at = (int32_t)((uint64_t)home - seg_lo);
            uint32_t seen = 0;
            for (;;) {
                if (at < 0 || at >= (int32_t)seg) {
                    /* the walk left the staged slice (a run of occupied slots longer than
                     * kHalo): finish it in global memory, like the direct kernel */
                    uint64_t slot = seg_lo + (int64_t)at;
                    for (;;) {
                        const uint64_t g = tv.keys[slot];
                        seen++;
                        if (g == km) { atomicAdd(&tv.counters[slot], 1u); n_hits++; break; }
                        if (g == 0) break;
                        slot += dn ? (uint64_t)-1 : 1;
                    }
                    break;
                }
                const uint64_t v = skey[at];
                seen++;
                if (v == km) { atomicAdd(&scnt[at], 1u); n_hits++; break; }   /* :98, :291 */
                if (v == 0) break;
                at += dn ? -1 : 1;                                         /* :97 */
            }
            n_probes += seen;
        }
        __syncthreads();
        /* flush: the interior of the owned range is touched by this workgroup only
         * (plain add); the kHalo slots at either end of it and the staged halo can
         * also receive a neighbour's counts (atomic add). */
        for (uint32_t i = tid; i < seg; i += kCountThreads) {
            const uint32_t c = scnt[i];
            if (c == 0) continue;
            const uint64_t slot = seg_lo + i;
            const bool interior = slot >= own_lo + kHalo && slot + kHalo < own_hi;
            uint32_t *p = tv.counters + (slot - tv.span_lo);
            if (interior) *p += c; else atomicAdd(p, c);
        }
        __syncthreads();
    }
    n_probes = wave_sum(n_probes); n_hits = wave_sum(n_hits);
    if (lane == 0) {
        atomicAdd(&stats->n_probes, n_probes);
        atomicAdd(&stats->n_hits, n_hits);
    }
}

/* Start of a partitioned batch: account the previous batch's fallback, clear the flag. */
__global__ void part_begin_kernel(DevStats *stats, unsigned long long *spill_count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stats->part_lost = 0;
        *spill_count = 0;
    }
}

__global__ void part_end_kernel(DevStats *stats)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && stats->part_lost != 0) stats->fallbacks++;
}

}  // namespace qm2
