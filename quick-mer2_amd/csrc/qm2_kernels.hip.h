/*
 * qm2_kernels.hip.h — the gfx950 kernels of the `count` hot path.
 *
 *   count_kernel    fused encode + probe + increment (the product path)
 *   encode_kernel   encode + bucket by owner shard   (hash-range sharding)
 *   probe_kernel    probe + increment of ready k-mers (hash-range sharding;
 *                   the reference's Kmer_count_TSK batch loop, :283-293)
 *   narrow_kernel   u32 counters -> u16 (the reference's counter type, :23)
 *
 * All are HBM-latency/bandwidth bound integer kernels; there is no contraction
 * anywhere on this path, so no MFMA.  Algorithmic bytes per launch:
 *   1 B per sequence byte + 8 B per slot inspected + 4 B per hit.
 */
#pragma once
#include "qm2_device.hip.h"

namespace qm2 {

constexpr int kNJ    = 16;             /* k-mers per lane per tile               */

/* One probe round over the lane's k-mers.  slot[] holds the key last loaded
 * for each pending k-mer.  Returns the still-pending mask. */
#define QM2_RESOLVE_ROUND()                                                        \
    _Pragma("unroll") for (int j = 0; j < kNJ; j++) {                              \
        const uint32_t bit = 1u << j;                                              \
        if (pending & bit) {                                                       \
            const uint64_t s = slot[j];                                            \
            if (s == km[j]) {            /* match (k-mer 0 matches an empty slot, \
                                            exactly like :96-98)                */ \
                atomicAdd(&tv.counters[idx[j]], 1u);                               \
                n_hits++;                                                          \
                pending &= ~bit;                                                   \
            } else if (s == 0) {         /* empty: not in the dictionary        */ \
                pending &= ~bit;                                                   \
            } else {                     /* occupied by another key: next slot  */ \
                idx[j] += (dirs & bit) ? 0xFFFFFFFFu : 1u;                                   \
                slot[j] = tv.keys[idx[j]];                                         \
                n_probes++;                                                        \
            }                                                                      \
        }                                                                          \
    }

template <bool K30, bool PK = false>
__global__ __launch_bounds__(kBlock) void count_kernel(SeqView sv, TableView tv, uint32_t k,
                                                       uint64_t n_tiles, DevStats *stats,
                                                       const unsigned long long *only_if, bool seq_stats)
{
    /* as the partitioned pipeline's fallback this launch runs only if the batch lost k-mers */
    if (only_if && *only_if == 0) return;
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave_id = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) / kWave;

    unsigned long long n_bases = 0, n_kmers = 0, n_probes = 0, n_hits = 0;

    for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const Window W = make_window<K30, PK>(sv, tile * kTileBytes, lane, k);
        n_bases += W.n_bases;
        n_kmers += __popc(W.emit);

        uint64_t km[kNJ];
        uint64_t slot[kNJ];
        uint32_t idx[kNJ];          /* index into tv.keys (span-relative)        */
        uint32_t pending = 0, dirs = 0;

        /* issue the first probe of every k-mer of this lane back to back */
#define QM2_FIRST(J)                                                               \
        {                                                                          \
            km[J] = kmer_at<K30, J>(W, k);                                         \
            const uint32_t home = djb32(km[J]) & tv.hmask;       /* :92 */         \
            if (W.emit & (0x8000u >> J)) {                                         \
                pending |= 1u << J;                                                \
                if (home & tv.hhalf) dirs |= 1u << J;            /* :94 */         \
                idx[J] = home - (uint32_t)tv.span_lo;                      \
                slot[J] = tv.keys[idx[J]];                                         \
                n_probes++;                                                        \
            } else {                                                               \
                idx[J] = 0; slot[J] = 0;                                           \
            }                                                                      \
        }
        QM2_FIRST(0)  QM2_FIRST(1)  QM2_FIRST(2)  QM2_FIRST(3)
        QM2_FIRST(4)  QM2_FIRST(5)  QM2_FIRST(6)  QM2_FIRST(7)
        QM2_FIRST(8)  QM2_FIRST(9)  QM2_FIRST(10) QM2_FIRST(11)
        QM2_FIRST(12) QM2_FIRST(13) QM2_FIRST(14) QM2_FIRST(15)
#undef QM2_FIRST

        while (pending) { QM2_RESOLVE_ROUND() }
    }

    if (stats) {
        n_bases = wave_sum(n_bases);  n_kmers = wave_sum(n_kmers);
        n_probes = wave_sum(n_probes); n_hits = wave_sum(n_hits);
        if (lane == 0) {
            if (seq_stats) {
                atomicAdd(&stats->n_bases, n_bases);
                atomicAdd(&stats->n_kmers, n_kmers);
            }
            atomicAdd(&stats->n_probes, n_probes);
            atomicAdd(&stats->n_hits, n_hits);
        }
    }
}

/* ---- split path ------------------------------------------------------------ */

/* Encode only: canonical k-mers appended to per-owner buckets.
 * owner = home_slot >> owner_shift. */
template <bool K30>
__global__ __launch_bounds__(kBlock) void encode_kernel(SeqView sv, uint32_t k, uint32_t hmask,
                                                        uint64_t n_tiles, uint64_t *out,
                                                        uint64_t bucket_cap, uint32_t n_buckets,
                                                        uint32_t owner_shift,
                                                        unsigned long long *counts, DevStats *stats)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave_id = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) / kWave;
    unsigned long long n_bases = 0, n_kmers = 0, dropped = 0;

    for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const Window W = make_window<K30>(sv, tile * kTileBytes, lane, k);
        n_bases += W.n_bases;
        n_kmers += __popc(W.emit);
#define QM2_EMIT(J)                                                                \
        {                                                                          \
            const uint64_t km = kmer_at<K30, J>(W, k);                             \
            const bool on = (W.emit & (0x8000u >> J)) != 0;                        \
            const uint32_t owner = (n_buckets > 1) ? ((djb32(km) & hmask) >> owner_shift) : 0u; \
            for (uint32_t b = 0; b < n_buckets; b++) {                             \
                const unsigned long long m = __ballot(on && owner == b);           \
                if (m == 0) continue;                                              \
                unsigned long long base = 0;                                       \
                const int leader = __ffsll((long long)m) - 1;                      \
                if (lane == leader) base = atomicAdd(&counts[b], (unsigned long long)__popcll(m)); \
                base = __shfl(base, leader);                                       \
                if (on && owner == b) {                                            \
                    const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1)); \
                    if (at < bucket_cap) out[(uint64_t)b * bucket_cap + at] = km;  \
                    else dropped++;                                                \
                }                                                                  \
            }                                                                      \
        }
        QM2_EMIT(0)  QM2_EMIT(1)  QM2_EMIT(2)  QM2_EMIT(3)
        QM2_EMIT(4)  QM2_EMIT(5)  QM2_EMIT(6)  QM2_EMIT(7)
        QM2_EMIT(8)  QM2_EMIT(9)  QM2_EMIT(10) QM2_EMIT(11)
        QM2_EMIT(12) QM2_EMIT(13) QM2_EMIT(14) QM2_EMIT(15)
#undef QM2_EMIT
    }
    if (stats) {
        n_bases = wave_sum(n_bases); n_kmers = wave_sum(n_kmers); dropped = wave_sum(dropped);
        if (lane == 0) {
            atomicAdd(&stats->n_bases, n_bases);
            atomicAdd(&stats->n_kmers, n_kmers);
            if (dropped) atomicAdd(&stats->overflow, dropped);
        }
    }
}

/* Probe + increment of ready-made canonical k-mers: lane handles kNJ k-mers
 * strided by the wave so that the k-mer loads are coalesced. */
/* skip_if / only_if: device-side conditions (the partitioned pipeline's "this batch lost k-mers" flag);
 * d_n / d_n32: the count lives on the device (n is then the capacity) */
__global__ __launch_bounds__(kBlock) void probe_kernel(const uint64_t *kmers, uint64_t n,
                                                       const unsigned long long *d_n, const uint32_t *d_n32,
                                                       TableView tv, DevStats *stats,
                                                       const unsigned long long *skip_if,
                                                       const unsigned long long *only_if)
{
    if (skip_if && *skip_if != 0) return;
    if (only_if && *only_if == 0) return;
    if (d_n) n = *d_n < n ? *d_n : n;
    if (d_n32) n = *d_n32 < n ? *d_n32 : n;
    const int lane = threadIdx.x & (kWave - 1);
    const uint64_t wave_id = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) / kWave;
    const uint64_t per_wave = (uint64_t)kWave * kNJ;
    const uint64_t n_groups = (n + per_wave - 1) / per_wave;
    unsigned long long n_probes = 0, n_hits = 0;
    uint32_t held_idx = 0xFFFFFFFFu, held = 0;          /* wave-uniform: increments not added yet, `held` of them on slot held_idx */

    for (uint64_t grp = wave_id; grp < n_groups; grp += n_waves) {
        uint64_t km[kNJ];
        uint64_t slot[kNJ];
        uint32_t idx[kNJ];
        uint32_t pending = 0, dirs = 0;
#pragma unroll
        for (int j = 0; j < kNJ; j++) {
            const uint64_t at = grp * per_wave + (uint64_t)j * kWave + lane;
            km[j] = 0; idx[j] = 0; slot[j] = 0;
            if (at < n && kmers[at] != ~0ull) {          /* ~0 = padding of the partitioned pipeline's spill */
                km[j] = kmers[at];
                const uint32_t home = djb32(km[j]) & tv.hmask;
                pending |= 1u << j;
                if (home & tv.hhalf) dirs |= 1u << j;
                idx[j] = home - (uint32_t)tv.span_lo;
                slot[j] = tv.keys[idx[j]];
                n_probes++;
            }
        }
        /* One probe round per pass, like QM2_RESOLVE_ROUND, but the wave keeps ONE slot whose increments it sums
         * instead of adding them one by one: the spill list holds the overflow of single buckets, i.e. floods of
         * one k-mer mixed with the ordinary k-mers of its bucket (poly-A: 30 M copies of k-mer 0 per 50 M reads at
         * 0.5 % poly-A reads; 30 M atomics on one counter took 340 ms).  A slot that 8 or more lanes hit in one
         * pass becomes the held one.  (Wave-uniform control flow around the ballots.) */
        while (__ballot(pending != 0)) {
#pragma unroll
            for (int j = 0; j < kNJ; j++) {
                const uint32_t bit = 1u << j;
                const bool on = (pending & bit) != 0;
                const uint64_t s = slot[j];
                const bool hit = on && s == km[j];              /* (k-mer 0 matches an empty slot, exactly like :96-98) */
                unsigned long long rest = __ballot(hit);
                if (rest) {
                    const unsigned long long mh = __ballot(hit && idx[j] == held_idx);
                    held += (uint32_t)__popcll(mh);
                    rest &= ~mh;
                    if (__popcll(rest) >= 8) {                  /* a new flood?  the slot of the first lane left */
                        const uint32_t cand = (uint32_t)__shfl((int)idx[j], __ffsll((long long)rest) - 1);
                        const unsigned long long mc = __ballot(hit && idx[j] == cand);
                        if (__popcll(mc) >= 8) {
                            if (held && lane == 0) atomicAdd(&tv.counters[held_idx], held);
                            held_idx = cand;
                            held = (uint32_t)__popcll(mc);
                            rest &= ~mc;
                        }
                    }
                    if ((rest >> lane) & 1ull) atomicAdd(&tv.counters[idx[j]], 1u);
                }
                if (hit) { n_hits++; pending &= ~bit; }
                else if (on && s == 0) pending &= ~bit;         /* empty: not in the dictionary */
                else if (on) {                                  /* occupied by another key: next slot */
                    idx[j] += (dirs & bit) ? 0xFFFFFFFFu : 1u;
                    slot[j] = tv.keys[idx[j]];
                    n_probes++;
                }
            }
        }
    }
    if (held && lane == 0) atomicAdd(&tv.counters[held_idx], held);
    if (stats) {
        n_probes = wave_sum(n_probes); n_hits = wave_sum(n_hits);
        /* (only waves that probed anything: the spill list is EMPTY on ordinary input, and 8192 idle waves adding
         *  zeros to the same two words were the whole 0.2 ms this launch took per batch) */
        if (lane == 0 && n_probes) {
            atomicAdd(&stats->n_probes, n_probes);
            if (n_hits) atomicAdd(&stats->n_hits, n_hits);
        }
    }
}

/* The same over many regions of one buffer (a shard's received sub-slabs: region r = kmers + r * cap, fills[r]
 * k-mers in it; peer p's sub-slab s is region p * peer_stride + s): only ever the exact fallback of the partitioned pipeline (only_if: the batch lost k-mers) or the
 * path of a table the partition cannot take, so plain — every k-mer probed and counted on its own. */
__global__ __launch_bounds__(kBlock) void probe_regions_kernel(const uint64_t *kmers, uint64_t cap, uint32_t n_peers, uint32_t n_subs,
                                                               uint32_t peer_stride, const uint32_t *fills, TableView tv, DevStats *stats,
                                                               const unsigned long long *only_if)
{
    if (only_if && *only_if == 0) return;
    const uint32_t n_regions = n_peers * n_subs;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x, n_threads = (uint64_t)gridDim.x * kBlock;
    const int lane = threadIdx.x & (kWave - 1);
    unsigned long long n_probes = 0, n_hits = 0;
    for (uint32_t q = 0; q < n_regions; q++) {
        const uint64_t r = (uint64_t)(q / n_subs) * peer_stride + q % n_subs;       /* sub-slab q % n_subs of peer q / n_subs */
        const uint64_t n = fills[r] < cap ? fills[r] : cap;
        const uint64_t *src = kmers + r * cap;
        for (uint64_t i = gid; i < n; i += n_threads) {
            const uint64_t km = src[i];
            if (km == ~0ull) continue;
            const uint32_t home = djb32(km) & tv.hmask;
            const uint32_t step = (home & tv.hhalf) ? 0xFFFFFFFFu : 1u;     /* :93-95 */
            uint32_t idx = home - (uint32_t)tv.span_lo;
            for (;;) {
                const uint64_t v = tv.keys[idx];
                n_probes++;
                if (v == km) { atomicAdd(&tv.counters[idx], 1u); n_hits++; break; }   /* :98, :291 */
                if (v == 0) break;
                idx += step;                                                 /* :97 */
            }
        }
    }
    n_probes = wave_sum(n_probes); n_hits = wave_sum(n_hits);
    if (lane == 0 && stats) {
        atomicAdd(&stats->n_probes, n_probes);
        atomicAdd(&stats->n_hits, n_hits);
    }
}

/* Packed input does not tell a line end from an 'N' (both are resets), so its kernels count every position as a base;
 * the host knows how many of a batch's positions are line ends or filler (qm2_packed.n_newlines) and takes them off. */
__global__ void stats_sub_bases_kernel(DevStats *stats, unsigned long long n)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->n_bases, 0ull - n);
}

/* u32 accumulators -> the reference's uint16 counters (mod 65536, :23/:291). */
__global__ __launch_bounds__(kBlock) void narrow_kernel(const uint32_t *in, uint16_t *out, uint64_t n)
{
    uint64_t i = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 8;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock * 8;
    for (; i + 8 <= n; i += stride) {
        const uint4 a = *reinterpret_cast<const uint4 *>(in + i);
        const uint4 b = *reinterpret_cast<const uint4 *>(in + i + 4);
        uint4 o;
        o.x = (a.x & 0xFFFFu) | (a.y << 16);
        o.y = (a.z & 0xFFFFu) | (a.w << 16);
        o.z = (b.x & 0xFFFFu) | (b.y << 16);
        o.w = (b.z & 0xFFFFu) | (b.w << 16);
        *reinterpret_cast<uint4 *>(out + i) = o;
    }
    if (i < n && i + 8 > n)
        for (uint64_t t = i; t < n; t++) out[t] = (uint16_t)in[t];
}

}  // namespace qm2
