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

#ifndef QM2_SPLIT_THREADS
#define QM2_SPLIT_THREADS 1024
#endif
#ifndef QM2_COUNT_THREADS
#define QM2_COUNT_THREADS 512
#endif
constexpr int kSplitThreads  = QM2_SPLIT_THREADS;    /* 1024: long runs per child beat occupancy (measured) */
constexpr int kSplitWaves    = kSplitThreads / kWave;
constexpr int kSplitSlots    = kSplitThreads * 16;   /* k-mers staged in LDS per chunk          */
constexpr int kSplitTextBytes = kSplitWaves * kTileBytes;   /* 15872 sequence bytes per chunk   */
constexpr int kMaxFan        = 256;                  /* children per split                      */
#ifndef QM2_RUN_ALIGN
#define QM2_RUN_ALIGN 1
#endif
constexpr int kRunAlign      = QM2_RUN_ALIGN;        /* 8 would pad every run a workgroup appends to a
                                                        child to whole 64-byte sectors (slack = kPad);
                                                        measured on MI355X: no gain (18.6 vs 18.8 ms per
                                                        C2 step), so runs are left unpadded            */
constexpr uint64_t kPad      = ~0ull;                /* never a k-mer (k <= 31 uses at most 62 bits)   */
constexpr int kHalo          = 256;                  /* table slots staged past a bucket's
                                                        middle-facing edge                      */
constexpr int kCountThreads  = QM2_COUNT_THREADS;

/* Workgroup barrier for LDS hand-offs.  hipcc's __syncthreads() also drains every outstanding
 * global load / store / atomic of the wave (s_waitcnt vmcnt(0)); these kernels keep a prefetch,
 * a cursor reservation and the previous chunk's copy-out stores in flight ACROSS their
 * barriers, so only the LDS traffic is waited for here.  (The compiler still inserts the vmcnt
 * wait in front of the first use of each loaded value.) */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct SplitDst {
    uint64_t           *kmers;      /* child c occupies [c*cap, (c+1)*cap)                      */
    uint32_t           *cursors;    /* per child: k-mers reserved so far (may exceed cap)        */
    uint32_t            cap;
    uint64_t           *spill;      /* k-mers that did not fit: probed by the direct kernel      */
    unsigned long long *spill_count;
    uint64_t            spill_cap;
};

struct SplitLds {
    alignas(16) uint64_t sorted[kSplitSlots + kMaxFan * (kRunAlign - 1)];
    uint32_t hist[kMaxFan];
    uint32_t off[kMaxFan];          /* exclusive prefix of the PADDED run lengths                */
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
    lds_barrier();                                           /* hist complete */
    /* exclusive scan of hist[0..fan) by the first fan threads (fan <= 256 = 4 waves) */
    uint32_t mine = 0, incl = 0;
    static_assert(kSplitThreads >= kMaxFan, "the scan uses one thread per child");
    if (tid < kMaxFan) {
        mine = tid < (int)fan ? L.hist[tid] : 0;
        mine = (mine + kRunAlign - 1) & ~(uint32_t)(kRunAlign - 1);        /* padded run length */
        incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) L.wsum[wave] = incl;
    }
    lds_barrier();
    uint32_t g = 0;
    const bool reserve = tid < (int)fan && mine != 0;
    if (tid < kMaxFan) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += L.wsum[w];
        L.off[tid] = before + incl - mine;
        if (reserve) {
            /* reserve the run in the child's region; the returned offset is only needed for the
             * copy-out, so the round trip overlaps the LDS scatter below */
            g = atomicAdd(&d.cursors[child0 + (uint64_t)tid], mine);
            const uint32_t real = L.hist[tid];                 /* pad the tail of the run (kRunAlign > 1) */
            for (uint32_t i = real; i < mine; i++) L.sorted[before + incl - mine + i] = kPad;
        }
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (tag[j] != 0xFFFFFFFFu) L.sorted[L.off[tag[j] >> 16] + (tag[j] & 0xFFFFu)] = km[j];
    if (reserve) {
        const uint32_t f = g >= d.cap ? 0u : (d.cap - g < mine ? d.cap - g : mine);
        L.gbase[tid] = g;
        L.fit[tid] = f;                                        /* cap is a multiple of kRunAlign, so is f */
        if (f < mine) L.sbase[tid] = atomicAdd(d.spill_count, (unsigned long long)(mine - f));
    }
    lds_barrier();
    /* run copy: wave w takes children w, w + nwaves, ...  Their run descriptors are fetched
     * once, one per lane, and broadcast with shuffles, so the loop below has no dependent LDS
     * look-ups in front of its copies. */
    static_assert(kMaxFan <= kSplitWaves * kWave, "one descriptor per lane");
    uint32_t m_n = 0, m_o = 0, m_f = 0, m_g = 0;
    {
        const uint32_t mb = wave + lane * kSplitWaves;
        if (mb < fan) {
            m_n = L.hist[mb];
            if (m_n) { m_o = L.off[mb]; m_f = L.fit[mb]; m_g = L.gbase[mb]; }
            m_n = (m_n + kRunAlign - 1) & ~(uint32_t)(kRunAlign - 1);
        }
    }
    const uint32_t per_wave = (fan + kSplitWaves - 1) / kSplitWaves;
    for (uint32_t it = 0; it < per_wave; it++) {
        const uint32_t n = __shfl(m_n, it);
        if (n == 0) continue;
        const uint32_t b = wave + it * kSplitWaves;
        const uint32_t o = __shfl(m_o, it), f = __shfl(m_f, it), gb = __shfl(m_g, it);
        uint64_t *dst = d.kmers + (child0 + b) * (uint64_t)d.cap + gb;
        if constexpr (kRunAlign >= 2) {   /* f, o and dst are multiples of kRunAlign: 16-byte lanes */
            for (uint32_t i = 2 * lane; i < f; i += 2 * kWave)
                *reinterpret_cast<ulonglong2 *>(dst + i) = *reinterpret_cast<const ulonglong2 *>(&L.sorted[o + i]);
        } else {
            for (uint32_t i = lane; i < f; i += kWave) dst[i] = L.sorted[o + i];
        }
        if (f < n) {                                           /* the child is full: spill the rest */
            const unsigned long long sb = L.sbase[b];
            for (uint32_t i = f + lane; i < n; i += kWave) {
                const uint64_t v = L.sorted[o + i];
                const unsigned long long at = sb + (i - f);
                if (at < d.spill_cap) d.spill[at] = v;         /* (pads spill too; consumers skip them) */
                else if (v != kPad) lost++;
            }
        }
    }
    lds_barrier();                                           /* LDS reusable */
}

/* Level 1: sequence text -> canonical k-mers -> 2^b1 buckets by the top bits of
 * the home slot.  One chunk = 16 tiles (one per wave). */
#if defined(QM2_ABL) && QM2_ABL >= 2
#define QM2_RANK(b) (b)
#else
#define QM2_RANK(b) atomicAdd(&L.hist[b], 1u)
#endif
template <bool K30>
__global__ __launch_bounds__(kSplitThreads) void split_text_kernel(SeqView sv, uint32_t k, uint32_t hmask,
                                                                   uint32_t shift, uint32_t fan,
                                                                   uint64_t n_chunks, SplitDst d, DevStats *stats)
{
    __shared__ SplitLds L;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    unsigned long long n_bases = 0, n_kmers = 0, lost = 0;
    /* the 16 bytes of this lane for the NEXT chunk are fetched while the current one is processed */
    uint4 bytes = load16_raw(sv, lane_offset(((uint64_t)blockIdx.x * kSplitWaves + wave) * kTileBytes, lane));
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        if (tid < kMaxFan) L.hist[tid] = 0;
        lds_barrier();
        const uint64_t tile = chunk * kSplitWaves + wave;
        const uint4 cur = bytes;
        bytes = load16_raw(sv, lane_offset(((chunk + gridDim.x) * kSplitWaves + wave) * kTileBytes, lane));
        const Window W = make_window_from<K30>(sv, tile * kTileBytes, lane, k, cur);
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
                tag[J] = (b << 16) | QM2_RANK(b);                                  \
            }                                                                      \
        }
        QM2_T(0) QM2_T(1) QM2_T(2) QM2_T(3) QM2_T(4) QM2_T(5) QM2_T(6) QM2_T(7)
        QM2_T(8) QM2_T(9) QM2_T(10) QM2_T(11) QM2_T(12) QM2_T(13) QM2_T(14) QM2_T(15)
#undef QM2_T
#ifdef QM2_ABL   /* timing ablations only (results are wrong): 1 = no commit, 2 = also no LDS rank atomics */
        {
            uint64_t x = 0; uint32_t y = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) { x ^= km[j]; y ^= tag[j]; }
            asm volatile("" :: "v"(x), "v"(y));
        }
#else
        split_commit(L, km, tag, 0, fan, d, lost);
#endif
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
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    unsigned long long lost = 0;
    const uint64_t *in = src + parent * (uint64_t)src_cap;
    /* workgroup x of this parent takes chunks x, x + gridDim.x, ...; the next chunk's k-mers are
     * loaded while the current one is sorted and copied out */
    uint64_t nxt[16];
    {
        const uint64_t first = (uint64_t)blockIdx.x * kSplitSlots;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t at = first + (uint64_t)j * kSplitThreads + tid;
            nxt[j] = at < n ? in[at] : kPad;
        }
    }
    for (uint64_t first = (uint64_t)blockIdx.x * kSplitSlots; first < n; first += (uint64_t)gridDim.x * kSplitSlots) {
        if (tid < kMaxFan) L.hist[tid] = 0;
        lds_barrier();
        uint64_t km[16];
        uint32_t tag[16];
        const uint64_t ahead = first + (uint64_t)gridDim.x * kSplitSlots;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            km[j] = nxt[j];
            const uint64_t at = ahead + (uint64_t)j * kSplitThreads + tid;
            nxt[j] = at < n ? in[at] : kPad;
            tag[j] = 0xFFFFFFFFu;
            if (km[j] != kPad) {
                const uint32_t b = ((djb32(km[j]) & hmask) >> shift) & (fan - 1);
                tag[j] = (b << 16) | atomicAdd(&L.hist[b], 1u);
            }
        }
        split_commit(L, km, tag, parent * fan, fan, d, lost);
    }
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

    constexpr int kSteps = (S + kHalo + kCountThreads - 1) / kCountThreads;
    constexpr int kGroup = 8;

    /* geometry of bucket b: own range, staged slice (own + kHalo slots on the side its probe
     * walks leave through, i.e. towards the middle of the table, :94-95), walk direction */
    struct Geo { uint64_t own_lo, own_hi, seg_lo; uint32_t seg; bool down; };
    auto geo = [&](uint32_t b) {
        Geo g;
        g.own_lo = (uint64_t)b << LG_S;
        g.own_hi = g.own_lo + S < H ? g.own_lo + S : H;
        g.down = (g.own_lo & (H >> 1)) != 0;                    /* upper half walks down */
        uint64_t lo = g.own_lo, hi = g.own_hi;
        if (H > (uint64_t)S) {
            if (g.down) lo = g.own_lo >= (uint64_t)kHalo ? g.own_lo - kHalo : 0;
            else        hi = g.own_hi + kHalo <= H ? g.own_hi + kHalo : H;
        }
        g.seg_lo = lo;
        g.seg = (uint32_t)(hi - lo);
        return g;
    };
    /* The table slice of the NEXT bucket is fetched into registers while the current one is
     * probed, and the next group of k-mers while the current group is probed: the workgroup
     * always has loads in flight. */
    uint64_t kv[kSteps];
    auto fetch_keys = [&](uint32_t b) {
        if (b >= n_buckets) return;
        const Geo g = geo(b);
#pragma unroll
        for (int st = 0; st < kSteps; st++) {
            const uint32_t i = tid + st * kCountThreads;
            kv[st] = i < g.seg ? tv.keys[g.seg_lo + i] : 0;
        }
    };
    fetch_keys(blockIdx.x);

    for (uint32_t b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        uint32_t n = cursors[b];
        if (n > cap) n = cap;
        const Geo g = geo(b);
        const uint64_t own_lo = g.own_lo, own_hi = g.own_hi, seg_lo = g.seg_lo;
        const uint32_t seg = g.seg;
        const bool down = g.down;
#pragma unroll
        for (int st = 0; st < kSteps; st++) {
            const uint32_t i = tid + st * kCountThreads;
            if (i < seg) { skey[i] = kv[st]; scnt[i] = 0; }
        }
        lds_barrier();
        fetch_keys(b + gridDim.x);
        const uint64_t *in = kmers + (uint64_t)b * cap;
        uint64_t nq[kGroup];
#pragma unroll
        for (int q = 0; q < kGroup; q++) {
            const uint32_t i = q * kCountThreads + tid;
            nq[q] = i < n ? in[i] : kPad;
        }
        for (uint32_t base = 0; base < n; base += kGroup * kCountThreads) {
            uint64_t kq[kGroup];
#pragma unroll
            for (int q = 0; q < kGroup; q++) {
                kq[q] = nq[q];
                const uint32_t i = base + (kGroup + q) * kCountThreads + tid;
                nq[q] = i < n ? in[i] : kPad;
            }
#pragma unroll
            for (int q = 0; q < kGroup; q++) {
                const uint64_t km = kq[q];
                if (km == kPad) continue;
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
                            const uint64_t gk = tv.keys[slot];
                            seen++;
                            if (gk == km) { atomicAdd(&tv.counters[slot], 1u); n_hits++; break; }
                            if (gk == 0) break;
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
        }
        lds_barrier();
        /* flush: the interior of the owned range is touched by this workgroup only (plain
         * read-modify-write, all reads issued before the writes); the kHalo slots at either
         * end of it and the staged halo can also receive a neighbour's counts (atomic add). */
        if (n) {
            uint32_t cv[kSteps], old[kSteps];
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const uint32_t i = tid + st * kCountThreads;
                cv[st] = i < seg ? scnt[i] : 0;
                old[st] = 0;
                if (cv[st]) {
                    const uint64_t slot = seg_lo + i;
                    const bool interior = slot >= own_lo + kHalo && slot + kHalo < own_hi;
                    uint32_t *p = tv.counters + (slot - tv.span_lo);
                    if (interior) old[st] = *p;
                    else { atomicAdd(p, cv[st]); cv[st] = 0; }
                }
            }
#pragma unroll
            for (int st = 0; st < kSteps; st++)
                if (cv[st]) tv.counters[seg_lo + tid + st * kCountThreads - tv.span_lo] = old[st] + cv[st];
        }
        lds_barrier();
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
