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
 *   probe_kernel        (direct path) for the rare spill: k-mers that do not fit their
 *                       bucket (skewed input, e.g. poly-A)
 *   count_kernel        (direct path) recount of the whole batch, decided on the device,
 *                       if even the spill list overflowed
 *
 * Exactness: every emitted k-mer lands in exactly one finest bucket or in the
 * spill list, and is probed exactly once under the reference's rule; integer
 * adds commute, so the order inside buckets is irrelevant.  A walk longer than
 * the staged halo finishes in global memory; the buckets such a walk can reach
 * (found once per table by long_run_kernel) give up their plain-store flush for
 * atomics, so nothing it adds is overwritten.
 */
#pragma once
#include <type_traits>
#include "qm2_device.hip.h"

namespace qm2 {

#ifndef QM2_SPLIT_THREADS
#define QM2_SPLIT_THREADS 1024
#endif
#ifndef QM2_COUNT_THREADS
#define QM2_COUNT_THREADS 512
#endif
constexpr int kSplitThreads  = QM2_SPLIT_THREADS;    /* 1024 threads x 16 k-mers: long runs per child beat
                                                        occupancy (512 x 16 and 1024 x 8, two workgroups
                                                        per CU, measured 10-25 % slower)                 */
constexpr int kSplitWaves    = kSplitThreads / kWave;
constexpr int kKpt           = 16;                   /* k-mers per thread per chunk = one tile per wave  */
constexpr int kSplitSlots    = kSplitThreads * kKpt; /* k-mers staged in LDS per chunk (128 KiB)         */
constexpr int kSplitTextBytes = kSplitWaves * kTileBytes;   /* 15872 sequence bytes per chunk   */
constexpr int kMaxFan        = 256;                  /* children per split                      */
constexpr uint64_t kPad      = ~0ull;                /* never a k-mer (k <= 31 uses at most 62 bits);
                                                        fills unused input slots                         */
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
    uint64_t sorted[kSplitSlots + 1];       /* the chunk sorted by child; last = trash slot          */
    uint32_t hist[kMaxFan + 1];             /* [kMaxFan] = trash bin of the empty slots (keeps the
                                               rank atomics and the scatter branch-free)             */
    uint32_t off[kMaxFan];                  /* exclusive prefix of hist                              */
    uint32_t gbase[kMaxFan];                /* reserved start in the child's region                  */
    uint32_t fit[kMaxFan];                  /* how many of this chunk's k-mers fit the child         */
    uint32_t wsum[kMaxFan / kWave];
};

/* What a wave needs to copy out its share of a sorted chunk: lane l holds the run of child
 * wave + l * kSplitWaves: its length, its start in L.sorted, how much of it fits the child,
 * and where that goes.  Read back with v_readlane at compile-time lane indices, so the
 * copy-out's control values live in scalar registers. */
struct RunDesc {
    uint32_t f, o;                  /* k-mers to copy, start in L.sorted */
    uint32_t dst_lo, dst_hi;
    bool live;
};

/* Sort one chunk by child in LDS.  Every thread holds kKpt k-mers km[j] with
 * tag[j] = (child << 16) | rank_in_child (rank from an LDS atomic on hist); empty slots carry
 * child id kMaxFan (trash bin).  On return L.sorted holds the chunk grouped by child and the
 * wave's run descriptors are in registers; the copy-out (copy_piece) is left to the caller,
 * which interleaves it with the NEXT chunk's compute so that the stores (HBM-bound, ~2 ms per
 * 10 GB) drain behind VALU work instead of stalling every wave at once. */
__device__ __forceinline__ RunDesc sort_chunk(SplitLds &L, const uint64_t (&km)[kKpt], const uint32_t (&tag)[kKpt],
                                              uint64_t child0, uint32_t fan, const SplitDst &d,
                                              unsigned long long &lost)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    lds_barrier();                 /* hist complete; every wave is done reading the previous L.sorted */
    /* exclusive scan of hist[0..fan) by the first kMaxFan threads (4 waves) */
    uint32_t mine = 0, incl = 0;
    static_assert(kSplitThreads >= kMaxFan, "the scan uses one thread per child");
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
    lds_barrier();
    uint32_t g = 0;
    const bool reserve = tid < (int)fan && mine != 0;
    if (tid < kMaxFan) {
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += L.wsum[w];
        L.off[tid] = before + incl - mine;
        /* reserve the run in the child's region; the returned offset is only needed for the
         * copy-out, so the round trip overlaps the LDS scatter below */
        if (reserve) g = atomicAdd(&d.cursors[child0 + (uint64_t)tid], mine);
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < kKpt; j++) {
        /* empty slots carry child id kMaxFan: they all land in the trash slot, no branch */
        const uint32_t child = tag[j] >> 16;
        const uint32_t pos = child >= kMaxFan ? (uint32_t)kSplitSlots : L.off[child & (kMaxFan - 1)] + (tag[j] & 0xFFFFu);
        L.sorted[pos] = km[j];
    }
    uint32_t f = mine;
    if (reserve) {
        f = g >= d.cap ? 0u : (d.cap - g < mine ? d.cap - g : mine);
        L.gbase[tid] = g;
        L.fit[tid] = f;
    }
    lds_barrier();
    if (reserve && f < mine) {
        /* rare (skewed input): the child is full; this thread moves the rest of the run to the
         * spill list, which the direct kernel probes at the end of the batch */
        const unsigned long long sb = atomicAdd(d.spill_count, (unsigned long long)(mine - f));
        const uint32_t o = L.off[tid];
        for (uint32_t i = f; i < mine; i++) {
            const unsigned long long at = sb + (i - f);
            if (at < d.spill_cap) d.spill[at] = L.sorted[o + i];
            else lost++;
        }
    }
    static_assert(kMaxFan <= kSplitWaves * kWave, "one descriptor per lane");
    RunDesc r = {0, 0, 0, 0, true};
    const uint32_t mb = wave + lane * kSplitWaves;
    if (mb < fan && L.hist[mb]) {
        r.f = L.fit[mb];
        r.o = L.off[mb];
        const uint64_t dst = (uint64_t)(d.kmers + (child0 + mb) * (uint64_t)d.cap + L.gbase[mb]);
        r.dst_lo = (uint32_t)dst;
        r.dst_hi = (uint32_t)(dst >> 32);
    }
    lds_barrier();                 /* descriptors are in registers: hist / off may be reused */
    return r;
}

/* Copy-out of the run with descriptor index IT (child wave + IT * kSplitWaves): one
 * full-wave contiguous store per 64 k-mers.  All 64 lanes must call. */
template <int IT>
__device__ __forceinline__ void copy_piece(const SplitLds &L, const RunDesc &r, const SplitDst &d)
{
    /* (the builtin returns int: go through uint32_t, or the low half sign-extends into the pointer) */
    const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)r.f, IT);
    if (f == 0) return;                                        /* scalar branch */
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)r.o, IT);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)r.dst_lo, IT);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)r.dst_hi, IT);
    /* a GLOBAL pointer (address space 1): rebuilt from two integers the compiler would otherwise treat
     * it as generic and emit flat_store, which also counts on lgkmcnt — and every LDS wait and
     * lds_barrier() in these kernels would then wait for the copy-out stores as well */
    typedef uint64_t __attribute__((address_space(1))) global_u64;
    global_u64 *dst = (global_u64 *)(((uint64_t)hi << 32) | (uint64_t)lo);
    for (uint32_t i = lane; i < f; i += kWave) dst[i] = L.sorted[o + i];
}

/* all pieces of a sorted chunk (kernel epilogue) */
__device__ __forceinline__ void copy_all(const SplitLds &L, const RunDesc &r, const SplitDst &d)
{
    copy_piece<0>(L, r, d);  copy_piece<1>(L, r, d);  copy_piece<2>(L, r, d);  copy_piece<3>(L, r, d);
    copy_piece<4>(L, r, d);  copy_piece<5>(L, r, d);  copy_piece<6>(L, r, d);  copy_piece<7>(L, r, d);
    copy_piece<8>(L, r, d);  copy_piece<9>(L, r, d);  copy_piece<10>(L, r, d); copy_piece<11>(L, r, d);
    copy_piece<12>(L, r, d); copy_piece<13>(L, r, d); copy_piece<14>(L, r, d); copy_piece<15>(L, r, d);
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
    static_assert(kMaxFan <= 16 * kSplitWaves, "a wave copies at most 16 runs per chunk");
    RunDesc pend = {0, 0, 0, 0, false};
    /* the 16 bytes of this lane for the NEXT chunk are fetched while the current one is processed */
    uint4 bytes = load16_raw(sv, lane_offset(((uint64_t)blockIdx.x * kSplitWaves + wave) * kTileBytes, lane));
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t tile = chunk * kSplitWaves + wave;
        const uint4 cur = bytes;
        bytes = load16_raw(sv, lane_offset(((chunk + gridDim.x) * kSplitWaves + wave) * kTileBytes, lane));
        const Window W = make_window_from<K30>(sv, tile * kTileBytes, lane, k, cur);
        n_bases += W.n_bases;
        n_kmers += __builtin_popcount(W.emit);
        if (tid <= kMaxFan) L.hist[tid] = 0;
        lds_barrier();
        uint64_t km[kKpt];
        uint32_t tag[kKpt];
        /* k-mer J of this chunk, then copy-out piece J of the previous chunk */
#define QM2_T(J)                                                                   \
        {                                                                          \
            km[J] = kmer_at<K30, J>(W, k);                                         \
            const uint32_t b = (W.emit & (0x8000u >> J)) ? (djb32(km[J]) & hmask) >> shift : (uint32_t)kMaxFan; \
            tag[J] = (b << 16) | (atomicAdd(&L.hist[b], 1u) & 0xFFFFu);            \
            if (pend.live) copy_piece<J>(L, pend, d);                              \
        }
        QM2_T(0) QM2_T(1) QM2_T(2) QM2_T(3) QM2_T(4) QM2_T(5) QM2_T(6) QM2_T(7)
        QM2_T(8) QM2_T(9) QM2_T(10) QM2_T(11) QM2_T(12) QM2_T(13) QM2_T(14) QM2_T(15)
#undef QM2_T
        pend = sort_chunk(L, km, tag, 0, fan, d, lost);
    }
    if (pend.live) copy_all(L, pend, d);
    n_bases = wave_sum(n_bases); n_kmers = wave_sum(n_kmers); lost = wave_sum(lost);
    if (lane == 0) {
        atomicAdd(&stats->n_bases, n_bases);
        atomicAdd(&stats->n_kmers, n_kmers);
        if (lost) atomicAdd(&stats->part_lost, lost);
    }
}

/* Level l >= 2: split every level l-1 bucket into 2^bl children.
 * grid.y = parent; workgroup x of a parent takes its chunks x, x + gridDim.x, ... */
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
    RunDesc pend = {0, 0, 0, 0, false};
    /* the next chunk's k-mers are loaded while the current one is sorted and copied out */
    uint64_t nxt[kKpt];
    {
        const uint64_t first = (uint64_t)blockIdx.x * kSplitSlots;
#pragma unroll
        for (int j = 0; j < kKpt; j++) {
            const uint64_t at = first + (uint64_t)j * kSplitThreads + tid;
            nxt[j] = at < n ? in[at] : kPad;
        }
    }
    for (uint64_t first = (uint64_t)blockIdx.x * kSplitSlots; first < n; first += (uint64_t)gridDim.x * kSplitSlots) {
        if (tid <= kMaxFan) L.hist[tid] = 0;
        lds_barrier();
        uint64_t km[kKpt];
        uint32_t tag[kKpt];
        const uint64_t ahead = first + (uint64_t)gridDim.x * kSplitSlots;
#define QM2_K(J)                                                                   \
        {                                                                          \
            km[J] = nxt[J];                                                        \
            const uint64_t at = ahead + (uint64_t)(J) * kSplitThreads + tid;       \
            nxt[J] = at < n ? in[at] : kPad;                                       \
            const uint32_t b = km[J] != kPad ? ((djb32(km[J]) & hmask) >> shift) & (fan - 1) : (uint32_t)kMaxFan; \
            tag[J] = (b << 16) | (atomicAdd(&L.hist[b], 1u) & 0xFFFFu);            \
            if (pend.live) copy_piece<J>(L, pend, d);                              \
        }
        QM2_K(0) QM2_K(1) QM2_K(2) QM2_K(3) QM2_K(4) QM2_K(5) QM2_K(6) QM2_K(7)
        QM2_K(8) QM2_K(9) QM2_K(10) QM2_K(11) QM2_K(12) QM2_K(13) QM2_K(14) QM2_K(15)
#undef QM2_K
        pend = sort_chunk(L, km, tag, parent * fan, fan, d, lost);
    }
    if (pend.live) copy_all(L, pend, d);
    lost = wave_sum(lost);
    if (lane == 0 && lost) atomicAdd(&stats->part_lost, lost);
}

/* Finest level: one workgroup per bucket of S = 2^LG_S table slots.
 *
 * Probing in LDS is cheap per step but DIVERGENT: at 53 % fill (hg38) a k-mer needs 1.75
 * slots on average while the slowest of a wave's 64 lanes needs ~12, and a plain per-lane loop
 * makes every lane pay for the slowest.  So each group of k-mers is probed in two phases:
 *   A  straight-line: the home slot and the one after it are inspected for every k-mer
 *      (resolves ~75 % at 53 % fill, ~99 % at 14 %);
 *   B  the unresolved ones are COMPACTED into a small per-wave LDS queue and probed on in
 *      dense rows, so the long tails of few k-mers no longer idle the other lanes.
 *
 * Template parameters (chosen per launch by launch_partitioned):
 *   PIPE  few k-mers per bucket (hg38-scale tables): the counter tile is prefetched with the
 *         key slice a bucket ahead and stored back plainly; otherwise the tile starts at zero
 *         and only the sectors that got counts are read and written at flush time;
 *   G     rows of kCountThreads k-mers per group (each thread holds G k-mers in registers).
 * Two workgroups per CU (LDS), 4 waves per SIMD: the register budget is 128 VGPRs. */
template <int LG_S, bool PIPE, int G>
__global__ __launch_bounds__(kCountThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void bucket_count_kernel(const uint64_t *kmers, const uint32_t *cursors, uint32_t cap, TableView tv, uint64_t H,
                         uint32_t n_buckets, const uint32_t *shared, DevStats *stats)
{
    if (stats->part_lost != 0) return;       /* this batch is recounted by the direct kernel */
    constexpr int S = 1 << LG_S;
    constexpr int kWaves = kCountThreads / kWave;
    __shared__ uint64_t skey[S + kHalo];
    __shared__ uint32_t scnt[S + kHalo];
    constexpr int kGroup = G;                         /* k-mers per thread per group */
    constexpr int kQueueCap = 32 * G;                 /* per wave; what does not fit is finished in place */
    __shared__ uint64_t qk[kWaves][kQueueCap];
    __shared__ uint16_t qa[kWaves][kQueueCap];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    uint32_t n_probes = 0, n_hits = 0;      /* per lane: < 2^32 for any batch that fits the plan */

    constexpr int kSteps = (S + kHalo + kCountThreads - 1) / kCountThreads;

    /* geometry of bucket b: own range, staged slice (own + kHalo slots on the side its probe
     * walks leave through, i.e. towards the middle of the table, :94-95), walk direction */
    /* int_lo / int_len: the private interior in slice coordinates (i - int_lo < int_len, unsigned) */
    struct Geo { uint64_t own_lo, own_hi, seg_lo; uint32_t seg, int_lo, int_len; bool down, shared; };
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
        /* a long run of occupied slots reaches into this bucket: nothing in it is private */
        g.shared = shared != nullptr && ((shared[b >> 5] >> (b & 31)) & 1u) != 0;
        const uint32_t own_len = (uint32_t)(g.own_hi - g.own_lo);
        g.int_lo = (uint32_t)(g.own_lo - lo) + kHalo;
        g.int_len = !g.shared && own_len > 2u * kHalo ? own_len - 2u * kHalo : 0u;
        return g;
    };
    /* Software pipeline over the buckets of this workgroup (stride gridDim.x): everything bucket
     * b+stride needs from HBM — its table slice, the old values of its counters, its fill and its
     * first group of k-mers — is loaded into registers while bucket b is probed, and later groups
     * of b while the current group is probed.  Nothing in the steady state waits for a load that
     * was issued less than one bucket ago. */
    const uint32_t stride = gridDim.x;
    uint64_t kv[kSteps];
    uint32_t ov[kSteps];
    auto fetch_slice = [&](uint32_t b) {
        if (b >= n_buckets) return;
        const Geo g = geo(b);
        const uint64_t *kp = tv.keys + g.seg_lo;
#pragma unroll
        for (int st = 0; st < kSteps; st++) {
            const uint32_t i = tid + st * kCountThreads;
            kv[st] = i < g.seg ? kp[i] : 0;
        }
        if constexpr (PIPE) {
            const uint32_t *cp = tv.counters + (g.seg_lo - tv.span_lo);
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const uint32_t i = tid + st * kCountThreads;
                ov[st] = i - g.int_lo < g.int_len ? cp[i] : 0;
            }
        }
    };
    auto fill_of = [&](uint32_t b) {
        uint32_t n = b < n_buckets ? cursors[b] : 0;
        return n > cap ? cap : n;
    };
    uint64_t nq[kGroup];
    uint32_t n_next = 0, n_after = 0;
    fetch_slice(blockIdx.x);
    if constexpr (PIPE) {
        n_next = fill_of(blockIdx.x);                         /* fill of the bucket about to start */
        const uint64_t *in0 = kmers + (uint64_t)blockIdx.x * cap;
#pragma unroll
        for (int q = 0; q < kGroup; q++) {
            const uint32_t i = q * kCountThreads + tid;
            nq[q] = i < n_next ? in0[i] : kPad;
        }
        n_after = fill_of(blockIdx.x + stride);               /* ... and of the one after it */
    }

    for (uint32_t b = blockIdx.x; b < n_buckets; b += stride) {
        const uint32_t n = PIPE ? n_next : fill_of(b);
        const Geo g = geo(b);
        const uint64_t seg_lo = g.seg_lo;
        const uint32_t seg = g.seg;
        const bool down = g.down;
        const uint32_t int_lo = g.int_lo, int_len = g.int_len;
        uint32_t *const cp = tv.counters + (seg_lo - tv.span_lo);
        /* the counter tile starts from the old values of the interior (0 elsewhere), so the LDS
         * atomics accumulate straight onto them */
#pragma unroll
        for (int st = 0; st < kSteps; st++) {
            const uint32_t i = tid + st * kCountThreads;
            if (i < seg) { skey[i] = kv[st]; scnt[i] = PIPE ? ov[st] : 0u; }
        }
        lds_barrier();
        const uint32_t bn = b + stride;
        fetch_slice(bn);
        if constexpr (PIPE) {
            n_next = n_after;
            n_after = fill_of(bn + stride);
        }

        /* continue a walk from staged position `at` (may be outside the slice: then, and when it
         * leaves the slice later, it finishes in global memory like the direct kernel) */
        auto walk = [&](uint64_t km, int32_t at, bool dn, uint32_t seen) {
            for (;;) {
                if (at < 0 || at >= (int32_t)seg) {
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
        };

        const uint64_t *in = kmers + (uint64_t)b * cap;
        const uint64_t *in_next = kmers + (uint64_t)bn * cap;
        if constexpr (!PIPE) {
#pragma unroll
            for (int q = 0; q < kGroup; q++) {
                const uint32_t i = q * kCountThreads + tid;
                nq[q] = i < n ? in[i] : kPad;
            }
        }
        uint32_t base = 0;
        /* PIPE runs the body once even for an empty bucket: it carries the next bucket's prefetch */
        while (PIPE ? (base == 0 || base < n) : base < n) {
            uint64_t kq[kGroup];
            const bool last = PIPE && base + kGroup * kCountThreads >= n;   /* workgroup-uniform */
#pragma unroll
            for (int q = 0; q < kGroup; q++) {
                kq[q] = nq[q];
                /* next group of this bucket, or (PIPE) behind the last one the first of the next bucket */
                const uint32_t j = q * kCountThreads + tid;
                const uint32_t i = base + kGroup * kCountThreads + j;
                nq[q] = last ? (j < n_next ? in_next[j] : kPad) : (i < n ? in[i] : kPad);
            }
            /* ---- phase A: two slots per k-mer, no loop.  All kGroup rows run as one straight-line
             * block even when the last ones hold no k-mer: skipping rows (a branch per row, or one
             * block per row count) was measured 12 % SLOWER, so the host picks kGroup to fit the
             * bucket fill instead. ---- */
            uint32_t queued = 0;                                            /* wave-uniform */
            auto phase_a = [&]() {
#pragma unroll
                for (int q = 0; q < kGroup; q++) {
                    const uint64_t km = kq[q];
                    const bool live = km != kPad;
                    const uint32_t home = djb32(km) & tv.hmask;                    /* :92 */
                    /* H <= S: the single bucket holds both halves, direction per k-mer */
                    const bool dn = H > (uint64_t)S ? down : (home & tv.hhalf) != 0;
                    const int32_t at0 = live ? (int32_t)((uint64_t)home - seg_lo) : 0;
                    const int32_t at1 = at0 + (dn ? -1 : 1);
                    const bool in1 = at1 >= 0 && at1 < (int32_t)seg;
                    const uint64_t v0 = skey[at0];
                    const uint64_t v1 = skey[in1 ? at1 : at0];
                    const bool hit0 = live && v0 == km;                          /* :98 */
                    const bool end0 = hit0 || v0 == 0;
                    const bool hit1 = live && !end0 && in1 && v1 == km;
                    const bool end1 = end0 || (in1 && (hit1 || v1 == 0));
                    if (hit0) atomicAdd(&scnt[at0], 1u);                          /* :291 */
                    if (hit1) atomicAdd(&scnt[at1], 1u);
                    n_hits += (hit0 || hit1) ? 1 : 0;
                    const bool more = live && !end1;       /* both occupied by other keys, or at1 outside */
                    n_probes += live ? (end0 ? 1u : (end1 ? 2u : 0u)) : 0u;   /* unresolved: walk() adds the total */
                    /* unresolved: compact into this wave's queue; entry = position to look at next,
                     * direction, and how many slots were counted so far */
                    const unsigned long long m = __ballot(more);
                    if (m) {
                        const uint32_t slot = queued + __popcll(m & ((1ull << lane) - 1));
                        const int32_t nxt = in1 ? at1 + (dn ? -1 : 1) : at1;
                        const uint32_t seen = in1 ? 2u : 1u;
                        if (more) {
                            if (slot < (uint32_t)kQueueCap) {
                                qk[wave][slot] = km;
                                qa[wave][slot] = (uint16_t)((uint32_t)(nxt + 1) | (dn ? 0x2000u : 0u) | ((seen - 1) << 14));
                            } else {
                                walk(km, nxt, dn, seen);             /* queue full: finish in place */
                            }
                        }
                        queued += __popcll(m);
                    }
                }
            };
            phase_a();
            /* ---- phase B: dense rows over the compacted queue ---- */
            const uint32_t qn = queued < (uint32_t)kQueueCap ? queued : (uint32_t)kQueueCap;
            for (uint32_t e = lane; e < qn; e += kWave) {
                const uint64_t km = qk[wave][e];
                const uint32_t a = qa[wave][e];
                walk(km, (int32_t)(a & 0x1FFFu) - 1, (a & 0x2000u) != 0, (a >> 14) + 1);
            }
            base += kGroup * kCountThreads;
        }
        lds_barrier();
        /* flush: the interior of the owned range is touched by this workgroup only, so the tile
         * (old values, fetched a bucket ago, plus this bucket's counts) is stored back plainly; the
         * kHalo slots at either end of it and the staged halo can also receive a neighbour's
         * counts, so there the tile holds deltas and they are added atomically. */
        if (n) {
            if constexpr (PIPE) {
#pragma unroll
                for (int st = 0; st < kSteps; st++) {
                    const uint32_t i = tid + st * kCountThreads;
                    if (i < seg) {
                        const uint32_t cv = scnt[i];
                        if (i - int_lo < int_len) cp[i] = cv;
                        else if (cv) atomicAdd(cp + i, cv);
                    }
                }
            } else {
                /* sparse table: only the sectors that got counts are read and written (all reads
                 * issued before the writes) */
                uint32_t cv[kSteps], old[kSteps];
#pragma unroll
                for (int st = 0; st < kSteps; st++) {
                    const uint32_t i = tid + st * kCountThreads;
                    cv[st] = i < seg ? scnt[i] : 0;
                    old[st] = 0;
                    if (cv[st]) {
                        if (i - int_lo < int_len) old[st] = cp[i];
                        else { atomicAdd(cp + i, cv[st]); cv[st] = 0; }
                    }
                }
#pragma unroll
                for (int st = 0; st < kSteps; st++)
                    if (cv[st]) cp[tid + st * kCountThreads] = old[st] + cv[st];
            }
        }
        lds_barrier();
    }
    const unsigned long long w_probes = wave_sum((unsigned long long)n_probes), w_hits = wave_sum((unsigned long long)n_hits);
    if (lane == 0) {
        atomicAdd(&stats->n_probes, w_probes);
        atomicAdd(&stats->n_hits, w_hits);
    }
}

/* bucket_count flushes the interior of a bucket with plain stores, which is only right while no
 * OTHER workgroup adds to those slots.  A walk visits one run of occupied slots plus the slot behind
 * it; runs shorter than kHalo keep every walk inside the walker's own slice (own slots + halo),
 * where the neighbour uses atomics.  A longer run lets walks go on into whatever it covers, so
 * every bucket that holds a slot of such a run, or the slot before / behind it (k-mer 0 counts on
 * the empty slot, :96-98), is marked SHARED: its whole tile is flushed with atomic adds.  Found once
 * per context; bit 1 of *found: a run too long to follow (the direct kernel counts that table). */
constexpr uint32_t kRunFollow = 1u << 20;
__global__ void long_run_kernel(const uint64_t *keys, uint64_t H, uint32_t limit, uint32_t lg_s, uint32_t *shared,
                                unsigned int *found)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < H; i += (uint64_t)gridDim.x * blockDim.x) {
        if (keys[i] == 0) continue;
        if (i != 0 && keys[i - 1] != 0) continue;             /* not the start of a run */
        uint64_t len = 1;
        while (len < kRunFollow && i + len < H && keys[i + len] != 0) len++;
        if (len < limit) continue;
        if (len >= kRunFollow) { atomicOr(found, 2u); continue; }
        atomicOr(found, 1u);
        const uint64_t lo = (i ? i - 1 : 0) >> lg_s, hi = (i + len < H ? i + len : H - 1) >> lg_s;
        for (uint64_t b = lo; b <= hi; b++) atomicOr(&shared[b >> 5], 1u << (b & 31));
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
