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
 *   table_image_kernel  once per key table: every slot as a bucket record, what bucket_count streams
 *
 * The finest level and the table slices are in RECORD form (qm2_device.hip.h, "bucket records"): 6 bytes per k-mer /
 * slot at hg38 scale instead of 8, with the home slot inside the bucket in the clear — whichever split writes the
 * finest level makes the records (it has the hash in hand), bucket_count compares records and hashes nothing.
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
#define QM2_COUNT_THREADS 1024
#endif
constexpr int kSplitThreads  = QM2_SPLIT_THREADS;    /* 1024 threads x 16 k-mers: long runs per child beat
                                                        occupancy (512 x 16 and 1024 x 8, two workgroups
                                                        per CU, measured 10-25 % slower)                 */
constexpr int kSplitWaves    = kSplitThreads / kWave;
constexpr int kKpt           = 16;                   /* k-mers per thread per chunk = one tile per wave  */
constexpr int kSplitSlots    = kSplitThreads * kKpt; /* k-mers staged in LDS per chunk (128 KiB)         */
constexpr int kSplitTextBytes = kSplitWaves * kTileBytes;   /* 15872 sequence bytes per chunk   */
constexpr int kMaxFan        = 1024;                 /* children per split: hg38-scale tables (20 bucket bits) take 2 levels */
constexpr uint64_t kPad      = ~0ull;                /* never a k-mer (k <= 31 uses at most 62 bits);
                                                        fills unused input slots                         */
#ifndef QM2_HALO
#define QM2_HALO 256
#endif
constexpr int kHalo          = QM2_HALO;             /* table slots staged past a bucket's
                                                        middle-facing edge                      */
constexpr int kCountThreads  = QM2_COUNT_THREADS;
#ifndef QM2_STAMPS_SORT
#define QM2_STAMPS_SORT 0         /* diagnostic build (-DQM2_STAMPS): 1 = also the phases of sort_chunk in split_text_kernel */
#endif
/* The chunk sort's workgroup barriers (sort_chunk).  0: five per chunk (zero the counts | rank | scan per wave | wave totals |
 * scatter).  2: four — two count arrays take turns and the one not in use is zeroed during the scan of the other, so the chunk
 * loop has no zero-and-barrier of its own.  Same box, C3: split_text 20.37 ms with five, 20.95 with four; split_kmer 24.23 with
 * five, 23.85 with four (EXPERIMENTS.md; a three-barrier arrangement in which every wave scans all the counts itself lost on
 * both: profiles/r05_c3/sort_chunk_three_barriers.patch). */
#ifndef QM2_SORT_SYNC_TEXT
#define QM2_SORT_SYNC_TEXT 0
#endif
#ifndef QM2_SORT_SYNC_KMER
#define QM2_SORT_SYNC_KMER 2
#endif
#ifndef QM2_COUNT_BLOCKED
#define QM2_COUNT_BLOCKED 0
#endif
#ifndef QM2_PREFETCH_LATE
#define QM2_PREFETCH_LATE 1       /* the next group is asked for behind the first probe set, not in front of it: away from the loaders' burst
                                     at the start of a bucket (C3, same box: bucket_count 23.83 -> 23.45 ms; 23.73 -> 23.40 with QM2_COUNT_BLOCKED) */
#endif
#ifndef QM2_COUNT_SET
#define QM2_COUNT_SET 6           /* k-mers probed together (measured at C3: 2 -> 23.2, 4 -> 22.9, 6 -> 22.0 ms) */
#endif

/* Workgroup barrier for LDS hand-offs.  hipcc's __syncthreads() also drains every outstanding
 * global load / store / atomic of the wave (s_waitcnt vmcnt(0)); these kernels keep a prefetch,
 * a cursor reservation and the previous chunk's copy-out stores in flight ACROSS their
 * barriers, so only the LDS traffic is waited for here.  (The compiler still inserts the vmcnt
 * wait in front of the first use of each loaded value.) */
/* Test build only (-DQM2_TEST_CHAOS=n, tests/test_gpu_parity.py::test_waves_far_out_of_step): every wave idles a
 * pseudo-random 0 .. 15 x n x 64 cycles at every workgroup barrier and between the phases in between, differently at
 * every visit.  Nothing in these kernels may depend on waves running in step — only on the barriers. */
#ifdef QM2_TEST_CHAOS
__device__ __forceinline__ void chaos(uint32_t site)
{
    static __device__ unsigned int visit;                         /* (any changing number will do: racy on purpose) */
    const uint32_t v = __builtin_amdgcn_readfirstlane((int)(visit++));
    const uint32_t w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t x = (((w + 1u) * 2654435761u) ^ ((site + 1u) * 40503u) ^ (v * 2246822519u)) >> 28;
    for (uint32_t i = 0; i < x * (uint32_t)QM2_TEST_CHAOS; i++) __builtin_amdgcn_s_sleep(1);
}
#define QM2_CHAOS(S) chaos(S)
#else
#define QM2_CHAOS(S) do { } while (0)
#endif
__device__ __forceinline__ void lds_barrier()
{
    QM2_CHAOS(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    QM2_CHAOS(1);
}

/* Diagnostic build only (-DQM2_STAMPS, tools/stamps.sh): s_memtime brackets around the phases of
 * bucket_count_kernel, summed per wave into qm2_stamp_acc (cdna_hip_programming.md section 7, In-kernel
 * stamps).  Compiled out of the product. */
#ifdef QM2_STAMPS
__device__ unsigned long long qm2_stamp_acc[32];
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define QM2_STAMP(I) do { const unsigned long long now_ = stamp_now(); st_acc[I] += now_ - st_last; st_last = now_; } while (0)
#else
#define QM2_STAMP(I) do { } while (0)
#endif

/* Two layouts of a level's buffer.
 *   child-major (SEG = false): child c occupies [c * cap, (c + 1) * cap).
 *   segmented   (SEG = true; the level that the text — or the slabs — are split into, when another split reads
 *     it): 64 KiB segments laid out [segment][child], k-mer e of child c at
 *         ((e >> kSegLg) << lg_row) + (c << kSegLg) + (e & (kSegKmers - 1)),    2^lg_row = children * kSegKmers.
 *     Why: that level is written as 1024 streams by every workgroup.  Child-major they lie ~70 MB apart over a 72 GB
 *     buffer (C3, 50 M reads per launch), and rocprofv3 shows what that costs: TCP_UTCL1_TRANSLATION_MISS = one per
 *     run written (5.0e8 per launch, 23 % of the requests; split_kmer, whose 1024 children are neighbours: 1e5) and
 *     GRBM_UTCL2_BUSY = the whole kernel.  tools/micro/tlb_probe.hip: a CU's translation cache holds 1024 write
 *     streams 2 MiB apart (no misses) but not 64 of them 64 MiB apart.  Uniform hashing fills all children at the
 *     same pace, so segmented, all 1024 fronts sit inside one or two 64 MiB rows. */
struct SplitDst {
    uint64_t           *kmers;      /* the level's buffer                                        */
    uint32_t           *cursors;    /* per child: k-mers reserved so far (may exceed cap)        */
    uint32_t            cap;        /* k-mers a child's region holds (segmented: whole segments) */
    uint64_t           *spill;      /* k-mers that did not fit: probed by the direct kernel      */
    unsigned long long *spill_count;
    uint64_t            spill_cap;
    uint32_t            lg_row;     /* segmented layout: log2(k-mers per row of segments)        */
    uint32_t            subs;       /* 1, or 8: every child's region is cut into one sub-region per XCD (below) */
    /* the FINEST level (FIN = 2 or 4 in the kernels below) is written as bucket records (qm2_device.hip.h): `kmers` is then
     * an array of 32-bit words R, region-major like the k-mers would be, and fin_w the array of second words beside it */
    void               *fin_w;
    uint32_t            lgH, lg_s;  /* log2 of the table's slots / of a bucket's (rec_w) */
};

/* Sub-regions (the level the text — or the slabs — are split into, when another split reads it).
 *
 * A run is ~16 k-mers and starts wherever the previous reservation of its child ended, so most 128-byte lines of
 * a child's region are written by two workgroups.  When those sit on different XCDs, the line is dirty in two L2s,
 * each of which writes its part back as a masked write: tools/micro/front_probe.hip measures 23 ms for 35 GB
 * written that way against 12.9 ms when every line only ever passes through ONE L2 (and 18.6 - 23 ms from one
 * process to the next — the text level's own spread, 21 - 29 ms, before this).  So each child has 8 sub-regions
 * with a cursor each, region (x * fan + c) for workgroups with blockIdx.x % 8 == x, which is the XCD the
 * hardware dispatches them to (round-robin; were it not, only this speed-up would be lost).  The next level
 * takes the 8 * fan sub-regions as parents; parent P feeds the children of P % fan. */
__device__ __forceinline__ uint64_t sub_region0(const SplitDst &d, uint32_t fan)
{
    return (uint64_t)(blockIdx.x & (d.subs - 1u)) * fan;
}


constexpr int kSpillRuns = kMaxFan;         /* overflowing runs of a chunk that the whole workgroup copies out together: every child may */
struct SplitLds {
    uint64_t sorted[kSplitSlots + 1];       /* the chunk sorted by child; last = trash slot          */
    uint32_t hist[2][kMaxFan + 1];          /* counts per child; [kMaxFan] = trash bin of the empty slots (keeps the rank
                                               atomics and the scatter branch-free).  Two arrays: with SYNC = 2 they take
                                               turns chunk by chunk (QM2_SORT_SYNC_*), else only [0] is used */
    uint16_t off[kMaxFan + 2];              /* exclusive prefix of hist (< 2^16: a chunk has <= 16 K k-mers); [kMaxFan] = the
                                               chunk's total: the empty slots scatter behind the real k-mers, where the
                                               copy-out never looks (total + empty slots = kSplitSlots exactly) */
    uint2    dl[kMaxFan];                   /* copy-out of the sorted chunk: element i of child c goes
                                               to slot i + dl.x of the child's region if i < dl.y
                                               (dl.x = reserved start - off, dl.y = off + what fits)  */
    uint32_t wsum[kMaxFan / kWave];
    uint32_t spill_n;                       /* runs of this chunk that did not fit their child ...   */
    uint32_t spill_need;                    /* ... their k-mers, summed (every run's place inside the chunk's piece of the spill list) ... */
    uint32_t spill_room;                    /* ... how many of them the spill list takes, and from where (copy_spills) */
    unsigned long long spill_base;
    uint2    spill_e[kSpillRuns];           /* ... {first element in sorted | how many << 16, place inside the chunk's piece} */
};
static_assert(kMaxFan % kSplitThreads == 0 || kSplitThreads > kMaxFan, "whole children per thread");
constexpr int kChildPerThread = kSplitThreads >= kMaxFan ? 1 : kMaxFan / kSplitThreads;
constexpr int kSplitWgPerCu = kSplitThreads >= 1024 ? 1 : 2;      /* what the LDS of a CU holds */
#if QM2_SPLIT_THREADS < 1024
#define QM2_SPLIT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define QM2_SPLIT_OCCUPANCY
#endif

/* The announced overflow runs of a chunk (sort_chunk), copied to the spill list by the whole workgroup.  Out of line:
 * inlined, its registers pushed values of the chunk loop into scratch and cost the loop 12 % — and it hardly ever runs.
 * ONE reservation per chunk (thread 0, for the sum of its runs): until round 5 every overflowing run reserved its own place
 * with a returning atomic on the one global counter, in the middle of the sort — with 5 % of the reads from 1000 repeat
 * families (a tenth of the finest buckets overflow) that was a million same-address atomics per launch, each of them waited
 * for by a whole workgroup: split_kmer 24.5 ms against 20.8 on uniform reads, 21.0 with the atomic taken out (timing build).
 * Returns what this lane's runs lost (the spill list full: the batch is then recounted by the direct kernel). */
__device__ __attribute__((noinline)) uint32_t copy_spills(SplitLds &L, uint64_t *spill, unsigned long long *spill_count, uint64_t spill_cap)
{
    /* (the destination by value: a reference to the kernel's SplitDst would move that whole struct into scratch) */
    if (threadIdx.x == 0) {
        const uint32_t need = L.spill_need;
        const unsigned long long sb = atomicAdd(spill_count, (unsigned long long)need);
        L.spill_base = sb;
        L.spill_room = sb >= spill_cap ? 0u : (spill_cap - sb < need ? (uint32_t)(spill_cap - sb) : need);     /* (spill_cap < 2^32) */
    }
    lds_barrier();
    const uint32_t ns = L.spill_n < (uint32_t)kSpillRuns ? L.spill_n : (uint32_t)kSpillRuns;
    const uint32_t room = L.spill_room;
    const unsigned long long base = L.spill_base;
    /* a wave per run (runs are ~16 k-mers) */
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t lost = 0;
    for (uint32_t r = wave; r < ns; r += kSplitWaves) {
        const uint2 e = L.spill_e[r];
        const uint32_t first = e.x & 0xFFFFu, len = e.x >> 16, at = e.y;
        for (uint32_t x = lane; x < len; x += kWave) if (at + x < room) spill[base + at + x] = L.sorted[first + x];
        if (lane == 0 && at + len > room) lost += at >= room ? len : at + len - room;
    }
    return lost;
}

/* Inclusive sum over the wave's 64 lanes in the vector ALU alone (DPP: four shifts inside the rows of 16, then lane 15 and
 * lane 31 broadcast onwards) — __shfl_up makes every step a ds_bpermute_b32, an LDS round trip with all the waves of the
 * workgroup in the same phase and nothing to hide it behind. */
#ifndef QM2_SCAN_DPP
#define QM2_SCAN_DPP 1
#endif
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
#if QM2_SCAN_DPP
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   /* row_shr:1 (lanes without a source add 0) */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   /* row_bcast:15 into rows 1 and 3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   /* row_bcast:31 into rows 2 and 3 */
#else
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t v = __shfl_up(x, o);
        if (lane >= o) x += v;
    }
#endif
    return x;
}

/* Sort one chunk by child in LDS.  Every thread holds kKpt k-mers km[j] with
 * tag[j] = (child << 16) | rank_in_child (rank from an LDS atomic on hist); empty slots carry
 * child id kMaxFan (trash bin), counted in L.hist[cur].  On return L.sorted holds the chunk grouped by child and L.dl
 * says where every element goes; the copy-out (copy_piece) is left to the caller, which
 * interleaves it with the NEXT chunk's compute so that the stores (HBM-bound, ~2 ms per 10 GB)
 * drain behind VALU work instead of stalling every wave at once. */
/* (diagnostic build, -DQM2_STAMPS -DQM2_STAMPS_SORT=1: the phases of sort_chunk inside split_text_kernel) */
#ifdef QM2_STAMPS
#define QM2_SORT_STAMP_ARGS , unsigned long long *st_acc, unsigned long long &st_last, bool st_on
#define QM2_SORT_STAMP(I) do { if (st_on) QM2_STAMP(I); } while (0)
#else
#define QM2_SORT_STAMP_ARGS
#define QM2_SORT_STAMP(I) do { } while (0)
#endif
template <int SYNC>
__device__ __forceinline__ void sort_chunk(SplitLds &L, uint32_t cur, const uint64_t (&km)[kKpt], const uint32_t (&tag)[kKpt],
                                           uint64_t child0, uint32_t fan, const SplitDst &d,
                                           unsigned long long &lost QM2_SORT_STAMP_ARGS)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    lds_barrier();                 /* hist complete; every wave is done reading the previous L.sorted / L.dl */
    QM2_SORT_STAMP(21);
    if (tid == 0) { L.spill_n = 0; L.spill_need = 0; }
    /* thread t takes children t, t + kSplitThreads, ... (kChildPerThread of them: consecutive lanes = consecutive
     * children, so the reservations below are the contiguous, full-rate atomic shape) */
    constexpr int C = kChildPerThread;
    uint32_t mine[C], g[C], off[C];
    const bool holder = tid < kMaxFan / C;
    if constexpr (SYNC == 2)
        for (int i = tid; i <= kMaxFan; i += kSplitThreads) L.hist[cur ^ 1u][i] = 0;  /* the next chunk's counts (last read in the scan before this one) */
    /* exclusive scan of hist[0..fan) */
    constexpr int kScanWaves = (kMaxFan / C) / kWave;          /* waves that hold children */
    uint32_t incl[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int child = tid + i * kSplitThreads;
        mine[i] = holder && child < (int)fan ? L.hist[cur][child] : 0;
        incl[i] = wave_scan_incl(mine[i]);
        if (holder && lane == kWave - 1) L.wsum[i * kScanWaves + wave] = incl[i];
    }
    lds_barrier();
    QM2_SORT_STAMP(22);
    {
        /* sums of the waves before this one: every wave scans the (at most 16) wave totals itself */
        static_assert(kMaxFan / kWave <= kWave, "one lane per wave total");
        const uint32_t ws = wave_scan_incl(lane < kMaxFan / kWave ? L.wsum[lane] : 0u);
#pragma unroll
        for (int i = 0; i < C; i++) {
            const int widx = i * kScanWaves + wave;            /* position of this wave's total for its i-th children */
            const uint32_t before = widx ? (uint32_t)__builtin_amdgcn_readlane((int)ws, __builtin_amdgcn_readfirstlane(widx - 1)) : 0u;   /* (uniform: no LDS round trip) */
            off[i] = before + incl[i] - mine[i];
            g[i] = 0;
            if (holder) {
                const int child = tid + i * kSplitThreads;
                L.off[child] = (uint16_t)off[i];
                /* reserve the run in the child's region; the returned offset is only needed for the
                 * copy-out, so the round trip overlaps the LDS scatter below */
                if (mine[i]) g[i] = atomicAdd(&d.cursors[child0 + (uint64_t)child], mine[i]);
            }
        }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)ws, kMaxFan / kWave - 1);
        if (tid == 0) L.off[kMaxFan] = (uint16_t)tot;
    }
    lds_barrier();
    QM2_SORT_STAMP(23);
    QM2_CHAOS(6);
#pragma unroll
    for (int j = 0; j < kKpt; j++) {
        /* empty slots carry child id kMaxFan: they land behind the chunk's last k-mer, no branch */
        const uint32_t pos = (uint32_t)L.off[tag[j] >> 16] + (tag[j] & 0xFFFFu);
        L.sorted[pos] = km[j];
    }
#ifdef QM2_STAMPS
    if (st_on) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        QM2_STAMP(24);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* the reservations' return (behind everything older of this wave) */
        QM2_STAMP(25);
    }
#endif
    QM2_CHAOS(7);
    uint32_t f[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
        f[i] = mine[i];
        if (holder) {
            if (mine[i]) f[i] = g[i] >= d.cap ? 0u : (d.cap - g[i] < mine[i] ? d.cap - g[i] : mine[i]);
            L.dl[tid + i * kSplitThreads] = make_uint2(g[i] - off[i], off[i] + f[i]);
            if (f[i] < mine[i]) {
                /* rare (skewed input: poly-A, a repeat family): the child is full; the rest of the run goes to the
                 * spill list, which the direct kernel probes at the end of the batch.  A flood puts thousands of
                 * k-mers of every chunk here, so the run is announced and the whole workgroup copies it (below);
                 * one thread doing it alone made 0.5 % poly-A reads cost 30 ms per 50 M reads.  What does not fit the
                 * spill list either (or there is none: qm2_route_device) counts as lost: the caller is told / the
                 * batch is recounted by the direct kernel. */
                const uint32_t need = mine[i] - f[i];                 /* (< 2^16: a chunk holds <= 16 K k-mers) */
                if (d.spill_cap) L.spill_e[atomicAdd(&L.spill_n, 1u)] = make_uint2((off[i] + f[i]) | (need << 16), atomicAdd(&L.spill_need, need));
                else lost += need;
            }
        }
    }
    lds_barrier();
    QM2_SORT_STAMP(26);
    if (L.spill_n) lost += copy_spills(L, d.spill, d.spill_count, d.spill_cap);       /* workgroup-uniform, rare */
}

/* child of a k-mer at a level: `bits` bits of its home slot from bit `shift` on (shift + bits <= log2 H, so
 * the & (H - 1) of :92 is implied; bits = 0: everything is child 0) — one v_bfe_u32 */
__device__ __forceinline__ uint32_t child_of(uint64_t km, uint32_t shift, uint32_t bits)
{
    return __builtin_amdgcn_ubfe(djb32(km), shift, bits);
}

/* two halves so that the caller can put the k-mer's own work between the LDS read and its use */
template <int PIECE>
__device__ __forceinline__ uint64_t copy_fetch(const SplitLds &L)
{
    return L.sorted[threadIdx.x + PIECE * kSplitThreads];
}

/* The child's {offset, limit} is asked for early (copy_lookup) and used late (copy_finish), with the k-mer's own work of
 * the chunk loop in between — as one piece, the LDS read and the wait for it sit back to back, one exposed round trip
 * per k-mer of every lane.  h: the k-mer's hash, which the finest level writes into its records. */
__device__ __forceinline__ uint64_t copy_lookup(const SplitLds &L, uint64_t km, uint32_t shift, uint32_t bits, uint32_t &child, uint32_t &h)
{
    h = djb32(km);
    child = __builtin_amdgcn_ubfe(h, shift, bits);
    return *reinterpret_cast<const uint64_t *>(&L.dl[child]);      /* one 8-byte read: x = low, y = high */
}

/* FIN = 0: the level holds k-mers (8 bytes).  FIN = 2 / 4: it is the finest level and holds bucket records, R in a
 * 4-byte array and the second word in a 2- / 4-byte array beside it (6 instead of 8 bytes per k-mer written here and read
 * by bucket_count, which then needs no hash either). */
template <int PIECE, bool SEG, int FIN>
__device__ __forceinline__ void copy_finish(uint64_t km, uint32_t child, uint32_t h, uint64_t dlw, const SplitDst &d, uint64_t child0)
{
    const uint32_t i = threadIdx.x + PIECE * kSplitThreads;
    /* no bounds test on i: beyond the chunk's last k-mer L.sorted holds stale k-mers, and whatever child such
     * a one hashes to, that child's limit (offset + fit <= the chunk's total <= i) keeps it from being stored */
    const uint32_t dx = (uint32_t)dlw, dy = (uint32_t)(dlw >> 32);
    /* (address-space-1 pointers: a generic one would make this a flat_store, which also counts on
     * lgkmcnt — and every LDS wait and lds_barrier() of these kernels would wait for the stores) */
    typedef uint64_t __attribute__((address_space(1))) global_u64;
    typedef uint32_t __attribute__((address_space(1))) global_u32;
    typedef uint16_t __attribute__((address_space(1))) global_u16;
    if constexpr (FIN != 0) {
        static_assert(!SEG, "the finest level is region-major");
        const uint64_t at = (child0 + child) * (uint64_t)d.cap + (i + dx);
        if (i < dy) {
            ((global_u32 *)(uintptr_t)d.kmers)[at] = rec_r(km);
            const uint32_t w = rec_w(km, h, d.lgH, d.lg_s);
            if constexpr (FIN == 2) ((global_u16 *)(uintptr_t)d.fin_w)[at] = (uint16_t)w;
            else                    ((global_u32 *)(uintptr_t)d.fin_w)[at] = w;
        }
    } else if constexpr (SEG) {
        global_u64 *dst = (global_u64 *)(uintptr_t)d.kmers;
        if (i < dy) dst[slot_of<true>(child0 + child, i + dx, d.cap, d.lg_row)] = km;
    } else {
        global_u64 *dst = (global_u64 *)(uintptr_t)(d.kmers + (child0 + child) * (uint64_t)d.cap);
        if (i < dy) dst[i + dx] = km;
    }
}

template <int PIECE, bool SEG, int FIN>
__device__ __forceinline__ void copy_piece(const SplitLds &L, const SplitDst &d, uint64_t child0, uint32_t shift, uint32_t bits)
{
    const uint64_t km = copy_fetch<PIECE>(L);
    uint32_t child, h;
    const uint64_t dlw = copy_lookup(L, km, shift, bits, child, h);
    copy_finish<PIECE, SEG, FIN>(km, child, h, dlw, d, child0);
}

template <bool SEG, int FIN>
__device__ __forceinline__ void copy_all(const SplitLds &L, const SplitDst &d, uint64_t child0, uint32_t shift, uint32_t bits)
{
#define QM2_CP(J) copy_piece<J, SEG, FIN>(L, d, child0, shift, bits);
    QM2_CP(0) QM2_CP(1) QM2_CP(2) QM2_CP(3) QM2_CP(4) QM2_CP(5) QM2_CP(6) QM2_CP(7)
    QM2_CP(8) QM2_CP(9) QM2_CP(10) QM2_CP(11) QM2_CP(12) QM2_CP(13) QM2_CP(14) QM2_CP(15)
#undef QM2_CP
    static_assert(kKpt == 16, "sixteen pieces per chunk");
}

/* Level 1: sequence text -> canonical k-mers -> 2^b1 buckets by the top bits of
 * the home slot.  One chunk = 16 tiles (one per wave). */
/* `call` (qm2_route_device): the launch's totals {bases, k-mers, k-mers that fitted nowhere} go there
 * instead of into the context's statistics. */
/* FX: 0, or (shift << 8 | bits) as compile-time constants — the split of an hg38-scale table (H = 2^32: 10 + 10 bucket
 * bits above 12 slot bits).  v_bfe_u32 takes one scalar operand; with both in SGPRs hipcc moves one into a VGPR in
 * front of EVERY hash (it has no register to keep it in): two instructions per k-mer. */
/* PK: the text arrives PACKED (qm2_device.hip.h, unpack16): 6 bytes per 16 positions instead of 16, nothing to classify. */
template <bool K30, bool SEG, uint32_t FX = 0, int FIN = 0, bool PK = false>
__global__ __launch_bounds__(kSplitThreads) QM2_SPLIT_OCCUPANCY void split_text_kernel(SeqView sv, uint32_t k, uint32_t shift, uint32_t bits,
                                                                   uint64_t n_chunks, SplitDst d, DevStats *stats,
                                                                   unsigned long long *call)
{
    if constexpr (FX != 0) { shift = FX >> 8; bits = FX & 0xFFu; }
    const uint32_t fan = 1u << bits;
    __shared__ SplitLds L;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    unsigned long long n_bases = 0, n_kmers = 0, lost = 0;
    bool pend = false;          /* (tested per step; running the first chunk's copy-out over an all-refusing L.dl instead, no test:
                                 *  split_text 21.4 -> 23.0 ms, split_kmer 24.6 -> 27.3: hipcc schedules the longer blocks worse) */
    const uint64_t child0 = sub_region0(d, fan);
    /* the 16 bytes of this lane for the NEXT chunk are fetched while the current one is processed */
    auto load_text = [&](uint64_t chunk_) {
        const int64_t g_ = lane_offset((chunk_ * kSplitWaves + wave) * kTileBytes, lane);
        return PK ? load16_packed_raw(sv, g_) : load16_raw(sv, g_);
    };
    uint4 bytes = load_text(blockIdx.x);
#ifdef QM2_STAMPS
    unsigned long long st_acc[28] = {0}, st_last = stamp_now();
#endif
    constexpr int SYNC = QM2_SORT_SYNC_TEXT;
    uint32_t hb = 0;            /* which count array this chunk ranks into (SYNC = 2: they take turns, sort_chunk zeroes the other) */
    if constexpr (SYNC == 2) {
        for (int i = tid; i < 2 * (kMaxFan + 1); i += kSplitThreads) (&L.hist[0][0])[i] = 0;
        lds_barrier();
    }
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint64_t tile = chunk * kSplitWaves + wave;
        const uint4 cur = bytes;
        bytes = load_text(chunk + gridDim.x);
        const Window W = make_window_from<K30, PK>(sv, tile * kTileBytes, lane, k, cur);
        n_bases += W.n_bases;
        n_kmers += __builtin_popcount(W.emit);
        QM2_STAMP(16);
        if constexpr (SYNC == 0) {
            for (int i = tid; i <= kMaxFan; i += kSplitThreads) L.hist[0][i] = 0;
            lds_barrier();
        }
        QM2_STAMP(17);
        uint64_t km[kKpt];
        uint32_t tag[kKpt];
        /* k-mer J of this chunk between the two halves of copy-out piece J of the previous chunk (split_text 21.96 -> 21.56 ms at C3
         * against lookup and use back to back) */
        uint64_t out = pend ? copy_fetch<0>(L) : 0;
#define QM2_T(J)                                                                   \
        {                                                                          \
            /* piece J + 1 is fetched a step ahead; piece J's child is looked up before the k-mer's own work and used behind it */ \
            const uint64_t out_next = pend ? copy_fetch<(J + 1 < kKpt ? J + 1 : J)>(L) : 0; \
            uint32_t oc = 0, oh = 0;                                               \
            const uint64_t dlw = pend ? copy_lookup(L, out, shift, bits, oc, oh) : 0; \
            km[J] = kmer_at<K30, J>(W, k);                                         \
            /* (a select without the branch around the hash was measured: no difference) */ \
            const uint32_t b = (W.emit & (0x8000u >> J)) ? child_of(km[J], shift, bits) : (uint32_t)kMaxFan; \
            tag[J] = (b << 16) | atomicAdd(&L.hist[hb][b], 1u);      /* (a count: < 2^15) */ \
            if (pend) copy_finish<J, SEG, FIN>(out, oc, oh, dlw, d, child0);       \
            out = out_next;                                                        \
        }
        QM2_T(0) QM2_T(1) QM2_T(2) QM2_T(3) QM2_T(4) QM2_T(5) QM2_T(6) QM2_T(7)
        QM2_T(8) QM2_T(9) QM2_T(10) QM2_T(11) QM2_T(12) QM2_T(13) QM2_T(14) QM2_T(15)
#undef QM2_T
        QM2_STAMP(18);
#ifdef QM2_STAMPS
        sort_chunk<SYNC>(L, hb, km, tag, child0, fan, d, lost, st_acc, st_last, QM2_STAMPS_SORT == 1);
#else
        sort_chunk<SYNC>(L, hb, km, tag, child0, fan, d, lost);
#endif
        QM2_STAMP(19);
        hb ^= (uint32_t)(SYNC == 2);
        pend = true;
    }
    if (pend) copy_all<SEG, FIN>(L, d, child0, shift, bits);
#ifdef QM2_STAMPS
    if (lane == 0) {
        for (int i = 16; i < 20; i++) atomicAdd(&qm2_stamp_acc[i], st_acc[i]);
        if (QM2_STAMPS_SORT == 1) { for (int i = 21; i < 27; i++) atomicAdd(&qm2_stamp_acc[i], st_acc[i]); atomicAdd(&qm2_stamp_acc[27], 1ull); }
        atomicAdd(&qm2_stamp_acc[20], 1ull);
    }
#endif
    n_bases = wave_sum(n_bases); n_kmers = wave_sum(n_kmers); lost = wave_sum(lost);
    if (lane == 0) {
        if (call) {
            atomicAdd(&call[0], n_bases);
            atomicAdd(&call[1], n_kmers);
            if (lost) atomicAdd(&call[2], lost);
        } else {
            atomicAdd(&stats->n_bases, n_bases);
            atomicAdd(&stats->n_kmers, n_kmers);
            if (lost) atomicAdd(&stats->part_lost, lost);
        }
    }
}

/* Level l >= 2: split every level l-1 bucket (parent) into its 2^bits children; workgroup x of a parent
 * takes its chunks x, x + wgs_per_parent, ...
 * FLAT (level 1 of a shard, qm2_probe_slabs_device): the parents are the slabs a shard received from its
 * peers — src_cap apart, src_cursors[slab] k-mers in each — and they all feed the same children 0 .. fan-1. */
template <bool FLAT, bool SRC_SEG, bool DST_SEG, uint32_t FX = 0, int FIN = 0>
__global__ __launch_bounds__(kSplitThreads) QM2_SPLIT_OCCUPANCY void split_kmer_kernel(const uint64_t *src, const uint32_t *src_cursors,
                                                                   uint64_t src_cap, uint32_t src_lg_row, uint32_t parent_mask, uint32_t shift, uint32_t bits,
                                                                   uint32_t wgs_per_parent, uint32_t n_parents,
                                                                   SplitDst d, DevStats *stats, uint32_t src_peer_stride = 0)
{
    if constexpr (FX != 0) { shift = FX >> 8; bits = FX & 0xFFu; }
    const uint32_t fan = 1u << bits;
    __shared__ SplitLds L;
    /* Block -> (parent, x).  Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one, each
     * with its own L2): all workgroups of a parent get the same b % 8, so the short runs (~16 k-mers)
     * that different workgroups append to one child meet in ONE L2 and leave it as whole lines.
     * Speed only: nothing depends on where a block really runs. */
    uint64_t parent;
    uint32_t bx;
    if ((n_parents & 7u) == 0 && !FLAT) {
        const uint32_t q = blockIdx.x >> 3, r = blockIdx.x & 7u;
        parent = (uint64_t)(q / wgs_per_parent) * 8u + r;
        bx = q % wgs_per_parent;
    } else { parent = blockIdx.x / wgs_per_parent; bx = blockIdx.x % wgs_per_parent; }
    /* where parent P lies in the source buffer.  Normally region P; a shard's received sub-slabs may lie [peer] x
     * peer_stride + [sub] (the senders' per-XCD sub-regions, x-major: qm2_probe_subslabs_device), parents still
     * numbered peer * 2^src_lg_row + sub */
    uint64_t region = parent;
    if (!FLAT && !SRC_SEG && src_peer_stride)
        region = (parent >> src_lg_row) * (uint64_t)src_peer_stride + (parent & ((1u << src_lg_row) - 1u));
    uint64_t n = src_cursors[region];
    if (n > src_cap) n = src_cap;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    unsigned long long lost = 0;
    /* (parent_mask: the parents are the sub-regions of the level above, see sub_region0) */
    const uint64_t child0 = FLAT ? sub_region0(d, fan) : (parent & parent_mask) * fan;
    bool pend = false;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    /* the next chunk's k-mers are loaded (16 bytes a lane: elements 2 t, 2 t + 1 of each row of 2048) while
     * the current one is sorted and copied out; regions start 16-byte aligned (capacities are even) */
    u32x4 nxt[kKpt / 2];
    auto fetch = [&](uint64_t first) {
#pragma unroll
        for (int j = 0; j < kKpt / 2; j++) {
            const uint64_t at = first + 2u * ((uint64_t)j * kSplitThreads + tid);
            /* (at is even and so is every segment's length: both k-mers of a load lie in one segment) */
            nxt[j] = at < n ? *reinterpret_cast<const u32x4 *>(src + slot_of<SRC_SEG>(region, (uint32_t)at, (uint32_t)src_cap, src_lg_row))
                            : u32x4{~0u, ~0u, ~0u, ~0u};
        }
    };
    fetch((uint64_t)bx * kSplitSlots);
    constexpr int SYNC = QM2_SORT_SYNC_KMER;
    uint32_t hb = 0;
    if constexpr (SYNC == 2) {
        for (int i = tid; i < 2 * (kMaxFan + 1); i += kSplitThreads) (&L.hist[0][0])[i] = 0;
        lds_barrier();
    }
    for (uint64_t first = (uint64_t)bx * kSplitSlots; first < n; first += (uint64_t)wgs_per_parent * kSplitSlots) {
        if constexpr (SYNC == 0) {
            for (int i = tid; i <= kMaxFan; i += kSplitThreads) L.hist[0][i] = 0;
            lds_barrier();
        }
        uint64_t km[kKpt];
        uint32_t tag[kKpt];
#pragma unroll
        for (int j = 0; j < kKpt / 2; j++) {
            const uint64_t at = first + 2u * ((uint64_t)j * kSplitThreads + tid);
            km[2 * j] = ((uint64_t)nxt[j].y << 32) | nxt[j].x;
            km[2 * j + 1] = at + 1 < n ? ((uint64_t)nxt[j].w << 32) | nxt[j].z : kPad;   /* (an odd fill: the pair's second half is stale) */
        }
        fetch(first + (uint64_t)wgs_per_parent * kSplitSlots);
        uint64_t out = pend ? copy_fetch<0>(L) : 0;
#define QM2_K(J)                                                                   \
        {                                                                          \
            const uint64_t out_next = pend ? copy_fetch<(J + 1 < kKpt ? J + 1 : J)>(L) : 0; \
            uint32_t oc = 0, oh = 0;                                               \
            const uint64_t dlw = pend ? copy_lookup(L, out, shift, bits, oc, oh) : 0; \
            const uint32_t b = km[J] != kPad ? child_of(km[J], shift, bits) : (uint32_t)kMaxFan; \
            tag[J] = (b << 16) | atomicAdd(&L.hist[hb][b], 1u);      /* (a count: < 2^15) */ \
            if (pend) copy_finish<J, DST_SEG, FIN>(out, oc, oh, dlw, d, child0);   \
            out = out_next;                                                        \
        }
        QM2_K(0) QM2_K(1) QM2_K(2) QM2_K(3) QM2_K(4) QM2_K(5) QM2_K(6) QM2_K(7)
        QM2_K(8) QM2_K(9) QM2_K(10) QM2_K(11) QM2_K(12) QM2_K(13) QM2_K(14) QM2_K(15)
#undef QM2_K
#ifdef QM2_STAMPS
        { unsigned long long st_none[1], st_t = 0; sort_chunk<SYNC>(L, hb, km, tag, child0, fan, d, lost, st_none, st_t, false); }
#else
        sort_chunk<SYNC>(L, hb, km, tag, child0, fan, d, lost);
#endif
        hb ^= (uint32_t)(SYNC == 2);
        pend = true;
    }
    if (pend) copy_all<DST_SEG, FIN>(L, d, child0, shift, bits);
    lost = wave_sum(lost);
    if (lane == 0 && lost) atomicAdd(&stats->part_lost, lost);
}

/* Finest level: one bucket of S = 2^LG_S table slots at a time per workgroup, one workgroup per CU.
 *
 * Everything here is in RECORD form (qm2_device.hip.h, "bucket records"): the bucket's k-mers arrive as
 * (R, w) — 32 bits of the k-mer the hash does not determine, and the hash bits the bucket does not say, whose low
 * LG_S bits are the home slot inside the bucket: this kernel never hashes — and the table slice comes from the
 * context's TABLE IMAGE (table_image_kernel), the same two words per slot, 6 instead of 8 bytes.  Two keys of one
 * bucket are equal iff their records are (one to one, given the bucket's hash bits).
 *
 * The slice of a bucket (its S slots + kHalo slots on the side probe walks leave through, i.e.
 * towards the middle of the table, :94-95) is staged in LDS MIRRORED for the upper half of the
 * table, so that in LDS every walk is p, p+1, p+2, ... whatever its direction in the table, and a k-mer's
 * home position is w & (S - 1), XOR (S - 1) in the upper half.  Next to the records and the counter tile sits
 * one FINGERPRINT byte per slot: 0 = empty, odd 1..255 = a function of R, 0xFE = beyond the staged slice.
 *
 * Roles (measured with in-kernel stamps on the one-role kernels of round 1 / early round 2: 18-25 %
 * of a wave's time went into ISSUING the slice loads — the per-CU fetch path takes ~10 B/clk, the
 * issuing wave stalls — and the prefetch registers left the probe code no room to overlap its LDS
 * round trips):
 *   waves 0..3   LOADERS: stream the table slice of the bucket after next into registers, stage the
 *                next bucket's slice (records, fingerprints, old counter values) into the OTHER LDS
 *                buffer, and flush the counter tile of the previous bucket;
 *   waves 4..15  PROBERS: probe the current bucket's k-mers against the current buffer.
 * One workgroup barrier per bucket: behind barrier k the probers are done with bucket k-1 and the
 * loaders have staged bucket k.  A loader thread flushes exactly the LDS quads it later stages — whichever
 * way the two buckets walk: slice_quad — so flush (k-1) -> stage (k+1) into the same buffer needs no second barrier.
 *
 * Probing is cheap per step but DIVERGENT as a per-lane loop (at 53 % fill a k-mer needs 1.75 slots
 * on average, the slowest of a wave's 64 lanes ~12).  So:
 *   A  straight-line, kSet k-mers at a time: the fingerprints of the 4 slots from the home slot on
 *      (two aligned dwords + v_alignbyte) are compared SWAR-wise with the k-mer's own fingerprint;
 *      the first match at or before the first empty slot is verified against the full record (exact:
 *      equal keys have equal fingerprints).  ~90 % of the k-mers end here at 53 % fill.  k-mer 0 (the
 *      reference counts it ON the first empty slot, :96-98) is left to B.
 *   B  the rest (no match and no empty slot among the 4, or a false fingerprint match) are
 *      compacted into a per-wave LDS queue and finished once per bucket in dense rows, eight
 *      fingerprints per step.
 *
 * Counter tile: starts from the old values of the bucket's private interior (0 elsewhere), so the
 * LDS atomics accumulate straight onto them and the interior is stored back plainly; the kHalo slots
 * at either end of the owned range and the staged halo can also receive a neighbour's counts, so
 * there the tile holds deltas that are added atomically.
 *
 * G2 = rows of 2 k-mers per prober thread per group, chosen per launch; two rows share a 16-byte load of R. */
struct CountGeo {
    uint64_t range_lo;      /* first slot of bucket 0 (0, or a shard's own_lo)                   */
    uint64_t lim_lo, lim_hi;/* the slots that are resident: [0, H) or the shard's span          */
    uint64_t half;          /* H >> 1: buckets at or above walk down                            */
};

/* (the two CODE bits above the slot bits of a record's second word, and the rules that set them: qm2_device.hip.h,
 * rec_image_w / rec_staged_w — compiled for the host too and pinned by tests/emul) */

/* fingerprint of a record: odd, 1..255, a function of R alone (the slot bits are what neighbours in a probe window have
 * in common, not what tells them apart); of an empty slot: 0; beyond the staged slice: 0xFE.  One v_dot4_u32_u8 folds
 * R's four bytes with odd weights.  (A single byte of the k-mer as fingerprint was measured in round 2: 25.5 instead of
 * 22.8 ms — the k-mers of a genome are not uniform in any one byte and every false match is a queue entry.) */
__device__ __forceinline__ uint32_t fp_word(uint32_t r) { return __builtin_amdgcn_udot4(r, 0x9D7B3D25u, 0u, false); }
__device__ __forceinline__ uint32_t fp8(uint32_t r) { return (fp_word(r) & 0xFFu) | 1u; }
/* the same in all four bytes of a word */
__device__ __forceinline__ uint32_t fp8x4(uint32_t r) { return __builtin_amdgcn_perm(0u, fp_word(r), 0u) | 0x01010101u; }
constexpr uint32_t kFpPad = 0xFEu;

/* The table image: (R, w) of every resident slot — what bucket_count's loader waves stage, 6 bytes a slot (8 when the
 * second word needs 32 bits) instead of the 8-byte key, with the hashing done once per context instead of once per batch. */
template <typename WT>
__global__ void table_image_kernel(const uint64_t *keys, uint64_t n_slots, uint64_t span_lo, uint32_t hmask, uint32_t lg_s,
                                   uint32_t *img_r, WT *img_w)
{
    const uint32_t lgH = (uint32_t)__builtin_popcount(hmask);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        img_r[i] = key ? rec_r(key) : 0u;
        img_w[i] = (WT)rec_image_w(key, span_lo + i, hmask, lgH, lg_s);
    }
}

/* LW loader waves, the rest of the 16 probe: 4 (round 2 chose 3 or 4 per launch — 40 M reads per launch 22.0 vs 24.4 ms,
 * 50 M 24.5 vs 23.5; with the table image 4 is the faster at every fill, qm2_hip.hip; 2 loader waves: 51 ms, their
 * prefetch registers spill).
 * WT: uint16_t / uint32_t, the width of the records' second word (k-mers and table image alike). */
template <int LG_S, int G2, int LW, typename WT>
__global__ __launch_bounds__(kCountThreads)
void bucket_count_kernel(const uint32_t *recs_r, const WT *recs_w, const uint32_t *cursors, uint32_t cap, TableView tv, CountGeo cg,
                         uint32_t n_buckets_all, const uint32_t *shared, DevStats *stats, uint32_t head_sleep)
{
    if (stats->part_lost != 0) return;       /* this batch is recounted by the direct kernel */
    constexpr int kLoaderWaves = LW, kLoaderThreads = LW * kWave;
    constexpr int kProberThreads = kCountThreads - kLoaderThreads, kProberWaves = kProberThreads / kWave;
    constexpr int S = 1 << LG_S;
    constexpr int kSeg = S + kHalo;
    constexpr int kQueueCap = 128;                    /* per wave and bucket; what does not fit is finished in place */
    constexpr bool W16 = sizeof(WT) == 2;
    constexpr uint32_t kCodeMask = 3u << LG_S;
    __shared__ uint64_t skey[2][kSeg];                /* staged records: R low, second word high */
    __shared__ uint32_t scnt[2][kSeg];
    __shared__ uint32_t sfp[2][kSeg / 4 + 4];         /* one byte per slot + 16 bytes of kFpPad */
    __shared__ uint64_t qk[kProberWaves][kQueueCap];
    __shared__ uint16_t qa[kProberWaves][kQueueCap];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    /* Which buckets a workgroup takes.  Round-robin (b, b + grid, ...) every bucket of a workgroup lies ~10 MB from its
     * last one in each of the six arrays it streams; QM2_COUNT_BLOCKED gives it a run of consecutive buckets instead
     * (measured at C3, same box: 23.73 vs 23.62 ms, and 18.91 vs 18.75 at 10 M reads per launch — no difference: the
     * translation caches are not what this kernel waits for). */
#if QM2_COUNT_BLOCKED
    const uint32_t per_wg = (n_buckets_all + gridDim.x - 1) / gridDim.x;
    const uint32_t stride = 1u, b_first = blockIdx.x * per_wg;
    const uint32_t n_buckets = b_first + per_wg < n_buckets_all ? b_first + per_wg : n_buckets_all;   /* this workgroup's end */
#else
    const uint32_t stride = gridDim.x, b_first = blockIdx.x, n_buckets = n_buckets_all;
#endif

    /* geometry of bucket b: staged slice, walk direction, private interior.  Everything is a
     * multiple of 4 slots (buckets, halo and spans are), so 16-byte accesses never straddle an edge.
     * int_lo / int_len: the interior in slice coordinates (i - int_lo < int_len, unsigned) */
    struct Geo { uint64_t seg_lo; uint32_t seg, int_lo, int_len; bool down; };
    auto geo = [&](uint32_t b) {
        Geo g;
        const uint64_t own_lo = cg.range_lo + ((uint64_t)b << LG_S), own_hi = own_lo + S;
        g.down = (own_lo & cg.half) != 0;                         /* upper half walks down */
        uint64_t lo = own_lo, hi = own_hi;
        if (g.down) lo = own_lo >= cg.lim_lo + kHalo ? own_lo - kHalo : cg.lim_lo;
        else        hi = own_hi + kHalo <= cg.lim_hi ? own_hi + kHalo : cg.lim_hi;
        g.seg_lo = lo;
        g.seg = (uint32_t)(hi - lo);
        /* a long run of occupied slots reaches into this bucket: nothing in it is private */
        const bool sh = shared != nullptr && ((shared[b >> 5] >> (b & 31)) & 1u) != 0;
        g.int_lo = (uint32_t)(own_lo - lo) + kHalo;
        g.int_len = sh ? 0u : (uint32_t)S - 2u * kHalo;
        return g;
    };
    /* The slice quad (4 slots) that LDS quad q holds: the same one, or — the slice of a bucket that walks down is staged
     * mirrored — the one as far from the slice's end.  The thread <-> LDS quad assignment of fetch / stage / flush never
     * depends on the bucket, so a thread only ever stages over counter slots it has flushed itself: with the mapping taken
     * on the LDS side instead (until round 4) the two buckets either side of the table's middle — one walks up, the next
     * one of the same buffer down — had thread A staging over a quad that thread B had not flushed yet, unless the loader
     * waves happened to run in step. */
    auto slice_quad = [](const Geo &g, uint32_t q) { return g.down ? g.seg / 4u - 1u - q : q; };
    auto fill_of = [&](uint32_t b) {
        uint32_t n = b < n_buckets ? cursors[b] : 0;
        return n > cap ? cap : n;
    };

    if (wave < kLoaderWaves) {
        /* ============================ LOADERS ============================ */
        __builtin_amdgcn_s_setprio(2);
        /* four slots per lane and step of everything: 16 bytes of R, 8 (16) of second words, 8 of 2-byte counters */
        constexpr int kSteps = (kSeg / 4 + 4 + kLoaderThreads - 1) / kLoaderThreads;      /* (+ the 16 pad fingerprints) */
        typedef typename std::conditional<W16, u32x2, u32x4>::type wvec;
        u32x4 rv[kSteps];
        wvec  wv[kSteps];
        u32x2 ov[kSteps];                        /* four 16-bit interior counters a step (tv.interior16) */
        auto fetch = [&](uint32_t b) {
            if (b >= n_buckets) return;
            const Geo g = geo(b);
            /* (g.seg_lo is a multiple of 4 slots: 16 / 8 / 8-byte aligned in the three arrays) */
            const u32x4 *rp = reinterpret_cast<const u32x4 *>(tv.img_r + (g.seg_lo - tv.span_lo));
            const wvec *wp = reinterpret_cast<const wvec *>(reinterpret_cast<const WT *>(tv.img_w) + (g.seg_lo - tv.span_lo));
            const u32x2 *cp2 = reinterpret_cast<const u32x2 *>(tv.interior16 + (g.seg_lo - tv.span_lo));
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                /* LDS quad i4 (positions 4 i4 .. 4 i4 + 3) is this thread's in EVERY bucket; a bucket that walks down is staged
                 * mirrored, so there the quad holds the slice's quad seg / 4 - 1 - i4, reversed (slice_quad) */
                const uint32_t i4 = tid + st * kLoaderThreads;
                const bool in = 4 * i4 < g.seg;
                const uint32_t gq = slice_quad(g, i4);
                rv[st] = in ? rp[gq] : u32x4{0, 0, 0, 0};
                if constexpr (W16) wv[st] = in ? wp[gq] : wvec{0, 0};
                else               wv[st] = in ? wp[gq] : wvec{0, 0, 0, 0};
                ov[st] = in && 4 * gq - g.int_lo < g.int_len ? cp2[gq] : u32x2{0, 0};
            }
        };
        /* registers -> LDS buffer `bf`: records, fingerprints, counter tile (fetch has already taken the mirror image of a bucket that walks down) */
        auto stage = [&](uint32_t b, int bf) {
            const Geo g = geo(b);
            /* the code a key that a k-mer of THIS bucket can equal carries in the image: 0 in the bucket's own slots,
             * in its halo (the next bucket's slots, walking up; the previous one's, walking down) 1 resp. 2 */
            const uint32_t own_from = g.down ? g.seg - (uint32_t)S : 0u;          /* own slots: [own_from, own_from + S) */
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const uint32_t i = 4u * (tid + st * kLoaderThreads);          /* LDS position */
                if (i < g.seg) {
                    const bool halo = 4u * slice_quad(g, i >> 2) - own_from >= (uint32_t)S;
                    uint32_t w[4];
                    if constexpr (W16) { w[0] = wv[st].x & 0xFFFFu; w[1] = wv[st].x >> 16; w[2] = wv[st].y & 0xFFFFu; w[3] = wv[st].y >> 16; }
                    else               { w[0] = wv[st].x; w[1] = wv[st].y; w[2] = wv[st].z; w[3] = wv[st].w; }
                    const uint32_t r[4] = {rv[st].x, rv[st].y, rv[st].z, rv[st].w};
                    uint32_t f[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        w[j] = rec_staged_w(w[j], halo, g.down, LG_S);
                        f[j] = rec_staged_empty(w[j], LG_S) ? 0u : fp8(r[j]);
                    }
                    if (g.down) {                                              /* (the slice quad's slots 3 .. 0 at i .. i + 3) */
                        *reinterpret_cast<u32x4 *>(&skey[bf][i]) = u32x4{r[3], w[3], r[2], w[2]};
                        *reinterpret_cast<u32x4 *>(&skey[bf][i + 2]) = u32x4{r[1], w[1], r[0], w[0]};
                        sfp[bf][i >> 2] = f[3] | (f[2] << 8) | (f[1] << 16) | (f[0] << 24);
                    } else {
                        *reinterpret_cast<u32x4 *>(&skey[bf][i]) = u32x4{r[0], w[0], r[1], w[1]};
                        *reinterpret_cast<u32x4 *>(&skey[bf][i + 2]) = u32x4{r[2], w[2], r[3], w[3]};
                        sfp[bf][i >> 2] = f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24);
                    }
                    const u32x4 o = u32x4{ov[st].x & 0xFFFFu, ov[st].x >> 16, ov[st].y & 0xFFFFu, ov[st].y >> 16};
                    *reinterpret_cast<u32x4 *>(&scnt[bf][i]) = g.down ? u32x4{o.w, o.z, o.y, o.x} : o;
                } else if (i < (uint32_t)kSeg + 16u) {
                    sfp[bf][i >> 2] = kFpPad * 0x01010101u;                /* beyond the slice */
                }
            }
        };
        /* counter tile of LDS buffer `bf` -> table */
        auto flush = [&](uint32_t b, int bf) {
            /* (a memory round trip at the head of the loaders' chain of every period — and taking the fill from LDS instead,
             * where the probers can leave it, made the kernel SLOWER, 23.6 -> 24.4 ms: the loaders are better off late, see
             * head_sleep below) */
            if (fill_of(b) == 0) return;                 /* nothing was counted: the tile is what was loaded */
            const Geo g = geo(b);
            uint32_t *const cp = tv.counters + (g.seg_lo - tv.span_lo);
            uint16_t *const cp16 = tv.interior16 + (g.seg_lo - tv.span_lo);
#pragma unroll
            for (int st = 0; st < kSteps; st++) {
                const uint32_t lp = 4u * (tid + st * kLoaderThreads);         /* LDS position: the quad this thread staged */
                if (lp < g.seg) {
                    const uint32_t i = 4u * slice_quad(g, lp >> 2);             /* ... and where it lies in the slice */
                    u32x4 cv = *reinterpret_cast<const u32x4 *>(&scnt[bf][lp]);
                    if (g.down) cv = u32x4{cv.w, cv.z, cv.y, cv.x};
                    if (i - g.int_lo < g.int_len) {
                        /* private interior: the low halves go back where they came from; what has outgrown 16 bits (a
                         * k-mer seen more than 65535 times: :291 wraps there, this keeps counting) moves to the wide array */
                        *reinterpret_cast<u32x2 *>(cp16 + i) = u32x2{(cv.x & 0xFFFFu) | (cv.y << 16), (cv.z & 0xFFFFu) | (cv.w << 16)};
                        if ((cv.x | cv.y | cv.z | cv.w) >> 16) {
                            if (cv.x >> 16) atomicAdd(cp + i, cv.x & 0xFFFF0000u);
                            if (cv.y >> 16) atomicAdd(cp + i + 1, cv.y & 0xFFFF0000u);
                            if (cv.z >> 16) atomicAdd(cp + i + 2, cv.z & 0xFFFF0000u);
                            if (cv.w >> 16) atomicAdd(cp + i + 3, cv.w & 0xFFFF0000u);
                        }
                    } else {
                        if (cv.x) atomicAdd(cp + i, cv.x);
                        if (cv.y) atomicAdd(cp + i + 1, cv.y);
                        if (cv.z) atomicAdd(cp + i + 2, cv.z);
                        if (cv.w) atomicAdd(cp + i + 3, cv.w);
                    }
                }
            }
        };
        uint32_t b = b_first;
        if (b < n_buckets) { fetch(b); stage(b, 0); }
        fetch(b + stride);
        lds_barrier();                                            /* barrier 0: bucket 0 is staged */
#ifdef QM2_STAMPS
        unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = stamp_now();
#endif
        uint32_t k = 0;
        for (; b < n_buckets; b += stride, k++) {
#ifdef QM2_TEST_LOADER_SKEW
            /* test build only (tests/test_gpu_parity.py::test_loader_waves_out_of_step): the last loader wave runs ~27 us
             * late, so that the others have staged the bucket after next before it flushes the one before — which must
             * not matter (slice_quad) */
            if (wave == kLoaderWaves - 1) for (int z = 0; z < QM2_TEST_LOADER_SKEW; z++) __builtin_amdgcn_s_sleep(127);
#endif
            QM2_CHAOS(2);
            /* The loaders hold back at the head of a period (head_sleep x 256 cycles; the host picks it, launch_partitioned):
             * their flush and stage are LDS traffic, and behind the barrier every prober wave starts its first probe set at
             * once — while a loader chain that is shorter than the probers' period has the time to give. */
            for (uint32_t z = 0; z < head_sleep; z++) __builtin_amdgcn_s_sleep(4);
            if (k) flush(b - stride, (int)((k + 1u) & 1u));       /* bucket k-1: the probers left it at barrier k */
            QM2_CHAOS(3);
            QM2_STAMP(0);
#ifdef QM2_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QM2_STAMP(10);
#endif
            if (b + stride < n_buckets) stage(b + stride, (int)((k + 1u) & 1u));
            QM2_CHAOS(4);
            QM2_STAMP(1);
            fetch(b + 2u * stride);
            QM2_STAMP(2);
            lds_barrier();                                        /* barrier k+1 */
            QM2_STAMP(3);
        }
        if (k) flush(b - stride, (int)((k + 1u) & 1u));
#ifdef QM2_STAMPS
        if (lane == 0) {
            for (int i = 0; i < 4; i++) atomicAdd(&qm2_stamp_acc[i], st_acc[i]);
            atomicAdd(&qm2_stamp_acc[10], st_acc[10]);
            atomicAdd(&qm2_stamp_acc[8], 1ull);
        }
#endif
        return;
    }

    /* ============================ PROBERS ============================ */
    const int pt = tid - kLoaderThreads, pw = wave - kLoaderWaves;
    uint32_t n_probes = 0;                  /* per lane: < 2^32 for any batch that fits the plan */
    uint32_t probes8 = 0;                   /* phase A's share of it, times 8 (every term a multiple of 8) */
    uint32_t n_hits = 0;                    /* phase A: per WAVE (scalar popcounts); lane 0 .. 63 add their own walk() hits */
    uint32_t w_hits_a = 0;
    constexpr uint32_t kGroupKmers = 2u * G2 * kProberThreads;
    constexpr int kSet = QM2_COUNT_SET;               /* k-mers probed together in phase A (even) */
    /* A group's records: rows 2q, 2q + 1 of a thread are the four k-mers e .. e + 3 of one 16-byte load of R (and one
     * 8- / 16-byte load of second words), e = 4 (q T + pt); an odd last row is two k-mers of an 8-byte load behind them
     * (regions start 128-byte aligned: capacities are multiples of 64).
     * UNCONDITIONAL loads: lanes beyond the fill re-read the bucket's first bytes (a hot line; what
     * they get is ignored, `live` is decided from e < n at use).  Conditional loads make hipcc wait
     * with vmcnt(0) at the first use of a k-mer — i.e. for the NEXT group's prefetch as well. */
    constexpr int kQuads = G2 / 2, kOdd = G2 & 1;
    typedef typename std::conditional<W16, u32x2, u32x4>::type w4vec;
    typedef typename std::conditional<W16, uint32_t, u32x2>::type w2vec;
    struct Group { u32x4 r4[kQuads ? kQuads : 1]; w4vec w4[kQuads ? kQuads : 1]; u32x2 r2; w2vec w2; };
    Group nq;
    auto load_group = [&](const uint32_t *src_r, const WT *src_w, uint32_t first, uint32_t n_src) {
#pragma unroll
        for (int q = 0; q < kQuads; q++) {
            const uint32_t e = first + 4u * (q * kProberThreads + pt);
            const uint32_t at = e < n_src ? e : 0u;
            nq.r4[q] = *reinterpret_cast<const u32x4 *>(src_r + at);
            nq.w4[q] = *reinterpret_cast<const w4vec *>(src_w + at);
        }
        if constexpr (kOdd != 0) {
            const uint32_t e = first + 4u * kQuads * kProberThreads + 2u * pt;
            const uint32_t at = e < n_src ? e : 0u;
            nq.r2 = *reinterpret_cast<const u32x2 *>(src_r + at);
            nq.w2 = *reinterpret_cast<const w2vec *>(src_w + at);
        }
    };
    /* k-mer 0 (poly-A): the reference counts it ON the first empty slot behind its home slot (:96-98).  Its record in the
     * one bucket that can hold it; a value no record equals (code bits set) everywhere else */
    const uint32_t lgH = (uint32_t)__builtin_popcount(tv.hmask);
    const uint32_t h_zero = 0x7BB34A05u;                                          /* djb32(0) */
    const uint64_t home_zero = h_zero & tv.hmask;
    const uint64_t rec_zero = (uint64_t)rec_w(0, h_zero, lgH, LG_S) << 32;        /* rec_r(0) = 0 */
    uint32_t n_next = fill_of(b_first);                       /* fill of the bucket about to start */
    if (b_first < n_buckets) load_group(recs_r + (uint64_t)b_first * cap, recs_w + (uint64_t)b_first * cap, 0, n_next);
    uint32_t n_after = fill_of(b_first + stride);             /* ... and of the one after it */
    lds_barrier();                                            /* barrier 0 */
#ifdef QM2_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = stamp_now();
#endif

    uint32_t k = 0;
    for (uint32_t b = b_first; b < n_buckets; b += stride, k++) {
        const uint32_t n = n_next;
        const Geo g = geo(b);
        const uint64_t seg_lo = g.seg_lo;
        const uint32_t seg = g.seg;
        const bool down = g.down;
        const int bf = (int)(k & 1u);
        const uint64_t *const bkey = skey[bf];
        uint32_t *const bcnt = scnt[bf];
        const uint32_t *const bfp = sfp[bf];
        /* LDS position of the home slot with bucket offset o: o going up (the slice starts at the bucket's first slot),
         * S - 1 - o going down (it ends behind its last) */
        const uint32_t xs = down ? (uint32_t)(S - 1) : 0u;
        const uint64_t own_lo = cg.range_lo + ((uint64_t)b << LG_S);
        const uint64_t zrec = home_zero >> LG_S == own_lo >> LG_S ? rec_zero : ~0ull;
        const uint32_t bn = b + stride;
        n_next = n_after;
        n_after = fill_of(bn + stride);
        /* (a scalar load in flight makes every LDS wait of the probe code an lgkmcnt(0): drain it here) */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        /* Decision on the 4 slots from position p on, from their fingerprint word w (slot p in the low
         * byte).  lm / le: bit index of the first fingerprint match / first empty slot, 0xFFFFFFFF if
         * none (the LOWEST flagged byte of the borrow trick is exact).
         *   lm <= le: a candidate at slot c = lm >> 3 — to be verified against the record; this also covers
         *             "neither" (lm = le = none): c reads as 3, the record there cannot be equal (its
         *             fingerprint is not), and the walk goes on behind slot 3, as it should;
         *   lm >  le: an empty slot comes first: not in the dictionary (:96-98) — except for k-mer 0,
         *             which the reference counts ON that empty slot; k-mer 0 is left to walk().
         * Slots the reference has inspected by then: min(c, e, 3) + 1. */
#define QM2_WINDOW(W, KM, LM, LE)                                                            \
        {                                                                                    \
            const uint32_t x_ = (W) ^ fp8x4((uint32_t)(KM));                                 \
            const uint32_t zm_ = (x_ - 0x01010101u) & ~x_ & 0x80808080u;                     \
            const uint32_t ze_ = ((W) - 0x01010101u) & ~(W) & 0x80808080u;                   \
            LM = (uint32_t)__ffs((int)zm_) - 1u;                                             \
            LE = (uint32_t)__ffs((int)ze_) - 1u;                                             \
        }
        /* the same over EIGHT slots (two fingerprint words): bit indices 0..63 */
#define QM2_WINDOW8(W0, W1, KM, LM, LE)                                                      \
        {                                                                                    \
            const uint32_t f_ = fp8x4((uint32_t)(KM));                                       \
            const uint32_t x0_ = (W0) ^ f_, x1_ = (W1) ^ f_;                                 \
            const uint64_t zm_ = ((uint64_t)((x1_ - 0x01010101u) & ~x1_ & 0x80808080u) << 32) | ((x0_ - 0x01010101u) & ~x0_ & 0x80808080u); \
            const uint64_t ze_ = ((uint64_t)(((W1) - 0x01010101u) & ~(W1) & 0x80808080u) << 32) | (((W0) - 0x01010101u) & ~(W0) & 0x80808080u); \
            LM = (uint32_t)__ffsll((long long)zm_) - 1u;                                     \
            LE = (uint32_t)__ffsll((long long)ze_) - 1u;                                     \
        }
        /* finish a walk that stands at staged position p; adds the slots it inspects from there on.
         * Beyond the slice — only behind a run of >= kHalo occupied slots — it goes on in global
         * memory, against the keys themselves (their records are made on the spot: rare). */
        auto walk = [&](uint64_t km, uint32_t p) {
            uint32_t seen = 0;
            const bool zero = km == zrec;
            for (;;) {
                if (p + 8u > seg || zero) {
                    for (;;) {
                        bool hit, empty;
                        uint32_t *ctr = nullptr;
                        if (p < seg) {
                            const uint64_t v = bkey[p];
                            empty = ((uint32_t)(v >> 32) & kCodeMask) == kCodeMask;
                            hit = zero ? empty : v == km;
                        } else {
                            const uint64_t slot = (down ? seg_lo + seg - 1u - (uint64_t)p : seg_lo + p) - tv.span_lo;
                            const uint64_t v = tv.keys[slot];
                            ctr = tv.counters + slot;
                            empty = v == 0;
                            const uint32_t hv = djb32(v);
                            hit = zero ? empty
                                       : !empty && rec_r(v) == (uint32_t)km && rec_w(v, hv, lgH, LG_S) == (uint32_t)(km >> 32) &&
                                         (hv & tv.hmask) >> LG_S == (uint32_t)(own_lo >> LG_S);
                        }
                        seen++;
                        if (hit) {                                          /* :98, :291 */
                            if (ctr) atomicAdd(ctr, 1u); else atomicAdd(&bcnt[p], 1u);
                            n_hits++;
                            break;
                        }
                        if (empty) break;
                        p++;                                                /* :97 */
                    }
                    break;
                }
                uint32_t lm, le;
                /* eight fingerprints a step: a wave's queue is finished when its LONGEST walk is, and every step
                 * is three dependent LDS round trips */
                const uint32_t f0 = bfp[p >> 2], f1 = bfp[(p >> 2) + 1], f2 = bfp[(p >> 2) + 2];
                const uint32_t w = __builtin_amdgcn_alignbyte(f1, f0, p), w1 = __builtin_amdgcn_alignbyte(f2, f1, p);
                QM2_WINDOW8(w, w1, km, lm, le)
                const uint32_t c = __builtin_amdgcn_ubfe(lm, 3, 3);
                seen += (min(min(lm, le), 63u) >> 3) + 1u;
                if (lm > le) break;                                         /* an empty slot first */
                if (lm <= 63u && bkey[p + c] == km) { atomicAdd(&bcnt[p + c], 1u); n_hits++; break; }
                p += c + 1u;
            }
            n_probes += seen;
        };

        const uint32_t *in_r = recs_r + (uint64_t)b * cap, *in_r_next = recs_r + (uint64_t)bn * cap;
        const WT *in_w = recs_w + (uint64_t)b * cap, *in_w_next = recs_w + (uint64_t)bn * cap;
        uint32_t queued = 0;                                                /* wave-uniform */
        /* ---- phase A: K k-mers at a time, four slots each, no loop and no branch until the queue
         * pushes: all fingerprint reads are issued before the first is used, then all record reads,
         * then the compares and the (unconditional: + 0 or + 1) counter adds, so the LDS round trips
         * of the K k-mers overlap instead of adding up.  Idle lanes (beyond the bucket's fill) hold
         * a real k-mer of the bucket (load_group) and run along; only their effects are masked. ---- */
        auto probe_set = [&](const auto &km, const auto &live) {
            constexpr int K = (int)(sizeof(km) / sizeof(km[0]));
            uint32_t pp[K], d0[K], d1[K], lm[K], le[K];
            uint64_t vv[K];
            constexpr uint32_t kWinBits = 2u, kWinLast = 31u;
#pragma unroll
            for (int j = 0; j < K; j++) {
                pp[j] = ((uint32_t)(km[j] >> 32) & (uint32_t)(S - 1)) ^ xs;   /* :92, done by the level above */
                d0[j] = bfp[pp[j] >> 2];
                d1[j] = bfp[(pp[j] >> 2) + 1];
            }
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t w = __builtin_amdgcn_alignbyte(d1[j], d0[j], pp[j]);   /* fingerprints of p .. p+3 */
                QM2_WINDOW(w, km[j], lm[j], le[j])
                vv[j] = bkey[pp[j] + __builtin_amdgcn_ubfe(lm[j], 3, kWinBits)];
            }
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t c = __builtin_amdgcn_ubfe(lm[j], 3, kWinBits);
                const bool zero = km[j] == zrec;
                const bool on = live[j] & !zero;                            /* k-mer 0: everything is left to walk() */
                const bool cand = lm[j] <= le[j];
                /* (lm <= kWinLast: a real fingerprint match — with none, slot c holds a record of another fingerprint,
                 *  or, in a slice cut short by its span, whatever the buffer held before) */
                /* (& and not &&: with && hipcc sinks the first record read of a set into the branch and waits for it there,
                 *  one exposed LDS round trip per set: bucket_count 26.2 -> 25.5 ms at C3) */
                const bool hit = on & cand & (lm[j] <= kWinLast) & (vv[j] == km[j]);  /* :98 */
                atomicAdd(&bcnt[pp[j] + c], hit ? 1u : 0u);                 /* :291 */
                w_hits_a += __popcll(__ballot(hit));
                /* slots inspected = (first flagged bit >> 3) + 1, and a flagged bit is 8 c + 7: summed as bit + 1,
                 * divided by 8 once per wave at the end */
                probes8 += on ? min(min(lm[j], le[j]), kWinLast) + 1u : 0u;
                /* a false fingerprint match, or nothing decisive in 4 slots: this wave's queue */
                const bool more = live[j] && (zero || (cand && !hit));
                const unsigned long long m = __ballot(more);
                if (m) {
                    const uint32_t slot = queued + __popcll(m & ((1ull << lane) - 1));
                    if (more && slot < (uint32_t)kQueueCap) {               /* (beyond the capacity: the redo pass below) */
                        qk[pw][slot] = km[j];
                        qa[pw][slot] = (uint16_t)(zero ? pp[j] : pp[j] + c + 1u);
                    }
                    queued += __popcll(m);
                }
            }
        };
        uint32_t base = 0;
        /* runs once even for an empty bucket: it carries the next bucket's prefetch */
        while (base == 0 || base < n) {
            const Group kq = nq;
            /* The group about to be probed has to have arrived before the next one is asked for: hipcc gives the prefetch
             * fresh registers and then waits for the group in hand with vmcnt(0) BEHIND the prefetch's loads — i.e. for
             * the prefetch as well, a whole memory round trip per group (bucket_count 25.0 -> 25.9 ms when the records came
             * in).  Waiting here costs nothing: these are the only loads a prober has in flight. */
            __builtin_amdgcn_s_waitcnt(0x0F70);            /* vmcnt(0) — as the builtin: hipcc's own wait insertion reads it, an asm statement it does not */
            QM2_STAMP(11);
            /* next group of this bucket, or behind the last one the first group of the next bucket */
            auto prefetch = [&]() {
                if (base + kGroupKmers >= n) {                               /* workgroup-uniform */
                    if (bn < n_buckets) load_group(in_r_next, in_w_next, 0, n_next); else load_group(in_r, in_w, 0, n_next);
                } else load_group(in_r, in_w, base + kGroupKmers, n);
            };
#if !QM2_PREFETCH_LATE
            prefetch();
#endif
            QM2_STAMP(4);
            /* All G2 rows run as straight-line blocks even when the last ones hold no k-mer
             * (skipping rows was measured slower in round 1): the host picks G2 to fit the fill. */
            auto probe_rows = [&](auto first_c, auto count_c) {
                constexpr int Q0 = decltype(first_c)::value, R = decltype(count_c)::value;
                uint64_t km[2 * R];
                bool live[2 * R];
#pragma unroll
                for (int t = 0; t < R; t++) {
                    const int row = Q0 + t;
                    uint32_t e, r0, r1, w0, w1;
                    if (row < 2 * kQuads) {
                        const int q = row >> 1, hf = row & 1;
                        e = base + 4u * (q * kProberThreads + pt) + 2u * hf;
                        r0 = hf ? kq.r4[q].z : kq.r4[q].x;
                        r1 = hf ? kq.r4[q].w : kq.r4[q].y;
                        if constexpr (W16) {
                            const uint32_t ww = hf ? kq.w4[q].y : kq.w4[q].x;
                            w0 = ww & 0xFFFFu; w1 = ww >> 16;
                        } else {
                            w0 = hf ? kq.w4[q].z : kq.w4[q].x;
                            w1 = hf ? kq.w4[q].w : kq.w4[q].y;
                        }
                    } else {
                        e = base + 4u * kQuads * kProberThreads + 2u * pt;
                        r0 = kq.r2.x; r1 = kq.r2.y;
                        if constexpr (W16) { w0 = kq.w2 & 0xFFFFu; w1 = kq.w2 >> 16; }
                        else               { w0 = kq.w2.x; w1 = kq.w2.y; }
                    }
                    km[2 * t] = ((uint64_t)w0 << 32) | r0;
                    km[2 * t + 1] = ((uint64_t)w1 << 32) | r1;
                    live[2 * t] = e < n;
                    live[2 * t + 1] = e + 1u < n;
                }
                probe_set(km, live);
            };
            constexpr int kR = kSet / 2;               /* rows per set */
#define QM2_SET(I)                                                                                     \
            if constexpr (G2 > (I) * kR)                                                                   \
                probe_rows(std::integral_constant<int, (I) * kR>{},                                        \
                           std::integral_constant<int, (G2 - (I) * kR < kR ? G2 - (I) * kR : kR)>{});
            QM2_SET(0)
#if QM2_PREFETCH_LATE
            prefetch();
#endif
            QM2_SET(1) QM2_SET(2) QM2_SET(3) QM2_SET(4) QM2_SET(5)
#undef QM2_SET
            static_assert(G2 <= 6 * kR, "at most six sets per group");
            QM2_STAMP(5);
            base += kGroupKmers;
        }
        /* ---- phase B: this wave's unresolved k-mers, dense rows, once per bucket ---- */
        QM2_CHAOS(5);
        {
            const uint32_t qn = queued < (uint32_t)kQueueCap ? queued : (uint32_t)kQueueCap;
            for (uint32_t e = lane; e < qn; e += kWave) walk(qk[pw][e], qa[pw][e]);
        }
        /* More unresolved k-mers than the queue holds (skewed input: many copies of a k-mer that sits
         * far from its home slot): go over this wave's k-mers of the bucket again, in the same order,
         * repeat phase A's decision WITHOUT its side effects, and finish in place those that would have
         * got a queue slot beyond the capacity.  Rare, exact, and keeps every walk out of phase A. */
        if (queued > (uint32_t)kQueueCap) {
            uint32_t q2 = 0;
#pragma unroll 1
            for (uint32_t base2 = 0; base2 < n; base2 += kGroupKmers) {
#pragma unroll 1
                for (uint32_t q = 0; q < 2u * G2; q++) {
                    /* (phase A's order: row q >> 1, k-mer q & 1 of it; rows 2 Q, 2 Q + 1 are the halves of quad Q) */
                    const uint32_t row = q >> 1;
                    const uint32_t e = row < 2u * kQuads ? base2 + 4u * ((row >> 1) * kProberThreads + pt) + 2u * (row & 1u) + (q & 1u)
                                                         : base2 + 4u * kQuads * kProberThreads + 2u * pt + (q & 1u);
                    const bool live = e < n;
                    const uint64_t km = live ? ((uint64_t)in_w[e] << 32) | in_r[e] : 0;
                    uint32_t at = 0;
                    bool more = false;
                    if (live) {
                        const uint32_t p = ((uint32_t)(km >> 32) & (uint32_t)(S - 1)) ^ xs;
                        uint32_t lm, le;
                        const uint32_t w = __builtin_amdgcn_alignbyte(bfp[(p >> 2) + 1], bfp[p >> 2], p);
                        QM2_WINDOW(w, km, lm, le)
                        const uint32_t c = __builtin_amdgcn_ubfe(lm, 3, 2);
                        const bool zero = km == zrec;
                        more = zero || (lm <= le && !(lm <= 31u && bkey[p + c] == km));
                        at = zero ? p : p + c + 1u;
                    }
                    const unsigned long long m = __ballot(more);           /* every lane: the same count as phase A's */
                    const uint32_t slot = q2 + __popcll(m & ((1ull << lane) - 1));
                    if (more && slot >= (uint32_t)kQueueCap) walk(km, at);
                    q2 += __popcll(m);
                }
            }
        }
#undef QM2_WINDOW
#undef QM2_WINDOW8
        QM2_STAMP(6);
        lds_barrier();                                                      /* barrier k+1 */
        QM2_STAMP(7);
    }
#ifdef QM2_STAMPS
    if (lane == 0) {
        for (int i = 4; i < 8; i++) atomicAdd(&qm2_stamp_acc[i], st_acc[i]);
        atomicAdd(&qm2_stamp_acc[11], st_acc[11]);
        atomicAdd(&qm2_stamp_acc[9], 1ull);
    }
#endif
    const unsigned long long w_probes = wave_sum((unsigned long long)n_probes + (probes8 >> 3)),
                             w_hits = wave_sum((unsigned long long)n_hits) + w_hits_a;
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
/* keys: the resident slots (slot0 first, n_slots of them); buckets are numbered from slot range_lo */
__global__ void long_run_kernel(const uint64_t *keys, uint64_t n_slots, uint64_t slot0, uint64_t range_lo, uint64_t n_buckets,
                                uint32_t limit, uint32_t lg_s, uint32_t *shared, unsigned int *found)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * blockDim.x) {
        if (keys[i] == 0) continue;
        if (i != 0 && keys[i - 1] != 0) continue;             /* not the start of a run */
        uint64_t len = 1;
        while (len < kRunFollow && i + len < n_slots && keys[i + len] != 0) len++;
        if (len < limit) continue;
        if (len >= kRunFollow) { atomicOr(found, 2u); continue; }
        atomicOr(found, 1u);
        /* the slots before / behind the run included; buckets exist for [range_lo, range_lo + (n_buckets << lg_s)) only */
        const uint64_t s_lo = slot0 + (i ? i - 1 : 0), s_hi = slot0 + (i + len < n_slots ? i + len : n_slots - 1);
        const uint64_t r_hi = range_lo + (n_buckets << lg_s);
        if (s_hi < range_lo || s_lo >= r_hi) continue;
        const uint64_t lo = (s_lo > range_lo ? s_lo - range_lo : 0) >> lg_s, hi = ((s_hi < r_hi ? s_hi : r_hi - 1) - range_lo) >> lg_s;
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

__global__ void part_end_kernel(DevStats *stats, const unsigned long long *spill_count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (stats->part_lost != 0) stats->fallbacks++;
        else stats->spilled += *spill_count;
    }
}

}  // namespace qm2
