/*
 * qm2_hip.hip — the HIP implementation of include/qm2.h (libqm2_hip.so).
 *
 * Host side of the counting phase: device residency of the dictionary
 * (QuicKmer.c:353-366), a pinned staging -> device-batch pipeline that replaces
 * the reference's producer/worker FIFOs (:368-384, :421-439), the choice between
 * the direct kernel and the partitioned pipeline per batch, and the drain
 * (:458-479).  Kernels: qm2_kernels.hip.h (direct, split path) and
 * qm2_partition.hip.h (partitioned pipeline).
 *
 * There is deliberately no CPU implementation behind this ABI: without a
 * usable HIP device every call fails loudly.
 */
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "qm2_ctx.h"
#include "qm2_kernels.hip.h"
#include "qm2_raw.hip.h"
#include "qm2_partition.hip.h"
#include "qm2_plan.h"

using namespace qm2;

/* ---- errors: puts(msg); return 1;  (QuicKmer.c:354-358) -------------------- */

static thread_local char g_err[512] = "";

int qm2_fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    puts(g_err);
    fflush(stdout);
    return 1;
}
#define fail qm2_fail

extern "C" const char *qm2_last_error(void) { return g_err; }
extern "C" uint32_t qm2_abi_version(void) { return QM2_ABI_VERSION; }

namespace {
constexpr size_t kDevPad = 64;
#ifndef QM2_LG_S
#define QM2_LG_S 12
#endif
#ifndef QM2_COUNT_GRID_MULT
#define QM2_COUNT_GRID_MULT 1        /* bucket_count workgroups = 1 per CU (what is resident) x this; 1 / 4 / 8: 22.6 / 22.9 / 23.0 ms */
#endif
constexpr int    kLgS = QM2_LG_S;          /* finest bucket = 2^kLgS table slots: 12 B of LDS per slot */
}  // namespace

static int use_device(qm2_ctx *c)
{
    HIP_OK(hipSetDevice(c->device));
    return 0;
}

static TableView table_view(const qm2_ctx *c)
{
    TableView tv;
    tv.keys = c->keys;
    tv.counters = c->counters;
    tv.interior16 = c->interior16;
    tv.img_r = c->img_r;
    tv.img_w = c->img_w;
    tv.span_lo = c->span_lo;
    tv.hmask = (uint32_t)(c->H - 1);
    tv.hhalf = (uint32_t)(c->H >> 1);
    return tv;
}

static int grid_for(const qm2_ctx *c, uint64_t n_wave_items)
{
    const uint64_t waves_per_block = kBlock / kWave;
    uint64_t blocks = (n_wave_items + waves_per_block - 1) / waves_per_block;
    const uint64_t cap = (uint64_t)c->n_cu * 8;     /* 8 blocks of 256 per CU = 32 waves/CU */
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

/* ---- per-kernel timing (QM2_FLAG_TIMING) ----------------------------------- */

static int time_begin(qm2_ctx *c, TimedLaunch *t, int cls)
{
    if (!(c->flags & QM2_FLAG_TIMING)) return 0;
    if (!c->pool.empty()) {
        *t = c->pool.back();
        c->pool.pop_back();
    } else {
        HIP_OK(hipEventCreate(&t->a));
        HIP_OK(hipEventCreate(&t->b));
    }
    t->cls = cls;
    t->batch_end = false;
    HIP_OK(hipEventRecord(t->a, c->stream));
    return 0;
}

static int time_end(qm2_ctx *c, TimedLaunch *t, bool batch_end = false)
{
    if (!(c->flags & QM2_FLAG_TIMING)) return 0;
    HIP_OK(hipEventRecord(t->b, c->stream));
    t->batch_end = batch_end;
    c->timed.push_back(*t);
    return 0;
}

static int harvest_timing(qm2_ctx *c)
{
    double batch_ms = 0;
    for (auto &t : c->timed) {
        float ms = 0.f;
        HIP_OK(hipEventSynchronize(t.b));
        HIP_OK(hipEventElapsedTime(&ms, t.a, t.b));
        c->prof.kernel_ms += ms;
        c->prof.class_ms[t.cls] += ms;
        c->prof.class_launches[t.cls]++;
        batch_ms += ms;
        if (t.batch_end) {
            c->prof.n_launches++;
            c->prof.last_kernel_ms = batch_ms;
            batch_ms = 0;
        }
        c->pool.push_back(t);
    }
    c->timed.clear();
    return 0;
}

/* ---- partition plan --------------------------------------------------------- */

/* first_bits > 0: level 0 is given from outside with that many bits (qm2_probe_subslabs_device); the levels behind it
 * share the rest */
static void part_init(qm2_ctx *c, int max_bits, int first_bits = 0)
{
    PartPlan &p = c->part;
    p.levels = 0;
    p.ext0 = first_bits > 0;
    p.max_bits = max_bits;
    if (c->own_hi == 0) return;               /* a span without its own range: probed directly */
    p.lg_range = 0;
    while ((1ull << p.lg_range) < c->own_hi - c->own_lo) p.lg_range++;
    const int nb = (int)p.lg_range - kLgS;                           /* bits of the finest bucket id */
    if (nb < 1 || c->lgH <= (uint32_t)kLgS) return;   /* fewer than two buckets: counted by the direct kernel */
    /* the finest level and the table image hold bucket records (qm2_device.hip.h): 6 bytes when the hash bits a bucket
     * does not imply fit 16 bits beside the slot and the two code bits (hg38 scale: H = 2^32), else 8 */
    const uint32_t ext = rec_ext_bits(c->k, c->lgH);
    if ((uint32_t)kLgS + 2u + ext > 32u) return;      /* (fewer than 2^14 slots, 2^16 at k = 31: the direct kernel) */
    c->rec_bytes = (uint32_t)kLgS + 2u + ext <= 16u ? 6u : 8u;
    if (first_bits) {
        if (nb - first_bits < 1) return;                             /* nothing left to split behind the given level */
        p.levels = 1 + (nb - first_bits + max_bits - 1) / max_bits;
    } else p.levels = nb <= max_bits ? 1 : (nb + max_bits - 1) / max_bits;
    if (p.levels > kMaxLevels) { p.levels = 0; return; }             /* cannot happen for H <= 2^32 */
    int left = nb, cum = 0;
    for (int l = 0; l < p.levels; l++) {
        /* the smaller shares go first (the text level is VALU-bound); hg38: 10 + 10 */
        const int b = first_bits && l == 0 ? first_bits : left / (p.levels - l);
        p.bits[l] = (uint32_t)b;
        cum += b;
        left -= b;
        p.shift[l] = p.lg_range - (uint32_t)cum;                      /* >= kLgS */
        p.buckets[l] = 1ull << cum;
    }
}

void qm2_part_free(PartPlan &p)
{
    for (int i = 0; i < 2; i++) { if (p.buf[i]) (void)hipFree(p.buf[i]); p.buf[i] = nullptr; p.buf_slots[i] = 0; }
    if (p.cursors) (void)hipFree(p.cursors);
    if (p.spill) (void)hipFree(p.spill);
    if (p.spill_count) (void)hipFree(p.spill_count);
    p.cursors = nullptr; p.spill = nullptr; p.spill_count = nullptr;
    p.reserved = 0;
}

/* The plan comes in two kinds: level 0 made by this context's own split (first_bits = 0: qm2_submit*,
 * qm2_submit_device, qm2_probe_slabs_device), or given from outside with `first_bits` bits
 * (qm2_probe_subslabs_device: no buffer and no cursors of ours at level 0).  A context may mix the calls; the plan
 * is rebuilt when the kind (or the width of the given level) changes — launching one kind on the other's plan
 * would put level 0's reservations into level 1's cursors and count stale buffer contents. */
static int part_plan_kind(qm2_ctx *c, int first_bits)
{
    PartPlan &p = c->part;
    const bool want_ext = first_bits > 0;
    if (p.ext0 == want_ext && (!want_ext || (p.levels != 0 && p.bits[0] == (uint32_t)first_bits))) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));                          /* buffers may be in use */
    if (p.reserved) c->prof.replans++;                                /* (a plan with buffers is thrown away: qm2_profile) */
    qm2_part_free(p);
    if (p.shared) { (void)hipFree(p.shared); p.shared = nullptr; }
    p.runs = 0;
    p.refused = 0;
    part_init(c, p.max_bits, first_bits);
    return 0;
}

/* Make sure the partition buffers can take a batch of n payload bytes (an upper
 * bound on its k-mers).  Returns 1 with g_err set when memory does not allow:
 * the caller then uses the direct kernel. */
static int part_reserve(qm2_ctx *c, uint64_t n)
{
    PartPlan &p = c->part;
    if (p.levels == 0) return 1;
    if (n <= p.reserved) return 0;
    if (p.refused && n >= p.refused) return 1;                        /* did not fit before: do not try again */
    HIP_OK(hipStreamSynchronize(c->stream));                          /* buffers may be in use */
    const uint64_t want = std::max<uint64_t>(n + n / 16, 1u << 20);   /* grow with some slack */
    qm2_part_free(p);
    /* k-mers expected of `want` bytes (the caller's hint; 1 per byte without one) */
    const double want_kmers = std::max((double)want * c->kappa_hint, 1048576.0);
    double mean = want_kmers;
    uint64_t slots[kMaxLevels] = {0, 0, 0};
    for (int l = 0; l < p.levels; l++) {
        /* every level from the batch's own bound (its bytes bound its k-mers; 150-bp reads use 0.8 of it):
         * expected fill + 12 % for skew (repeated k-mers all land in one child) + the Poisson spread of uniform
         * hashing; what still does not fit goes to the spill list.  (Until late in round 2 each level was sized
         * from the CAPACITY of the one above — 1.12 x 1.22 compounded, 96 instead of 79 GB for the finest level
         * of a 7.55 GB batch — because the smaller stride had measured 10 % slower in split_kmer; with the
         * per-XCD sub-regions above it the difference is 1.5 %: 21.2 vs 20.8 ms.) */
        mean = want_kmers / (double)p.buckets[l];
        const double capd = mean * 1.12 + 8.0 * std::sqrt(mean + 1.0) + 128.0;
        if (capd > 4.0e9) return fail("qm2: batch too large for the partition plan");
        p.cap[l] = ((uint32_t)capd + 63u) & ~63u;          /* whole sectors */
        if (l == 0 && p.ext0) {                                  /* the caller's buffer: nothing of ours */
            p.subs0 = 1; p.seg0 = false; p.cap[0] = 0;
            continue;
        }
        if (l == 0) {
            /* a level that another split reads: one sub-region per XCD and child (sub_region0), each with the
             * Poisson room of its own, smaller, fill */
            p.subs0 = p.levels >= 2 && p.cap[0] >= 8u * 4096u && !getenv("QM2_NO_SUBS") ? 8u : 1u;
            if (p.subs0 > 1) {
                const double m8 = capd / 8.0;
                p.cap[0] = ((uint32_t)(m8 + 8.0 * std::sqrt(m8 + 1.0) + 128.0) + 63u) & ~63u;
            }
            /* segmented on top of that (C3, 50 M reads per launch: split_text 19.2 -> 18.4 ms, split_kmer 21.8 -> 21.4;
             * without the sub-regions it was the slower layout: 27.7 ms): whole segments per region then */
            p.seg0 = p.subs0 > 1 && p.cap[0] >= 4u * kSegKmers && !getenv("QM2_NO_SEG");
            if (p.seg0) p.cap[0] = (p.cap[0] + kSegKmers - 1u) & ~(kSegKmers - 1u);
        }
        slots[l] = p.buckets[l] * (l == 0 ? p.subs0 : 1u) * (uint64_t)p.cap[l];
        /* (the finest level: records of 6 or 8 bytes, two arrays behind each other; counted in 8-byte units here) */
        if (l == p.levels - 1) slots[l] = (slots[l] * c->rec_bytes + 7) / 8;
    }
    p.buf_slots[0] = std::max(slots[0], slots[2]);
    p.buf_slots[1] = slots[1];
    p.n_cursors = 0;
    for (int l = 0; l < p.levels; l++) { p.cursor_off[l] = p.n_cursors; p.n_cursors += l == 0 && p.ext0 ? 0 : p.buckets[l] * (l == 0 ? p.subs0 : 1u); }
    p.spill_cap = std::min<uint64_t>((uint64_t)want_kmers / 4 + 4096, 0xFFFFFFFFull);  /* 2 B of HBM per input byte; places are 32-bit in sort_chunk's announcements */
    hipError_t e = hipSuccess;
    /* (QM2_TIMING: what each allocation took — tens of GB of hipMalloc are the driver's time, seconds on some boxes) */
    const bool timing = getenv("QM2_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char *what, double gb) {
        const double t = now();
        if (timing) printf("QM2 prepare: %s %.1f GB in %.3f s\n", what, gb, t - t_last);
        t_last = t;
    };
    for (int i = 0; i < 2 && e == hipSuccess; i++)
        if (p.buf_slots[i]) { e = hipMalloc((void **)&p.buf[i], p.buf_slots[i] * sizeof(uint64_t)); c->n_allocations++; lap("partition buffer", p.buf_slots[i] * 8e-9); }
    if (e == hipSuccess) e = hipMalloc((void **)&p.cursors, p.n_cursors * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&p.spill, p.spill_cap * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&p.spill_count, sizeof(unsigned long long));
    lap("cursors + spill list", p.spill_cap * 8e-9);
    if (e == hipSuccess && !c->interior16) {
        /* the 2-byte counters of the bucket interiors (qm2_ctx.h): once per context, they outlive a re-plan */
        const uint64_t n_slots = c->span_hi - c->span_lo;
        e = hipMalloc((void **)&c->interior16, n_slots * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemsetAsync(c->interior16, 0, n_slots * sizeof(uint16_t), c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && c->interior16) { (void)hipFree(c->interior16); c->interior16 = nullptr; }
        c->n_allocations++;
        lap("interior counters (cleared)", n_slots * 2e-9);
    }
    if (e == hipSuccess && !c->img_r) {
        /* the table image (qm2_ctx.h): once per context as well; filled behind the look at the runs (runs_fit) */
        const uint64_t n_slots = c->span_hi - c->span_lo;
        e = hipMalloc((void **)&c->img_r, n_slots * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&c->img_w, n_slots * (c->rec_bytes - 4u));
        if (e != hipSuccess) {
            if (c->img_r) (void)hipFree(c->img_r);
            c->img_r = nullptr; c->img_w = nullptr;
        }
        c->img_ready = false;
        c->n_allocations += 2;
        lap("table image", n_slots * (double)c->rec_bytes * 1e-9);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        qm2_part_free(p);
        p.refused = n;
        snprintf(g_err, sizeof g_err, "qm2: partition buffers for %llu bytes do not fit (%s)",
                 (unsigned long long)n, hipGetErrorString(e));
        return 1;
    }
    p.reserved = want;
    if (getenv("QM2_PLAN_PRINT"))
        fprintf(stderr, "QM2 plan for %llu bytes (%.0f k-mers expected): levels %d, bits %u+%u+%u, sub-regions %u%s, cap %u / %u / %u k-mers, buffers %.1f + %.1f GB, spill %.1f GB, %llu cursors\n",
                (unsigned long long)n, want_kmers, p.levels, p.bits[0], p.bits[1], p.bits[2], p.subs0, p.seg0 ? " segmented" : "", p.cap[0], p.cap[1], p.cap[2],
                p.buf_slots[0] * 8e-9, p.buf_slots[1] * 8e-9, p.spill_cap * 8e-9, (unsigned long long)p.n_cursors);
    return 0;
}

/* ---- launching one batch ----------------------------------------------------- */

static void stats_peek_harvest(qm2_ctx *c);

static SeqView seq_view(const uint8_t *data, size_t n, size_t carry, const uint64_t *d_sup, uint64_t n_sup)
{
    SeqView sv;
    const uintptr_t p = (uintptr_t)data;
    sv.lead = p & 15u;
    sv.base = (const uint8_t *)(p - sv.lead);
    sv.emit_from = sv.lead + carry;
    sv.n_total = sv.lead + n;
    sv.suppress = d_sup;
    sv.n_suppress = n_sup;
    sv.packed = 0;
    return sv;
}

/* n positions of packed input (include/qm2.h, "packed sequence"): whole blocks at a 16-byte aligned address */
static SeqView seq_view_packed(const uint8_t *blocks, size_t n_positions, const uint64_t *d_sup, uint64_t n_sup)
{
    SeqView sv;
    sv.base = blocks;
    sv.lead = 0;
    sv.emit_from = 0;
    sv.n_total = n_positions;
    sv.suppress = d_sup;
    sv.n_suppress = n_sup;
    sv.packed = 1;
    return sv;
}

static int launch_direct(qm2_ctx *c, const SeqView &sv, const unsigned long long *only_if, bool seq_stats,
                         bool batch_end)
{
    const uint64_t n_tiles = (sv.n_total + kTileBytes - 1) / kTileBytes;
    const int grid = grid_for(c, n_tiles);
    TimedLaunch t;
    if (time_begin(c, &t, QM2_K_COUNT_DIRECT)) return 1;
#define QM2_DIRECT(K30_, PK_) hipLaunchKernelGGL((count_kernel<K30_, PK_>), dim3(grid), dim3(kBlock), 0, c->stream, sv, table_view(c), c->k, \
                                                n_tiles, c->stats, only_if, seq_stats)
    if (sv.packed) { if (c->k == 30) QM2_DIRECT(true, true); else QM2_DIRECT(false, true); }
    else           { if (c->k == 30) QM2_DIRECT(true, false); else QM2_DIRECT(false, false); }
#undef QM2_DIRECT
    HIP_OK(hipGetLastError());
    return time_end(c, &t, batch_end);
}

/* The source of a partitioned batch: sequence text (qm2_submit*), or the slabs of k-mers a shard received
 * from its peers (qm2_probe_slabs_device). */
struct PartSource {
    const SeqView *sv = nullptr;
    const uint64_t *slabs = nullptr;
    uint64_t slab_cap = 0;
    uint32_t n_slabs = 0;
    const uint32_t *fill = nullptr;
    uint32_t n_subs = 0;            /* > 0: the slabs are [peer][sub][slab_cap]: a shard's level 0, partitioned by its senders */
    uint32_t peer_stride = 0;       /* ... regions from one peer's sub-slab 0 to the next one's (0 = n_subs) */
    uint64_t n_upper = 0;           /* upper bound on the k-mers (text: its sequence bytes) */
    uint64_t n_bytes = 0;           /* bytes launched (what kmers_per_byte is measured against) */
};

/* The split kernels by what their destination is: a level of k-mers (fin = 0; segmented or region-major), or the finest
 * level, written as bucket records with a second word of `fin` = 2 or 4 bytes (always region-major). */
template <bool K30, bool SEG, uint32_t FX, bool PK, typename... A>
static void launch_text_split(int fin, int grid, hipStream_t s, A... a)
{
    if constexpr (FX != 0) split_text_kernel<K30, SEG, FX, 0, PK><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);   /* (a level above the finest) */
    else if (fin == 2)     split_text_kernel<K30, false, 0, 2, PK><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
    else if (fin == 4)     split_text_kernel<K30, false, 0, 4, PK><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
    else                   split_text_kernel<K30, SEG, 0, 0, PK><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
}
/* ... and by what the text is: ASCII, or packed (PK) */
template <bool PK, typename... A>
static void launch_text_split_for(const qm2_ctx *c, const PartPlan &p, int fin, int grid, hipStream_t s, A... a)
{
    /* (the hg38-scale split with its shift and width as constants: split_text 21.57 -> 21.37, split_kmer 24.47 -> 24.17 ms) */
    constexpr uint32_t kFx1 = (22u << 8) | 10u;       /* H = 2^32: level 1 takes bits 22..31 of the home slot */
    if (c->k == 30 && !fin && p.seg0 && p.shift[0] == 22 && p.bits[0] == 10) launch_text_split<true, true, kFx1, PK>(0, grid, s, a...);
    else if (c->k == 30) { if (p.seg0) launch_text_split<true, true, 0, PK>(fin, grid, s, a...); else launch_text_split<true, false, 0, PK>(fin, grid, s, a...); }
    else                 { if (p.seg0) launch_text_split<false, true, 0, PK>(fin, grid, s, a...); else launch_text_split<false, false, 0, PK>(fin, grid, s, a...); }
}
template <bool FLAT, bool SRC_SEG, bool DST_SEG, uint32_t FX, typename... A>
static void launch_kmer_split(int fin, uint32_t grid, hipStream_t s, A... a)
{
    if constexpr (FX != 0) split_kmer_kernel<FLAT, SRC_SEG, false, FX, 2><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...); /* (the finest level of an hg38-scale table) */
    else if (fin == 2)     split_kmer_kernel<FLAT, SRC_SEG, false, 0, 2><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
    else if (fin == 4)     split_kmer_kernel<FLAT, SRC_SEG, false, 0, 4><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
    else                   split_kmer_kernel<FLAT, SRC_SEG, DST_SEG, 0, 0><<<dim3(grid), dim3(kSplitThreads), 0, s>>>(a...);
}

static int launch_partitioned(qm2_ctx *c, const PartSource &src)
{
    PartPlan &p = c->part;
    hipStream_t s = c->stream;
    HIP_OK(hipMemsetAsync(p.cursors, 0, p.n_cursors * sizeof(uint32_t), s));
    hipLaunchKernelGGL(part_begin_kernel, dim3(1), dim3(64), 0, s, c->stats, p.spill_count);
    HIP_OK(hipGetLastError());

    TimedLaunch t;
    const int L = p.levels - 1;                       /* the finest level: bucket records, R array then second words */
    const int fin_w = (int)c->rec_bytes - 4;
    uint32_t *const fin_r = (uint32_t *)p.buf[L & 1];
    void *const fin_w_at = (void *)(fin_r + p.buckets[L] * (uint64_t)p.cap[L]);
    /* level 1: text or slabs -> buckets[0] */
    if (!src.n_subs) {
        uint32_t lg_subs = 0;
        while ((1u << lg_subs) < p.subs0) lg_subs++;
        const int fin = L == 0 ? fin_w : 0;
        SplitDst d = {p.buf[0], p.cursors + p.cursor_off[0], p.cap[0], p.spill, p.spill_count, p.spill_cap,
                      p.bits[0] + lg_subs + kSegLg, p.subs0, fin ? fin_w_at : nullptr, c->lgH, (uint32_t)kLgS};
        if (src.sv) {
            const SeqView &sv = *src.sv;
            const uint64_t n_chunks = (sv.n_total + kSplitTextBytes - 1) / kSplitTextBytes;
            const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * kSplitWgPerCu);
            if (time_begin(c, &t, QM2_K_SPLIT_TEXT)) return 1;
#define QM2_TEXT_ARGS sv, c->k, p.shift[0], p.bits[0], n_chunks, d, c->stats, (unsigned long long *)nullptr
            if (sv.packed) launch_text_split_for<true>(c, p, fin, grid, s, QM2_TEXT_ARGS);
            else           launch_text_split_for<false>(c, p, fin, grid, s, QM2_TEXT_ARGS);
#undef QM2_TEXT_ARGS
        } else {
            const uint32_t chunks = (uint32_t)std::max<uint64_t>(1, ((src.slab_cap + kSplitSlots - 1) / kSplitSlots + 3) / 4);
            if (time_begin(c, &t, QM2_K_SPLIT_KMER)) return 1;
#define QM2_FLAT_ARGS src.slabs, src.fill, src.slab_cap, 0u, ~0u, p.shift[0], p.bits[0], chunks, src.n_slabs, d, c->stats, 0u
            if (p.seg0) launch_kmer_split<true, false, true, 0>(fin, chunks * src.n_slabs, s, QM2_FLAT_ARGS);
            else        launch_kmer_split<true, false, false, 0>(fin, chunks * src.n_slabs, s, QM2_FLAT_ARGS);
#undef QM2_FLAT_ARGS
        }
        HIP_OK(hipGetLastError());
        if (time_end(c, &t)) return 1;
    }
    /* deeper levels: buckets[l-1] -> buckets[l], ping-pong between the two buffers */
    for (int l = 1; l < p.levels; l++) {
        const int fin = l == L ? fin_w : 0;
        SplitDst d = {p.buf[l & 1], p.cursors + p.cursor_off[l], p.cap[l], p.spill, p.spill_count, p.spill_cap, 0, 1,
                      fin ? fin_w_at : nullptr, c->lgH, (uint32_t)kLgS};
        const uint32_t n_parents = (uint32_t)p.buckets[l - 1] * (l == 1 ? p.subs0 : 1u);
        uint32_t lg_subs = 0;
        while ((1u << lg_subs) < p.subs0) lg_subs++;
        /* ~2 chunks of the parent's CAPACITY per workgroup (a parent is ~2/3 full, so most workgroups take one or two):
         * C3, 60 M reads per launch, same box: 1 / 2 / 3 / 4 / 8 / 16 -> 25.6 / 24.4 / 24.6 / 25.7-26.4 / 28.4 / 28.3 ms
         * (profiles/r03_c3/split_kmer_chunks_per_workgroup.txt); no difference at 50 M reads or at chr22 scale */
        static const uint32_t per_wg = getenv("QM2_SPLIT_CHUNKS_PER_WG") ? std::max(1, atoi(getenv("QM2_SPLIT_CHUNKS_PER_WG"))) : 2u;
        const uint32_t chunks = std::max<uint32_t>(1u, ((p.cap[l - 1] + kSplitSlots - 1) / kSplitSlots + per_wg - 1) / per_wg);
        if (time_begin(c, &t, QM2_K_SPLIT_KMER)) return 1;
#define QM2_LEVEL_ARGS p.buf[(l - 1) & 1], p.cursors + p.cursor_off[l - 1], (uint64_t)p.cap[l - 1], p.bits[0] + lg_subs + kSegLg, \
                       (uint32_t)p.buckets[l - 1] - 1u, p.shift[l], p.bits[l], chunks, n_parents, d, c->stats, 0u
        if (l == 1 && src.n_subs) {
            /* the received sub-slabs are the parents: region (peer, sub) feeds the children of sub — the layout of the
             * per-XCD sub-regions with the sender in the XCD's place */
            const uint32_t np = src.n_slabs * src.n_subs;
            const uint32_t ch = std::max<uint32_t>(1u, (uint32_t)(((src.slab_cap + kSplitSlots - 1) / kSplitSlots + 3) / 4));
            uint32_t lg_ns = 0;
            while ((1u << lg_ns) < src.n_subs) lg_ns++;
            launch_kmer_split<false, false, false, 0>(fin, ch * np, s, src.slabs, src.fill, src.slab_cap, lg_ns, src.n_subs - 1u, p.shift[l], p.bits[l], ch, np, d,
                                                      c->stats, src.peer_stride && src.peer_stride != src.n_subs ? src.peer_stride : 0u);
        } else if (l == 1 && p.seg0 && fin == 2 && p.shift[l] == 12 && p.bits[l] == 10) {
            constexpr uint32_t kFx2 = (12u << 8) | 10u;       /* ... level 2 bits 12..21: the finest level of an hg38-scale table */
            launch_kmer_split<false, true, false, kFx2>(2, chunks * n_parents, s, QM2_LEVEL_ARGS);
        } else if (l == 1 && p.seg0) launch_kmer_split<false, true, false, 0>(fin, chunks * n_parents, s, QM2_LEVEL_ARGS);
        else                         launch_kmer_split<false, false, false, 0>(fin, chunks * n_parents, s, QM2_LEVEL_ARGS);
#undef QM2_LEVEL_ARGS
        HIP_OK(hipGetLastError());
        if (time_end(c, &t)) return 1;
    }
    /* finest buckets + table slices -> counters (skipped on device if the batch lost k-mers) */
    {
        const uint32_t n_buckets = (uint32_t)p.buckets[L];
        const int grid = (int)std::min<uint64_t>(n_buckets, (uint64_t)c->n_cu * QM2_COUNT_GRID_MULT);
        if (time_begin(c, &t, QM2_K_BUCKET_COUNT)) return 1;
        c->interior_dirty = true;
        /* the group size (rows of 2 k-mers per prober thread) is picked so that a bucket's fill (mean
         * + 3 sigma of uniform hashing; k-mers per byte measured on earlier batches through an asynchronous
         * copy of the stats block) wastes the fewest rows */
        stats_peek_harvest(c);
        /* (before the first snapshot has landed: the caller's hint, against the part of the bytes that can be sequence) */
        const double kpb = c->peek_bytes ? c->kmers_per_byte : c->kappa_hint * (double)src.n_upper / (double)std::max<uint64_t>(src.n_bytes, 1);
        const uint64_t per_bucket = (uint64_t)((double)src.n_bytes * (src.sv ? kpb : 1.0 / 1.3)) / n_buckets;
        int g = 4;
        constexpr int lw = 4;
        {
            /* group size: the fewest rows of 2 k-mers per prober thread that take a bucket's fill; ties: the larger group.
             * (Loader waves: 4 of the 16.  Until round 4 a launch chose between 3 and 4 — round 2: 22.0 vs 24.4 ms at 40 M
             * reads per launch; with the table image the three-wave loader chain is the longer one at every fill: 40 M reads
             * 22.1 vs 21.4 ms, 20 M 18.5 vs 15.7, 60 M 25.3 vs 23.5 — profiles/r04_c3/batch_sizes.txt.) */
            const double fill = (double)per_bucket + 3.0 * std::sqrt((double)per_bucket + 1.0);
            const int prober_threads = kCountThreads - lw * kWave;
            const int rows = std::max(1, (int)std::ceil(fill / (2.0 * prober_threads)));
            int best_rows = 1 << 30;
            for (int cand = 2; cand <= 6; cand++) {
                const int run = (rows + cand - 1) / cand * cand;           /* rows executed */
                if (run <= best_rows) { best_rows = run; g = cand; }
            }
        }
        {   /* (experiments: force the group size) */
            static const int env_g = getenv("QM2_COUNT_G") ? atoi(getenv("QM2_COUNT_G")) : 0;
            if (env_g >= 2 && env_g <= 6) g = env_g;
        }
        /* How long the loader waves hold back behind every barrier, in steps of 256 cycles (bucket_count_kernel): nothing
         * while the loaders' chain (flush -> stage -> fetch) is the longer leg of a period — up to ~3000 k-mers per bucket —
         * then growing with the probers' share.  C3, same box, 0 / 4 / 8 / 12 / 16 steps: 60 M reads per launch (8640 k-mers
         * per bucket) 23.47 / 23.21 / 23.04 / 22.80 / 23.64 ms; 40 M (5760) 21.02 / 20.49 / 20.55 / 21.06 / 23.34; 20 M
         * (2880) 15.74 / 16.20 / 17.78 / 19.32 / 21.24 (profiles/r04_c3/bucket_count_head_sleep.txt). */
        uint32_t head_sleep = per_bucket > 3000 ? (uint32_t)std::min<uint64_t>((per_bucket - 3000) / 512, 12) : 0u;
        {   /* (experiments: force it) */
            static const int env_hs = getenv("QM2_COUNT_HEAD_SLEEP") ? atoi(getenv("QM2_COUNT_HEAD_SLEEP")) : -1;
            if (env_hs >= 0) head_sleep = (uint32_t)std::min(env_hs, 64);
        }
        const CountGeo cg = {c->own_lo, c->span_lo, c->span_hi, c->H >> 1};
#define QM2_LAUNCH_COUNT(G_, LW_, WT_)                                                                          \
        hipLaunchKernelGGL((bucket_count_kernel<kLgS, G_, LW_, WT_>), dim3(grid), dim3(kCountThreads), 0, s,   \
                           (const uint32_t *)fin_r, (const WT_ *)fin_w_at, p.cursors + p.cursor_off[L], p.cap[L], table_view(c), cg, n_buckets, \
                           p.shared, c->stats, head_sleep)
#define QM2_LAUNCH_COUNT_LW(G_) do {                                                                            \
            if (fin_w == 2) QM2_LAUNCH_COUNT(G_, lw, uint16_t); else QM2_LAUNCH_COUNT(G_, lw, uint32_t);        \
        } while (0)
        switch (g) {
            case 2: QM2_LAUNCH_COUNT_LW(2); break;
            case 3: QM2_LAUNCH_COUNT_LW(3); break;
            case 4: QM2_LAUNCH_COUNT_LW(4); break;
            case 5: QM2_LAUNCH_COUNT_LW(5); break;
            default: QM2_LAUNCH_COUNT_LW(6); break;
        }
#undef QM2_LAUNCH_COUNT_LW
#undef QM2_LAUNCH_COUNT
        HIP_OK(hipGetLastError());
        if (time_end(c, &t)) return 1;
    }
    /* spill list (skew): the reference's batch loop on the direct path */
    {
        const uint64_t groups = (p.spill_cap + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
        const int grid = grid_for(c, std::min<uint64_t>(groups, 1u << 16));
        if (time_begin(c, &t, QM2_K_PROBE)) return 1;
        hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(kBlock), 0, s, p.spill, p.spill_cap, p.spill_count,
                           (const uint32_t *)nullptr, table_view(c), c->stats, &c->stats->part_lost,
                           (const unsigned long long *)nullptr);
        HIP_OK(hipGetLastError());
        if (time_end(c, &t)) return 1;
    }
    /* exact fallback, decided on the device: only if even the spill list overflowed */
    if (src.sv) {
        if (launch_direct(c, *src.sv, &c->stats->part_lost, false, true)) return 1;
    } else {
        const uint64_t groups = (src.slab_cap + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
        if (src.n_subs) {
            /* (hundreds of regions: ONE launch that goes over them, instead of one each) */
            if (time_begin(c, &t, QM2_K_PROBE)) return 1;
            hipLaunchKernelGGL(probe_regions_kernel, dim3(grid_for(c, std::min<uint64_t>(groups * src.n_subs * src.n_slabs, 1u << 16))), dim3(kBlock), 0, s,
                               src.slabs, src.slab_cap, src.n_slabs, src.n_subs, src.peer_stride ? src.peer_stride : src.n_subs, src.fill,
                               table_view(c), c->stats, &c->stats->part_lost);
            HIP_OK(hipGetLastError());
            if (time_end(c, &t, true)) return 1;
        } else
        for (uint32_t g = 0; g < src.n_slabs; g++) {
            if (time_begin(c, &t, QM2_K_PROBE)) return 1;
            hipLaunchKernelGGL(probe_kernel, dim3(grid_for(c, groups)), dim3(kBlock), 0, s, src.slabs + g * src.slab_cap,
                               src.slab_cap, (const unsigned long long *)nullptr, src.fill + g, table_view(c), c->stats,
                               (const unsigned long long *)nullptr, &c->stats->part_lost);
            HIP_OK(hipGetLastError());
            if (time_end(c, &t, g + 1 == src.n_slabs)) return 1;
        }
    }
    hipLaunchKernelGGL(part_end_kernel, dim3(1), dim3(64), 0, s, c->stats, p.spill_count);
    HIP_OK(hipGetLastError());
    return 0;
}

/* Count the sequence lines at `data` (device).  `carry` bytes at the start are
 * look-back only.  Chooses the pipeline: the partitioned one streams the whole
 * table once per batch (~16 B per slot), the direct one pays ~60 ps per k-mer in
 * sector gathers and memory-side atomics; break-even is near n = H / 13. */
/* Looks at the runs of occupied slots once per context (see long_run_kernel); *ok = the
 * partitioned pipeline may be used with this key table. */
static int runs_fit(qm2_ctx *c, bool *ok)
{
    PartPlan &p = c->part;
    std::lock_guard<std::mutex> guard(c->runs_mu);
    if (p.runs == 0) {
        const uint64_t n_buckets = p.buckets[p.levels - 1];
        const size_t words = (size_t)((n_buckets + 31) / 32);
        unsigned int *d_found = nullptr, found = 0;
        uint32_t *d_shared = nullptr;
        hipError_t e = hipMalloc((void **)&d_found, sizeof(unsigned int));
        c->n_allocations++;
        if (e == hipSuccess) e = hipMalloc((void **)&d_shared, words * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_found, 0, sizeof(unsigned int), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_shared, 0, words * sizeof(uint32_t), c->stream);
        if (e == hipSuccess) {
            const uint64_t n_slots = c->span_hi - c->span_lo;
            const int grid = (int)std::min<uint64_t>((n_slots + 255) / 256, (uint64_t)c->n_cu * 64);
            hipLaunchKernelGGL(long_run_kernel, dim3(grid), dim3(256), 0, c->stream, c->keys, n_slots, c->span_lo, c->own_lo,
                               n_buckets, (uint32_t)kHalo, (uint32_t)kLgS, d_shared, d_found);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&found, d_found, sizeof found, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_found) (void)hipFree(d_found);
        if (e != hipSuccess) {
            if (d_shared) (void)hipFree(d_shared);
            return fail("qm2: looking for long runs of occupied slots failed: %s", hipGetErrorString(e));
        }
        p.runs = (found & 2u) ? 3 : (found & 1u) ? 2 : 1;
        if (p.runs == 2) p.shared = d_shared;
        else (void)hipFree(d_shared);
    }
    *ok = p.runs != 3;
    return 0;
}

/* Fill the table image (qm2_ctx.h) once the keys are there and part_reserve has made room for it: stream-ordered in front
 * of every bucket_count launch, a 56-GB pass at hg38 scale (11 ms), once per context. */
static int table_image(qm2_ctx *c)
{
    std::lock_guard<std::mutex> guard(c->runs_mu);
    if (c->img_ready || !c->img_r || c->keys_pending) return 0;
    const uint64_t n_slots = c->span_hi - c->span_lo;
    const int grid = (int)std::min<uint64_t>((n_slots + 255) / 256, (uint64_t)c->n_cu * 32);
    if (c->rec_bytes == 6)
        hipLaunchKernelGGL(table_image_kernel<uint16_t>, dim3(grid), dim3(256), 0, c->stream, (const uint64_t *)c->keys, n_slots, c->span_lo,
                           (uint32_t)(c->H - 1), (uint32_t)kLgS, c->img_r, (uint16_t *)c->img_w);
    else
        hipLaunchKernelGGL(table_image_kernel<uint32_t>, dim3(grid), dim3(256), 0, c->stream, (const uint64_t *)c->keys, n_slots, c->span_lo,
                           (uint32_t)(c->H - 1), (uint32_t)kLgS, c->img_r, (uint32_t *)c->img_w);
    HIP_OK(hipGetLastError());
    c->img_ready = true;
    return 0;
}

/* Snapshot of the statistics behind the batch just launched, copied to pinned memory by the
 * stream itself: qm2_stats_peek and the group-size choice of the next batches read it once it
 * has landed, nobody waits for it. */
static int stats_peek_async(qm2_ctx *c)
{
    if (c->peek_pending || !c->stats_peek) return 0;
    HIP_OK(hipMemcpyAsync(c->stats_peek, c->stats, sizeof(DevStats), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipEventRecord(c->peek_evt, c->stream));
    c->peek_pending = true;
    c->peek_bytes_pending = c->bytes_launched;
    return 0;
}

static void stats_peek_harvest(qm2_ctx *c)
{
    if (c->peek_pending && hipEventQuery(c->peek_evt) == hipSuccess) {
        c->peek_pending = false;
        c->peek = *c->stats_peek;
        c->peek_bytes = c->peek_bytes_pending;
        if (c->peek_bytes) c->kmers_per_byte = std::min(1.0, (double)c->peek.n_kmers / (double)c->peek_bytes);
    }
}

/* seq_bound: an upper bound on the batch's sequence bytes when it is smaller than n (raw FASTQ text: half), else 0.
 * packed: `data` is n POSITIONS of packed sequence (whole blocks), packed_nl of them line ends or filler. */
static int launch_batch(qm2_ctx *c, const uint8_t *data, size_t n, size_t carry, const uint64_t *d_sup,
                        uint64_t n_sup, uint64_t seq_bound = 0, bool packed = false, uint64_t packed_nl = 0)
{
    if (n == 0) return 0;
    if (c->keys_consumed)
        return fail("qm2: the key table was overwritten by qm2_chain_gather; this context cannot count any more");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    const SeqView sv = packed ? seq_view_packed(data, n, d_sup, n_sup) : seq_view(data, n, carry, d_sup, n_sup);
    if (packed) {                                                    /* (in front of the count: stream-ordered either way) */
        hipLaunchKernelGGL(stats_sub_bases_kernel, dim3(1), dim3(64), 0, c->stream, c->stats, (unsigned long long)packed_nl);
        HIP_OK(hipGetLastError());
    }
    c->bytes_launched += n;
    if (part_plan_kind(c, 0)) return 1;                               /* (after qm2_probe_subslabs_device: the plan of our own level 0 again) */
    bool part = (c->flags & QM2_FLAG_PARTITION) != 0 || ((uint64_t)n >= c->H / 8 && n >= (1u << 20));
    if (c->flags & QM2_FLAG_DIRECT) part = false;
    if (c->span_lo != 0 || c->span_hi != c->H) part = false;         /* (text into a shard: only what stays in its span) */
    if (part && c->part.levels == 0) part = false;
    if (part) {                                                       /* occupied runs vs the halo */
        bool ok = false;
        if (runs_fit(c, &ok)) return 1;
        if (!ok) part = false;
    }
    const uint64_t bound = seq_bound && seq_bound < n ? seq_bound : (uint64_t)n;
    if (part && part_reserve(c, bound)) { part = false; c->prof.refused_batches++; }   /* no memory: direct (qm2_profile says so) */
    if (part && table_image(c)) return 1;
    if (part) {
        c->prof.partitioned_batches++;
        PartSource src;
        src.sv = &sv;
        src.n_upper = bound + 16;
        src.n_bytes = sv.n_total;
        if (launch_partitioned(c, src)) return 1;
    } else {
        c->prof.direct_batches++;
        if (launch_direct(c, sv, nullptr, true, true)) return 1;
    }
    return stats_peek_async(c);
}

/* ---- open / close ---------------------------------------------------------- */

extern "C" int qm2_open(qm2_ctx **out, const uint64_t *keys, uint64_t hash_size, uint32_t k,
                        const qm2_opts *opts)
{
    if (!out) return fail("qm2_open: null out pointer");
    *out = nullptr;
    qm2_opts o;
    memset(&o, 0, sizeof o);
    o.device = -1;
    if (opts) {
        size_t sz = opts->struct_size ? opts->struct_size : sizeof o;
        memcpy(&o, opts, std::min(sz, sizeof o));
    }
    if (hash_size < 2 || (hash_size & (hash_size - 1)) || hash_size > (1ull << 32))
        return fail("qm2_open: hash size 0x%llX is not a power of two in [2, 2^32]",
                    (unsigned long long)hash_size);
    if (k < 1 || k > 31)
        return fail("qm2_open: k-mer size %u unsupported (1..31; k=32 is undefined in the reference)", k);
    if (!keys && !o.ext_keys && !(o.flags & QM2_FLAG_KEYS_LATER)) return fail("qm2_open: no keys given");
    if (((uintptr_t)o.ext_keys | (uintptr_t)o.ext_counters) & 15u)
        return fail("qm2_open: ext_keys / ext_counters must be 16-byte aligned");
    if ((o.flags & QM2_FLAG_DIRECT) && (o.flags & QM2_FLAG_PARTITION))
        return fail("qm2_open: QM2_FLAG_DIRECT and QM2_FLAG_PARTITION exclude each other");

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail("qm2_open: no usable HIP device (%s); there is no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));

    qm2_ctx *c = new qm2_ctx();
    memset(&c->prof, 0, sizeof c->prof);
    if (o.device >= 0) c->device = o.device;
    else if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    if (c->device >= n_dev) {
        delete c;
        return fail("qm2_open: device %d out of range (%d present)", o.device, n_dev);
    }
    c->k = k;
    c->H = hash_size;
    while ((1ull << c->lgH) < hash_size) c->lgH++;
    c->flags = o.flags;
    c->span_lo = o.span_lo;
    c->span_hi = o.span_hi ? o.span_hi : hash_size;
    if (c->span_lo >= c->span_hi || c->span_hi > hash_size) {
        delete c;
        return fail("qm2_open: bad span [%llu, %llu)", (unsigned long long)o.span_lo,
                    (unsigned long long)o.span_hi);
    }
    const bool whole = c->span_lo == 0 && c->span_hi == hash_size;
    c->own_lo = whole ? 0 : o.own_lo;
    c->own_hi = whole ? hash_size : o.own_hi;
    if (!whole && c->own_hi != 0) {
        const uint64_t r = c->own_hi - c->own_lo;
        if (c->own_lo >= c->own_hi || (r & (r - 1)) || (c->own_lo & (r - 1)) || c->own_lo < c->span_lo || c->own_hi > c->span_hi ||
            ((c->span_lo | c->span_hi) & 63u)) {
            delete c;
            return fail("qm2_open: bad own range [%llu, %llu) for span [%llu, %llu) (a power of two of slots, aligned, inside a span of whole 64-slot blocks)",
                        (unsigned long long)o.own_lo, (unsigned long long)o.own_hi, (unsigned long long)o.span_lo, (unsigned long long)o.span_hi);
        }
    }
    const uint64_t n_slots = c->span_hi - c->span_lo;

#define OPEN_OK(call)                                                             \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            fail("Memory allocation failed (%s: %s)", #call, hipGetErrorString(e_)); \
            qm2_close(c);                                                         \
            return 1;                                                             \
        }                                                                         \
    } while (0)

    OPEN_OK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    OPEN_OK(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    if (o.stream) c->stream = (hipStream_t)o.stream;
    else {
        OPEN_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    OPEN_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (o.ext_keys) c->keys = (uint64_t *)o.ext_keys;
    else {
        OPEN_OK(hipMalloc((void **)&c->keys, n_slots * sizeof(uint64_t)));
        c->own_keys = true;
        if (keys) OPEN_OK(hipMemcpy(c->keys, keys, n_slots * sizeof(uint64_t), hipMemcpyHostToDevice));
        else c->keys_pending = true;                          /* QM2_FLAG_KEYS_LATER */
    }
    if (o.ext_counters) c->counters = (uint32_t *)o.ext_counters;
    else {
        OPEN_OK(hipMalloc((void **)&c->counters, n_slots * sizeof(uint32_t)));
        c->own_counters = true;
        /* stream-ordered: a plain hipMemset runs on the null stream, which the context's
         * non-blocking stream does not wait for — kernels could start before it finishes */
        OPEN_OK(hipMemsetAsync(c->counters, 0, n_slots * sizeof(uint32_t), c->stream));
    }
    OPEN_OK(hipMalloc((void **)&c->stats, sizeof(DevStats)));
    OPEN_OK(hipMemsetAsync(c->stats, 0, sizeof(DevStats), c->stream));
    OPEN_OK(hipStreamSynchronize(c->stream));
    OPEN_OK(hipHostMalloc((void **)&c->stats_peek, sizeof(DevStats), hipHostMallocDefault));
    OPEN_OK(hipEventCreateWithFlags(&c->peek_evt, hipEventDisableTiming));

    c->staging_bytes = o.staging_bytes ? (size_t)o.staging_bytes : ((size_t)64 << 20);
    if (c->staging_bytes < 64) c->staging_bytes = 64;
    /* default device batch: one byte per table slot, 1..6 GiB — the partitioned pipeline streams the
     * whole table once per batch, so batches should not be small against it (hg38 scale: 64 G bases/s
     * at 3 G bases per batch, 70 at 4.5 and 6) */
    c->batch_bytes = o.batch_bytes ? (size_t)o.batch_bytes
                                   : (size_t)std::min<uint64_t>(std::max<uint64_t>(hash_size + hash_size / 2, 1ull << 30), 6ull << 30);
    if (c->batch_bytes < 64) c->batch_bytes = 64;
    c->slots.resize(o.n_staging ? std::max<uint32_t>(o.n_staging, 2) : 4);   /* allocated lazily */
    if (o.kmers_per_kib) c->kappa_hint = std::min(1.0, std::max(0.05, (double)o.kmers_per_kib / 1024.0));
    part_init(c, o.part_max_bits >= 1 && o.part_max_bits <= 10 ? (int)o.part_max_bits : 10);
#undef OPEN_OK
    *out = c;
    return 0;
}

extern "C" void qm2_close(qm2_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    /* only this context's own work: a device-wide synchronize would stall the other contexts of a
     * multi-context process (replicas, shards) */
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->slots) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto &b : c->batch) {
        if (b.dev) (void)hipFree(b.dev);
        if (b.dev_wraps) (void)hipFree(b.dev_wraps);
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.copied) (void)hipEventDestroy(b.copied);
    }
    for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto &t : c->pool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    qm2_part_free(c->part);
    if (c->part.shared) { (void)hipFree(c->part.shared); c->part.shared = nullptr; }
    if (c->own_chain && c->chain) (void)hipFree(c->chain);
    if (c->gathered) (void)hipFree(c->gathered);
    if (c->own_keys && c->keys) (void)hipFree(c->keys);
    if (c->own_counters && c->counters) (void)hipFree(c->counters);
    if (c->interior16) (void)hipFree(c->interior16);
    if (c->img_r) (void)hipFree(c->img_r);
    if (c->img_w) (void)hipFree(c->img_w);
    if (c->narrow) (void)hipFree(c->narrow);
    if (c->stats) (void)hipFree(c->stats);
    if (c->stats_peek) (void)hipHostFree(c->stats_peek);
    if (c->raw_out) (void)hipFree(c->raw_out);
    if (c->raw_tiles) (void)hipFree(c->raw_tiles);
    if (c->raw_flags) (void)hipFree(c->raw_flags);
    if (c->raw_wraps) (void)hipFree(c->raw_wraps);
    if (c->peek_evt) (void)hipEventDestroy(c->peek_evt);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void *qm2_counters_device(qm2_ctx *c)
{
    /* (whoever asks for the array is about to read it: the interiors' 2-byte counts go in first, behind the work queued so far) */
    if (c && c->interior_dirty && (qm2_fold_counters(c) || hipStreamSynchronize(c->stream) != hipSuccess)) return nullptr;
    return c ? c->counters : nullptr;
}
extern "C" void *qm2_keys_device(qm2_ctx *c) { return c ? c->keys : nullptr; }

/* ---- host-buffer pipeline: pinned staging -> device batch -> kernels --------- */

/* bytes a device batch needs for text (ASCII or raw), and for packed blocks of batch_bytes positions */
static size_t batch_text_bytes(const qm2_ctx *c) { return c->batch_bytes + kCarry + kDevPad; }
static size_t batch_packed_bytes(const qm2_ctx *c) { return (c->batch_bytes / 64 + 1) * QM2_PACKED_BLOCK_BYTES + kDevPad; }

static int batch_ready(qm2_ctx *c, Batch &b, size_t need)
{
    /* (events first: b.dev marks the batch as complete, so a failure in between is retried whole) */
    if (!b.done) HIP_OK(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
    if (!b.copied) HIP_OK(hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
    if (b.busy) {
        HIP_OK(hipEventSynchronize(b.done));
        b.busy = false;
    }
    if (b.dev && b.cap < need) {              /* prepared for packed input only (QM2_PREPARE_PACKED), and text arrives: grow */
        HIP_OK(hipFree(b.dev));
        b.dev = nullptr;
        b.cap = 0;
    }
    if (!b.dev) {
        HIP_OK(hipMalloc((void **)&b.dev, need));
        b.cap = need;
        c->n_allocations++;
    }
    return 0;
}

static int raw_classify_and_count(qm2_ctx *c, const void *d_raw, size_t n, uint32_t fastq, uint32_t skip,
                                  uint32_t *verdict, bool continues);
static int raw_reserve(qm2_ctx *c, size_t n);

/* Launch the kernels for the batch being filled and switch to the other one. */
static int batch_flush(qm2_ctx *c)
{
    Batch &b = c->batch[c->cur];
    if (b.fill <= b.carry) { b.fill = b.carry = 0; b.wraps.clear(); return 0; }
    /* the kernels must see every staged copy of this batch */
    HIP_OK(hipEventRecord(b.copied, c->copy_stream));
    HIP_OK(hipStreamWaitEvent(c->stream, b.copied, 0));
    if (b.raw) {                              /* raw text: line discipline on the device, then the count */
        uint32_t verdict = 0;
        if (raw_classify_and_count(c, b.dev, b.fill, b.raw_fastq, b.raw_skip, &verdict, true)) return 1;
        if (verdict) c->raw_verdict |= verdict;
        HIP_OK(hipEventRecord(b.done, c->stream));
        b.busy = true;
        b.fill = b.carry = 0;
        b.raw = false;
        c->cur ^= 1;
        return 0;
    }
    uint64_t *d_sup = nullptr;
    const uint64_t n_sup = b.wraps.size();
    const bool packed = b.packed;
    const uint64_t packed_nl = b.packed_nl;
    b.packed = false; b.packed_nl = 0;
    if (n_sup) {                              /* long lines are rare: a small blocking copy is fine */
        if (b.wraps_cap < n_sup) {
            if (b.dev_wraps) HIP_OK(hipFree(b.dev_wraps));
            b.wraps_cap = std::max<size_t>(n_sup, 1024);
            HIP_OK(hipMalloc((void **)&b.dev_wraps, b.wraps_cap * sizeof(uint64_t)));
            c->n_allocations++;
        }
        HIP_OK(hipMemcpy(b.dev_wraps, b.wraps.data(), n_sup * sizeof(uint64_t), hipMemcpyHostToDevice));
        d_sup = b.dev_wraps;
    }
    if (packed ? launch_batch(c, b.dev, b.fill / QM2_PACKED_BLOCK_BYTES * 64, 0, d_sup, n_sup, 0, true, packed_nl)
               : launch_batch(c, b.dev, b.fill, b.carry, d_sup, n_sup)) return 1;
    HIP_OK(hipEventRecord(b.done, c->stream));
    b.busy = true;
    b.fill = b.carry = 0;
    b.wraps.clear();
    c->cur ^= 1;
    return 0;
}

int qm2_slot_ready(qm2_ctx *c, Slot &s)
{
    if (!s.host) {
        HIP_OK(hipHostMalloc((void **)&s.host, c->staging_bytes, hipHostMallocDefault));
        c->n_allocations++;
        HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (s.busy) {
        HIP_OK(hipEventSynchronize(s.done));
        s.busy = false;
    }
    return 0;
}

/* Stage [src, src+n) behind the current batch's fill.  `owner` = the pinned slot src lies in
 * (qm2_staging_commit: copied from in place), or null: src is pageable and goes through the
 * free pinned slots. */
static int stage_bytes(qm2_ctx *c, Batch &b, const char *src, size_t n, Slot *owner)
{
    if (owner) {
        HIP_OK(hipMemcpyAsync(b.dev + b.fill, src, n, hipMemcpyHostToDevice, c->copy_stream));
        HIP_OK(hipEventRecord(owner->done, c->copy_stream));
        owner->busy = true;
        b.fill += n;
        c->prof.staged_bytes += n;
        return 0;
    }
    size_t done = 0;
    while (done < n) {
        Slot *s = nullptr;
        for (size_t tries = 0; tries < c->slots.size() && !s; tries++) {
            Slot &cand = c->slots[c->next_slot];
            c->next_slot = (c->next_slot + 1) % (uint32_t)c->slots.size();
            if (!cand.lent) s = &cand;
        }
        if (!s) return fail("qm2_submit: every staging buffer is held by qm2_staging_acquire");
        if (qm2_slot_ready(c, *s)) return 1;
        const size_t take = std::min(c->staging_bytes, n - done);
        memcpy(s->host, src + done, take);
        HIP_OK(hipMemcpyAsync(b.dev + b.fill, s->host, take, hipMemcpyHostToDevice, c->copy_stream));
        HIP_OK(hipEventRecord(s->done, c->copy_stream));
        s->busy = true;
        b.fill += take;
        c->prof.staged_bytes += take;
        done += take;
    }
    return 0;
}

static int submit_lines(qm2_ctx *c, const char *buf, size_t n, Slot *owner);
static int submit_packing(qm2_ctx *c, const char *buf, size_t n);

extern "C" int qm2_submit(qm2_ctx *c, const char *buf, size_t n)
{
    if (!c) return fail("qm2_submit: null context");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (n == 0) return 0;
    if (!buf) return fail("qm2_submit: null buffer");
    if (buf[n - 1] != '\n')
        return fail("qm2_submit: buffer must end with a newline (whole sequence lines only)");
    if (use_device(c)) return 1;
    if (!(c->flags & QM2_FLAG_SUBMIT_ASCII)) return submit_packing(c, buf, n);
    return submit_lines(c, buf, n, nullptr);
}

/* One packed piece, already in pinned slot `s` (blocks from its first byte), behind the current batch's fill. */
static bool early_flush_enabled()
{
    static const bool on = [] { const char *e = getenv("QM2_EARLY_FLUSH"); return !(e && e[0] == '0'); }();
    return on;
}

static int stage_packed(qm2_ctx *c, Slot &s, const qm2_packed &p, const uint64_t *wraps)
{
    const uint64_t positions = p.n_blocks * 64;
    if (positions > c->batch_bytes) return fail("qm2: a packed piece of %llu positions is larger than a device batch", (unsigned long long)positions);
    {
        Batch &b0 = c->batch[c->cur];
        if (b0.fill && (!b0.packed || b0.fill / QM2_PACKED_BLOCK_BYTES * 64 + positions > c->batch_bytes) && batch_flush(c)) return 1;
    }
    Batch &b = c->batch[c->cur];
    if (batch_ready(c, b, batch_packed_bytes(c))) return 1;
    if (b.fill == 0) { b.packed = true; b.packed_nl = 0; b.carry = 0; b.raw = false; b.wraps.clear(); }
    const uint64_t pos0 = b.fill / QM2_PACKED_BLOCK_BYTES * 64;
    HIP_OK(hipMemcpyAsync(b.dev + b.fill, s.host, p.n_blocks * QM2_PACKED_BLOCK_BYTES, hipMemcpyHostToDevice, c->copy_stream));
    HIP_OK(hipEventRecord(s.done, c->copy_stream));
    s.busy = true;
    b.fill += p.n_blocks * QM2_PACKED_BLOCK_BYTES;
    c->prof.staged_bytes += p.n_blocks * QM2_PACKED_BLOCK_BYTES;
    b.packed_nl += p.n_newlines;
    for (uint64_t i = 0; i < p.n_wraps; i++) b.wraps.push_back(wraps[i] + pos0);
    if (b.fill / QM2_PACKED_BLOCK_BYTES * 64 >= c->batch_bytes) return batch_flush(c);
    /* An idle device does not wait for a whole batch: with a quarter of one in hand and the other batch's kernels finished
     * (or never started: the first batch of a job), what is there is launched.  Fed slower than it counts, the device works
     * on quarter batches with the link's latency instead of a whole batch's; fed faster, the other batch is busy whenever
     * this one is looked at and batches fill up as before.  (Host-fed bench, 8 x 9 G bases: the first launch 65 ms -> 16 ms
     * after the first byte.)  QM2_EARLY_FLUSH=0 switches it off. */
    /* (never so early that the piece would be too small for the partitioned pipeline: launch_batch, n >= H / 8) */
    if (b.fill / QM2_PACKED_BLOCK_BYTES * 64 >= std::max<uint64_t>(c->batch_bytes / 4, c->H / 8) && early_flush_enabled()) {
        Batch &o = c->batch[c->cur ^ 1];
        bool idle = !o.busy;
        if (!idle) {
            const hipError_t q = hipEventQuery(o.done);
            if (q == hipSuccess) idle = true;
            else if (q != hipErrorNotReady) HIP_OK(q);
            else (void)hipGetLastError();
        }
        if (idle) return batch_flush(c);
    }
    return 0;
}

/* qm2_submit's default: the pass it makes over the caller's lines anyway (into a pinned buffer) packs them, so that
 * 0.375 bytes per base cross the link (qm2_pack.c).  Pieces are whole lines; a line that no piece holds (longer than a
 * staging buffer's worth of positions), or one with more run-counter wraps than the list takes, goes the ASCII way. */
static int submit_packing(qm2_ctx *c, const char *buf, size_t n)
{
    /* (whole blocks: what a piece's text rounds up to must still fit a batch) */
    const size_t piece_max = std::min<size_t>((c->staging_bytes / QM2_PACKED_BLOCK_BYTES) * 64, c->batch_bytes) & ~(size_t)63;
    if (piece_max == 0) return submit_lines(c, buf, n, nullptr);
    uint64_t wraps[256];
    size_t off = 0;
    while (off < n) {
        size_t take = std::min(piece_max, n - off);
        if (take < n - off) {
            const void *nl = take ? memrchr(buf + off, '\n', take) : nullptr;
            if (!nl) {                                         /* one line longer than a piece: as ASCII, up to its end */
                const char *e = (const char *)memchr(buf + off, '\n', n - off);
                const size_t len = (size_t)(e - (buf + off)) + 1;
                if (submit_lines(c, buf + off, len, nullptr)) return 1;
                off += len;
                continue;
            }
            take = (size_t)((const char *)nl - (buf + off)) + 1;
        }
        Slot *s = nullptr;
        for (size_t tries = 0; tries < c->slots.size() && !s; tries++) {
            Slot &cand = c->slots[c->next_slot];
            c->next_slot = (c->next_slot + 1) % (uint32_t)c->slots.size();
            if (!cand.lent) s = &cand;
        }
        if (!s) return fail("qm2_submit: every staging buffer is held by qm2_staging_acquire");
        if (qm2_slot_ready(c, *s)) return 1;
        qm2_packed p;
        const int rc = qm2_pack_lines(buf + off, take, s->host, c->staging_bytes, wraps, sizeof wraps / sizeof wraps[0], &p);
        if (rc == 1) return fail("qm2_submit: packing failed");
        if (rc == 2) { if (submit_lines(c, buf + off, take, nullptr)) return 1; }
        else if (stage_packed(c, *s, p, wraps)) return 1;
        off += take;
    }
    return 0;
}

static int submit_lines(qm2_ctx *c, const char *buf, size_t n, Slot *owner)
{
    if ((c->batch[c->cur].raw || c->batch[c->cur].packed) && batch_flush(c)) return 1;   /* a raw / packed batch is pending: it goes first */
    std::vector<Part> parts;
    {
        const Batch &b0 = c->batch[c->cur];
        const size_t pending = b0.fill - b0.carry;
        plan_parts(buf, n, pending < c->batch_bytes ? c->batch_bytes - pending : 0, c->batch_bytes, parts);
    }
    for (size_t i = 0; i < parts.size(); i++) {
        const Part &pt = parts[i];
        if (pt.take == 0) {                   /* no line end fits behind the pending bytes */
            if (batch_flush(c)) return 1;
            continue;
        }
        Batch &b = c->batch[c->cur];
        if (batch_ready(c, b, batch_text_bytes(c))) return 1;
        const size_t at = b.fill;             /* first staged byte of this part lands at dev[at] */
        if (pt.carry) b.carry = pt.carry;     /* (only ever the first bytes of an empty batch) */
        if (stage_bytes(c, b, buf + pt.off - pt.carry, pt.carry + pt.take, owner)) return 1;
        for (uint64_t w : pt.wraps) b.wraps.push_back(w + at);
        /* every part but the last fills its batch; the last one may */
        if (i + 1 < parts.size() || b.fill - b.carry >= c->batch_bytes)
            if (batch_flush(c)) return 1;
    }
    return 0;
}

extern "C" int qm2_staging_acquire(qm2_ctx *c, char **buf, size_t *capacity)
{
    if (!c || !buf || !capacity) return fail("qm2_staging_acquire: null argument");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (use_device(c)) return 1;
    size_t lent = 0;
    for (auto &s : c->slots) lent += s.lent ? 1 : 0;
    if (lent + 1 >= c->slots.size()) return fail("qm2_staging_acquire: too many staging buffers held");
    for (size_t tries = 0; tries < c->slots.size(); tries++) {
        Slot &s = c->slots[c->next_slot];
        c->next_slot = (c->next_slot + 1) % (uint32_t)c->slots.size();
        if (s.lent) continue;
        if (qm2_slot_ready(c, s)) return 1;
        s.lent = true;
        *buf = s.host;
        *capacity = c->staging_bytes;
        return 0;
    }
    return fail("qm2_staging_acquire: no staging buffer free");
}

extern "C" int qm2_staging_commit(qm2_ctx *c, char *buf, size_t n)
{
    if (!c || !buf) return fail("qm2_staging_commit: null argument");
    Slot *owner = nullptr;
    for (auto &s : c->slots) if (s.lent && s.host == buf) owner = &s;
    if (!owner) return fail("qm2_staging_commit: not a buffer from qm2_staging_acquire");
    owner->lent = false;
    if (n == 0) return 0;
    if (n > c->staging_bytes) return fail("qm2_staging_commit: more bytes than the buffer holds");
    if (buf[n - 1] != '\n')
        return fail("qm2_staging_commit: buffer must end with a newline (whole sequence lines only)");
    if (use_device(c)) return 1;
    return submit_lines(c, buf, n, owner);
}

/* qm2_submit with the packing spread over n_threads host threads: the caller's buffer is cut into pieces of whole lines,
 * the threads pack them into pinned staging buffers (qm2_pack_lines), the calling thread hands the buffers out and
 * commits the pieces in order — so that one call feeds the link at what several cores pack (a single thread packs
 * ~30 GB/s of text on a Zen 5 core, the link takes the packed form of ~137 G bases/s). */
extern "C" int qm2_submit_threads(qm2_ctx *c, const char *buf, size_t n, uint32_t n_threads)
{
    if (!c) return fail("qm2_submit_threads: null context");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (n == 0) return 0;
    if (!buf) return fail("qm2_submit_threads: null buffer");
    if (buf[n - 1] != '\n') return fail("qm2_submit_threads: buffer must end with a newline (whole sequence lines only)");
    if (use_device(c)) return 1;
    size_t free_slots = 0;
    for (auto &s : c->slots) free_slots += s.lent ? 0 : 1;
    if (n_threads == 0 || (c->flags & QM2_FLAG_SUBMIT_ASCII) || free_slots < 2) return (c->flags & QM2_FLAG_SUBMIT_ASCII) ? submit_lines(c, buf, n, nullptr) : submit_packing(c, buf, n);
    const size_t piece_cap = std::min<size_t>((c->staging_bytes / QM2_PACKED_BLOCK_BYTES) * 64, c->batch_bytes) & ~(size_t)63;
    /* pieces of ~32 MiB of text (what a reader thread of the CLI handles at a time), whole lines */
    const size_t piece_max = std::min<size_t>(piece_cap, (size_t)32 << 20);
    if (piece_max < 4096) return submit_packing(c, buf, n);
    struct Job { size_t off = 0, len = 0; Slot *slot = nullptr; qm2_packed p = {0, 0, 0}; uint64_t wraps[64]; int rc = 0; bool done = false; bool ascii = false; bool slot_released = false; };
    std::vector<Job> jobs;
    for (size_t off = 0; off < n;) {
        size_t take = std::min(piece_max, n - off);
        bool ascii = false;
        if (take < n - off) {
            const void *nl = memrchr(buf + off, '\n', take);
            if (nl) take = (size_t)((const char *)nl - (buf + off)) + 1;
            else { take = (size_t)((const char *)memchr(buf + off, '\n', n - off) - (buf + off)) + 1; ascii = true; }   /* a line longer than a piece */
        }
        Job j;
        j.off = off; j.len = take; j.ascii = ascii;
        jobs.push_back(j);
        off += take;
    }
    const size_t window = std::min<size_t>(std::min<size_t>((size_t)n_threads + 2, free_slots - 1), jobs.size());
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    size_t next_job = 0, issued = 0;              /* jobs [next_job, issued) wait for a thread */
    bool quit = false;
    auto worker = [&]() {
        for (;;) {
            Job *j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || next_job < issued; });
                if (next_job >= issued) return;
                j = &jobs[next_job++];
            }
            if (!j->ascii) j->rc = qm2_pack_lines(buf + j->off, j->len, j->slot->host, c->staging_bytes, j->wraps, 64, &j->p);
            {
                std::lock_guard<std::mutex> lk(mu);
                j->done = true;
            }
            cv_done.notify_all();
        }
    };
    std::vector<std::thread> pool;
    const size_t n_workers = std::min<size_t>(n_threads, jobs.size());
    for (size_t t = 0; t < n_workers; t++) pool.emplace_back(worker);
    int rc = 0;
    size_t committed = 0;
    auto issue_one = [&]() -> int {
        Job &j = jobs[issued];
        if (!j.ascii) {
            Slot *s = nullptr;
            for (size_t tries = 0; tries < c->slots.size() && !s; tries++) {
                Slot &cand = c->slots[c->next_slot];
                c->next_slot = (c->next_slot + 1) % (uint32_t)c->slots.size();
                if (!cand.lent) s = &cand;
            }
            if (!s) return fail("qm2_submit_threads: every staging buffer is held by qm2_staging_acquire");
            if (qm2_slot_ready(c, *s)) return 1;
            s->lent = true;                       /* (ours until its piece is committed) */
            j.slot = s;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            issued++;
        }
        cv_work.notify_one();
        return 0;
    };
    /* pieces are committed as they get ready, in ANY order (counting does not care): one slow thread — 14 of them on a
     * 16-CPU share get descheduled now and then — then holds up its own piece only, not the window behind it */
    std::vector<size_t> open;                     /* issued, not yet committed */
    while (!rc && committed < jobs.size()) {
        while (!rc && issued < jobs.size() && open.size() < window) { open.push_back(issued); rc = issue_one(); if (rc) open.pop_back(); }
        if (rc) break;
        size_t at = 0;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { for (size_t i = 0; i < open.size(); i++) if (jobs[open[i]].done) { at = i; return true; } return false; });
        }
        Job &j = jobs[open[at]];
        open.erase(open.begin() + (long)at);
        if (j.slot) j.slot->lent = false;
        j.slot_released = true;
        if (j.ascii || j.rc == 2) rc = submit_lines(c, buf + j.off, j.len, nullptr);          /* does not pack: the ASCII way */
        else if (j.rc) rc = fail("qm2_submit_threads: packing failed");
        else rc = stage_packed(c, *j.slot, j.p, j.wraps);
        committed++;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        quit = true;
        /* (jobs issued but not committed after an error: their threads finish them, nobody looks) */
    }
    cv_work.notify_all();
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < issued; i++) if (jobs[i].slot && !jobs[i].slot_released) jobs[i].slot->lent = false;
    return rc;
}

extern "C" int qm2_staging_commit_packed(qm2_ctx *c, char *buf, const qm2_packed *p, const uint64_t *wraps)
{
    if (!c || !buf || !p) return fail("qm2_staging_commit_packed: null argument");
    Slot *owner = nullptr;
    for (auto &s : c->slots) if (s.lent && s.host == buf) owner = &s;
    if (!owner) return fail("qm2_staging_commit_packed: not a buffer from qm2_staging_acquire");
    owner->lent = false;
    if (p->n_blocks == 0) return 0;
    if (p->n_blocks > c->staging_bytes / QM2_PACKED_BLOCK_BYTES) return fail("qm2_staging_commit_packed: more blocks than the buffer holds");
    if (p->n_newlines > p->n_blocks * 64) return fail("qm2_staging_commit_packed: more line ends than positions");
    if (p->n_wraps && !wraps) return fail("qm2_staging_commit_packed: null wrap list");
    if (use_device(c)) return 1;
    return stage_packed(c, *owner, *p, wraps);
}

extern "C" int qm2_submit_packed_device(qm2_ctx *c, const void *d_blocks, const qm2_packed *p, const uint64_t *d_wraps)
{
    if (!c || !p) return fail("qm2_submit_packed_device: null argument");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (p->n_blocks == 0) return 0;
    if (!d_blocks) return fail("qm2_submit_packed_device: null buffer");
    if ((uintptr_t)d_blocks & 15u) return fail("qm2_submit_packed_device: the blocks must be 16-byte aligned");
    if (p->n_newlines > p->n_blocks * 64) return fail("qm2_submit_packed_device: more line ends than positions");
    if (p->n_wraps && !d_wraps) return fail("qm2_submit_packed_device: null wrap list");
    if (use_device(c)) return 1;
    return launch_batch(c, (const uint8_t *)d_blocks, p->n_blocks * 64, 0, d_wraps, p->n_wraps, 0, true, p->n_newlines);
}

extern "C" int qm2_submit_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup,
                                 uint64_t n_sup)
{
    if (!c) return fail("qm2_submit_device: null context");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (n == 0) return 0;
    if (!d_seq) return fail("qm2_submit_device: null buffer");
    if (use_device(c)) return 1;
    return launch_batch(c, (const uint8_t *)d_seq, n, 0, d_sup, n_sup);
}

/* classify the n raw bytes at d_raw (device, 16-byte aligned) and, unless refused, count them.
 * continues: the block follows the previous raw block of this context in the same text (staged input) */
/* scratch of the device line discipline for blocks of up to n raw bytes */
static int raw_reserve(qm2_ctx *c, size_t n)
{
    hipStream_t s = c->stream;
    const uint64_t n_tiles = (n + kRawTile - 1) / kRawTile;
    if (n + 32 > c->raw_cap) {
        HIP_OK(hipStreamSynchronize(s));
        if (c->raw_out) HIP_OK(hipFree(c->raw_out));
        c->raw_out = nullptr;
        c->raw_cap = n + n / 8 + 4096;
        HIP_OK(hipMalloc((void **)&c->raw_out, c->raw_cap));
        c->n_allocations++;
    }
    if (n_tiles > c->raw_tile_cap) {
        HIP_OK(hipStreamSynchronize(s));
        if (c->raw_tiles) HIP_OK(hipFree(c->raw_tiles));
        c->raw_tiles = nullptr;
        c->raw_tile_cap = n_tiles + n_tiles / 8 + 16;
        HIP_OK(hipMalloc(&c->raw_tiles, c->raw_tile_cap * 40));
        c->n_allocations++;
    }
    if (!c->raw_flags) {
        HIP_OK(hipMalloc((void **)&c->raw_flags, 4 * sizeof(unsigned int)));
        c->n_allocations++;
        HIP_OK(hipMalloc((void **)&c->raw_wraps, kRawWrapCap * sizeof(uint64_t)));
    }
    return 0;
}

static int raw_classify_and_count(qm2_ctx *c, const void *d_raw, size_t n, uint32_t fastq, uint32_t skip,
                                  uint32_t *verdict, bool continues)
{
    hipStream_t s = c->stream;
    const uint64_t n_tiles = (n + kRawTile - 1) / kRawTile;
    if (raw_reserve(c, n)) return 1;
    /* scratch: three channels (i64 last newline, i64 last reset, u32 newline count) per tile and per
     * group; the group arrays use the first n_groups entries of their cap-sized slices */
    const uint64_t n_groups = (n_tiles + kRawGroup - 1) / kRawGroup, cap = c->raw_tile_cap;
    long long *i64s = (long long *)c->raw_tiles;
    uint32_t *u32s = (uint32_t *)(i64s + 4 * cap);
    const RawArrays tile = {u32s, i64s, i64s + cap}, group = {u32s + cap, i64s + 2 * cap, i64s + 3 * cap};
    RawView rv;
    rv.in = (const uint8_t *)d_raw;
    rv.out = c->raw_out;
    rv.n = (uint64_t)n;
    rv.fastq = fastq ? 1u : 0u;
    /* '>' lines at sequence positions seen in earlier batches of this text moved every later sequence
     * line on by one (qm2_raw.hip.h): the caller's skip follows the line count only */
    rv.skip = fastq ? (skip + (continues ? c->raw_shift : 0u)) & 3u : 0u;
    rv.n_desync = 0;
    TimedLaunch t;
    if (time_begin(c, &t, QM2_K_LINES)) return 1;
    hipLaunchKernelGGL(raw_tile_kernel, dim3((uint32_t)n_tiles), dim3(kRawThreads), 0, s, rv, tile);
    hipLaunchKernelGGL(raw_group_kernel, dim3((uint32_t)n_groups), dim3(kRawThreads), 0, s, tile, n_tiles, group);
    hipLaunchKernelGGL(raw_scan_kernel, dim3(1), dim3(1024), 0, s, group, n_groups);
    /* raw_flags: [0] refusal bits, [1] number of wrap positions, [2..3] u64 first unlisted desync line */
    struct { unsigned int bits, n_wraps; unsigned long long next_desync; } fl;
    for (;;) {
        fl.bits = 0; fl.n_wraps = 0; fl.next_desync = ~0ull;
        HIP_OK(hipMemcpyAsync(c->raw_flags, &fl, sizeof fl, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(raw_apply_kernel, dim3((uint32_t)n_tiles), dim3(kRawThreads), 0, s, rv, tile, group, c->raw_flags,
                           c->raw_wraps, c->raw_flags + 1, (unsigned long long *)(c->raw_flags + 2));
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(&fl, c->raw_flags, sizeof fl, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (fl.bits || fl.next_desync == ~0ull) break;
        if (rv.n_desync == kRawMaxDesync || (c->flags & QM2_FLAG_RAW_STRICT)) { fl.bits |= kRawDesync; break; }
        rv.desync[rv.n_desync++] = fl.next_desync;           /* found in ascending order: the list stays sorted */
    }
    if (time_end(c, &t)) return 1;
    if (fl.bits) { *verdict = fl.bits; return 0; }           /* not counted: the caller takes the host path */
    if (continues) c->raw_shift = (c->raw_shift + rv.n_desync) & 3u;
    const unsigned int n_wraps = fl.n_wraps;
    if (n_wraps > 1) {                                       /* the kernels want the wrap list ascending; it is short */
        std::vector<uint64_t> w(n_wraps);
        HIP_OK(hipMemcpy(w.data(), c->raw_wraps, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::sort(w.begin(), w.end());
        HIP_OK(hipMemcpy(c->raw_wraps, w.data(), w.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    /* + the terminator raw_apply_kernel wrote at [n]; the wrap list plays the part of qm2_plan.h's */
    /* FASTQ: a record's qualities are as long as its sequence, so at most half of the text is sequence (QM2_PREPARE_FASTQ) */
    return launch_batch(c, c->raw_out, n + 1, 0, n_wraps ? c->raw_wraps : nullptr, n_wraps, fastq ? n / 2 + 64 : 0);
}

extern "C" int qm2_submit_raw_device(qm2_ctx *c, const void *d_raw, size_t n, uint32_t fastq, uint32_t skip,
                                     uint32_t *verdict)
{
    if (!c || !verdict) return fail("qm2_submit_raw_device: null argument");
    *verdict = 0;
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (n == 0) return 0;
    if (!d_raw) return fail("qm2_submit_raw_device: null buffer");
    if ((uintptr_t)d_raw & 15u) return fail("qm2_submit_raw_device: the buffer must be 16-byte aligned");
    if (fastq && skip > 3) return fail("qm2_submit_raw_device: skip is 0..3");
    if (use_device(c)) return 1;
    return raw_classify_and_count(c, d_raw, n, fastq, skip, verdict, false);
}

extern "C" int qm2_staging_commit_raw(qm2_ctx *c, char *buf, size_t offset, size_t n, uint32_t fastq, uint32_t skip)
{
    if (!c || !buf) return fail("qm2_staging_commit_raw: null argument");
    Slot *owner = nullptr;
    for (auto &s : c->slots) if (s.lent && s.host == buf) owner = &s;
    if (!owner) return fail("qm2_staging_commit_raw: not a buffer from qm2_staging_acquire");
    owner->lent = false;
    if (n == 0 || c->raw_verdict) return 0;                   /* after a refusal nothing raw is counted any more */
    if (offset > c->staging_bytes || n > c->staging_bytes - offset)
        return fail("qm2_staging_commit_raw: more bytes than the buffer holds");
    if (n > c->batch_bytes) return fail("qm2_staging_commit_raw: block larger than a device batch");
    if (fastq && skip > 3) return fail("qm2_staging_commit_raw: skip is 0..3");
    if (use_device(c)) return 1;
    {
        Batch &b = c->batch[c->cur];
        if (b.fill && (!b.raw || b.fill + n > c->batch_bytes) && batch_flush(c)) return 1;
    }
    if (c->raw_verdict) return 0;
    Batch &b = c->batch[c->cur];
    if (batch_ready(c, b, batch_text_bytes(c))) return 1;
    if (b.fill == 0) { b.raw = true; b.raw_fastq = fastq ? 1u : 0u; b.raw_skip = skip; b.carry = 0; }
    HIP_OK(hipMemcpyAsync(b.dev + b.fill, buf + offset, n, hipMemcpyHostToDevice, c->copy_stream));
    HIP_OK(hipEventRecord(owner->done, c->copy_stream));
    owner->busy = true;
    b.fill += n;
    c->prof.staged_bytes += n;
    return 0;
}

extern "C" uint32_t qm2_raw_verdict(qm2_ctx *c) { return c ? c->raw_verdict : 0; }

extern "C" int qm2_keys_from_fd(qm2_ctx *c, int fd, uint64_t file_offset)
{
    if (!c) return fail("qm2_keys_from_fd: null context");
    if (!c->keys_pending) return fail("qm2_keys_from_fd: the context was not opened with QM2_FLAG_KEYS_LATER (or is loaded already)");
    if (use_device(c)) return 1;
    const uint64_t n_slots = c->span_hi - c->span_lo;
    if (qm2_stream_fd(c, fd, file_offset + c->span_lo * sizeof(uint64_t), c->keys, n_slots * sizeof(uint64_t), "Dictionary file"))
        return 1;
    c->keys_pending = false;
    if (c->part.reserved) {                                   /* prepared: look at the runs of occupied slots now, not in the first batch */
        bool ok = false;
        if (runs_fit(c, &ok)) return 1;
        if (table_image(c)) return 1;
    }
    return 0;
}

extern "C" int qm2_sync(qm2_ctx *c)
{
    if (!c) return fail("qm2_sync: null context");
    if (use_device(c)) return 1;
    if (batch_flush(c)) return 1;
    HIP_OK(hipStreamSynchronize(c->copy_stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    for (auto &s : c->slots) s.busy = false;
    for (auto &b : c->batch) b.busy = false;
    if (harvest_timing(c)) return 1;
    DevStats st;
    HIP_OK(hipMemcpy(&st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    if (st.overflow)
        return fail("qm2: %llu k-mers did not fit their exchange bucket", st.overflow);
    c->peek_pending = false;
    c->peek = st;
    c->peek_bytes = c->bytes_launched;
    if (c->bytes_launched) c->kmers_per_byte = std::min(1.0, (double)st.n_kmers / (double)c->bytes_launched);
    return 0;
}

extern "C" int qm2_flush(qm2_ctx *c)
{
    if (!c) return fail("qm2_flush: null context");
    if (use_device(c)) return 1;
    return batch_flush(c);
}

/* dst += src, whole counter arrays (replicas of one table on one or two devices) */
/* counters[i] += interior16[i]; interior16[i] = 0 (qm2_ctx.h): four slots a step, groups without a count are skipped */
__global__ void fold_counters_kernel(uint32_t *counters, uint16_t *interior16, uint64_t n)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t groups = n / 4;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (uint64_t)gridDim.x * blockDim.x) {
        const u32x2 v = reinterpret_cast<const u32x2 *>(interior16)[g];
        if ((v.x | v.y) == 0) continue;
        u32x4 cv = reinterpret_cast<const u32x4 *>(counters)[g];
        cv.x += v.x & 0xFFFFu; cv.y += v.x >> 16; cv.z += v.y & 0xFFFFu; cv.w += v.y >> 16;
        reinterpret_cast<u32x4 *>(counters)[g] = cv;
        reinterpret_cast<u32x2 *>(interior16)[g] = u32x2{0, 0};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint64_t i = groups * 4; i < n; i++) { counters[i] += interior16[i]; interior16[i] = 0; }
}

int qm2_fold_counters(qm2_ctx *c)
{
    if (!c->interior16 || !c->interior_dirty) return 0;
    if (use_device(c)) return 1;
    const uint64_t n_slots = c->span_hi - c->span_lo;
    const int grid = (int)std::min<uint64_t>((n_slots / 4 + 255) / 256 + 1, (uint64_t)c->n_cu * 16);
    hipLaunchKernelGGL(fold_counters_kernel, dim3(grid), dim3(256), 0, c->stream, c->counters, c->interior16, n_slots);
    HIP_OK(hipGetLastError());
    c->interior_dirty = false;
    return 0;
}

__global__ void add_counters_kernel(uint32_t *dst, const uint32_t *src, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

extern "C" int qm2_merge_counters(qm2_ctx *dst, qm2_ctx *src)
{
    if (!dst || !src || dst == src) return fail("qm2_merge_counters: two different contexts needed");
    /* replicas hold the same span; a hash-range shard's span (its own slots plus the overhang past its
     * middle-facing edge) lies inside the whole table its counts are folded into */
    if (dst->H != src->H || src->span_lo < dst->span_lo || src->span_hi > dst->span_hi)
        return fail("qm2_merge_counters: the source's slots are not part of the destination's table");
    if (qm2_sync(src) || qm2_sync(dst)) return 1;
    /* (both arrays whole: the interiors' 2-byte counts first) */
    if (qm2_fold_counters(src) || qm2_fold_counters(dst)) return 1;
    if (use_device(src)) return 1;
    HIP_OK(hipStreamSynchronize(src->stream));
    if (use_device(dst)) return 1;
    HIP_OK(hipStreamSynchronize(dst->stream));
    const uint64_t n = src->span_hi - src->span_lo;
    uint32_t *const into = dst->counters + (src->span_lo - dst->span_lo);
    const int grid = dst->n_cu * 8;
    if (dst->device == src->device) {
        hipLaunchKernelGGL(add_counters_kernel, dim3(grid), dim3(256), 0, dst->stream, into, src->counters, n);
        HIP_OK(hipGetLastError());
    } else {
        /* a piece at a time through a scratch buffer on the destination (peer access need not be enabled) */
        const uint64_t piece = std::min<uint64_t>(n, 1ull << 26);
        uint32_t *scratch = nullptr;
        HIP_OK(hipMalloc((void **)&scratch, piece * sizeof(uint32_t)));
        hipError_t e = hipSuccess;
        for (uint64_t at = 0; at < n && e == hipSuccess; at += piece) {
            const uint64_t m = std::min(piece, n - at);
            e = hipMemcpyPeerAsync(scratch, dst->device, src->counters + at, src->device, m * sizeof(uint32_t), dst->stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(add_counters_kernel, dim3(grid), dim3(256), 0, dst->stream, into + at, scratch, m);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(dst->stream);
        (void)hipFree(scratch);
        if (e != hipSuccess) return fail("qm2_merge_counters: %s", hipGetErrorString(e));
    }
    HIP_OK(hipStreamSynchronize(dst->stream));
    return 0;
}

extern "C" int qm2_stats_peek(qm2_ctx *c, qm2_stats *st)
{
    if (!c || !st) return fail("qm2_stats_peek: null argument");
    stats_peek_harvest(c);
    st->n_bases = c->peek.n_bases;
    st->n_kmers = c->peek.n_kmers;
    st->n_probes = c->peek.n_probes;
    st->n_hits = c->peek.n_hits;
    return 0;
}

/* Everything the counting phase will allocate, ahead of time (the CLI calls this while nothing is
 * waiting for it): so no hipMalloc / hipHostMalloc sits between the first submit and qm2_finish. */
extern "C" int qm2_prepare(qm2_ctx *c, uint64_t batch_bytes, uint32_t what)
{
    if (!c) return fail("qm2_prepare: null context");
    if (use_device(c)) return 1;
    const uint64_t bb = batch_bytes ? batch_bytes : (uint64_t)c->batch_bytes;
    if (what & QM2_PREPARE_HOST) {
        if (bb > c->batch_bytes) return fail("qm2_prepare: batch larger than the context's device batch");
        for (auto &b : c->batch) if (batch_ready(c, b, (what & QM2_PREPARE_PACKED) ? batch_packed_bytes(c) : batch_text_bytes(c))) return 1;
        for (auto &s : c->slots) if (qm2_slot_ready(c, s)) return 1;
    }
    if (what & QM2_PREPARE_RAW) {
        if (raw_reserve(c, bb + 1)) return 1;
    }
    const bool want_part = !(what & QM2_PREPARE_NO_PART) && !(c->flags & QM2_FLAG_DIRECT) && c->part.levels != 0 &&
                           ((c->flags & QM2_FLAG_PARTITION) || (bb >= (c->span_hi - c->span_lo) / 8 && bb >= (1u << 20)));
    if (want_part) {
        /* (no room: those batches take the direct kernel) */
        if (part_reserve(c, (what & QM2_PREPARE_FASTQ ? bb / 2 + 64 : bb) + kCarry + 64) && getenv("QM2_TIMING"))
            printf("QM2 prepare: %s; batches of that size take the direct kernel\n", g_err);
        if (!c->keys_pending && !c->keys_consumed) {
            bool ok = false;
            if (runs_fit(c, &ok)) return 1;
            if (table_image(c)) return 1;
        }
    }
    return 0;
}

extern "C" int qm2_finish(qm2_ctx *c, uint16_t *depth_out, qm2_stats *st)
{
    if (qm2_sync(c)) return 1;
    /* the counters are complete from here on (own array or the caller's ext_counters) */
    if (c->interior_dirty) { if (qm2_fold_counters(c)) return 1; HIP_OK(hipStreamSynchronize(c->stream)); }
    const uint64_t n_slots = c->span_hi - c->span_lo;
    if (depth_out) {
        if (!c->narrow) HIP_OK(hipMalloc((void **)&c->narrow, n_slots * sizeof(uint16_t)));
        const int grid = (int)std::min<uint64_t>((n_slots / 8 + kBlock - 1) / kBlock + 1, (uint64_t)c->n_cu * 8);
        hipLaunchKernelGGL(narrow_kernel, dim3(grid), dim3(kBlock), 0, c->stream, c->counters, c->narrow, n_slots);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(depth_out, c->narrow, n_slots * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    if (st) {
        DevStats d;
        HIP_OK(hipMemcpy(&d, c->stats, sizeof d, hipMemcpyDeviceToHost));
        st->n_bases = d.n_bases;
        st->n_kmers = d.n_kmers;
        st->n_probes = d.n_probes;
        st->n_hits = d.n_hits;
    }
    return 0;
}

extern "C" int qm2_reset(qm2_ctx *c)
{
    if (qm2_sync(c)) return 1;
    const uint64_t n_slots = c->span_hi - c->span_lo;
    HIP_OK(hipMemsetAsync(c->counters, 0, n_slots * sizeof(uint32_t), c->stream));
    if (c->interior16) HIP_OK(hipMemsetAsync(c->interior16, 0, n_slots * sizeof(uint16_t), c->stream));
    c->interior_dirty = false;
    HIP_OK(hipMemsetAsync(c->stats, 0, sizeof(DevStats), c->stream));
    c->bytes_launched = 0;
    c->peek_pending = false;
    c->peek = DevStats();
    c->peek_bytes = 0;
    c->raw_verdict = 0;
    c->raw_shift = 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    memset(&c->prof, 0, sizeof c->prof);
    c->n_allocations = 0;
    return 0;
}

extern "C" int qm2_keys_changed(qm2_ctx *c)
{
    if (!c) return fail("qm2_keys_changed: null context");
    if (qm2_sync(c)) return 1;
    std::lock_guard<std::mutex> guard(c->runs_mu);
    c->img_ready = false;                                     /* the table image is made again (table_image) */
    c->part.runs = 0;                                         /* ... and so is the look at the runs of occupied slots (runs_fit) */
    if (c->part.shared) { (void)hipFree(c->part.shared); c->part.shared = nullptr; }
    return 0;
}

extern "C" int qm2_get_profile(qm2_ctx *c, qm2_profile *p)
{
    if (!c || !p) return fail("qm2_get_profile: null argument");
    if (qm2_sync(c)) return 1;
    DevStats st;
    HIP_OK(hipMemcpy(&st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    c->prof.fallback_batches = st.fallbacks;
    c->prof.spilled_kmers = st.spilled;
    c->prof.allocations = c->n_allocations.load();
    *p = c->prof;
    return 0;
}

#ifdef QM2_STAMPS
/* diagnostic build only: read and clear the phase stamps of bucket_count_kernel */
extern "C" int qm2_debug_stamps(unsigned long long *out32)
{
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(qm2::qm2_stamp_acc), 32 * sizeof(unsigned long long)));
    unsigned long long z[32] = {0};
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(qm2::qm2_stamp_acc), z, sizeof z));
    return 0;
}
#endif

/* ---- split path ------------------------------------------------------------ */

extern "C" int qm2_encode_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup,
                                 uint64_t n_sup, uint64_t *d_out, uint64_t bucket_cap,
                                 uint32_t n_buckets, uint64_t *d_counts)
{
    if (!c) return fail("qm2_encode_device: null context");
    if (n == 0) return 0;
    if (!d_seq || !d_out || !d_counts) return fail("qm2_encode_device: null buffer");
    if (n_buckets == 0 || (n_buckets & (n_buckets - 1)) || n_buckets > c->H)
        return fail("qm2_encode_device: bucket count %u is not a power of two <= H", n_buckets);
    if (use_device(c)) return 1;
    const SeqView sv = seq_view((const uint8_t *)d_seq, n, 0, d_sup, n_sup);
    const uint64_t n_tiles = (sv.n_total + kTileBytes - 1) / kTileBytes;
    uint32_t lgB = 0;
    while ((1u << lgB) < n_buckets) lgB++;
    const uint32_t owner_shift = c->lgH - lgB;
    const int grid = grid_for(c, n_tiles);
    TimedLaunch t;
    if (time_begin(c, &t, QM2_K_ENCODE)) return 1;
    if (c->k == 30)
        hipLaunchKernelGGL(encode_kernel<true>, dim3(grid), dim3(kBlock), 0, c->stream, sv, c->k,
                           (uint32_t)(c->H - 1), n_tiles, d_out, bucket_cap, n_buckets, owner_shift,
                           (unsigned long long *)d_counts, c->stats);
    else
        hipLaunchKernelGGL(encode_kernel<false>, dim3(grid), dim3(kBlock), 0, c->stream, sv, c->k,
                           (uint32_t)(c->H - 1), n_tiles, d_out, bucket_cap, n_buckets, owner_shift,
                           (unsigned long long *)d_counts, c->stats);
    HIP_OK(hipGetLastError());
    return time_end(c, &t, true);
}

extern "C" int qm2_probe_device(qm2_ctx *c, const uint64_t *d_kmers, uint64_t n, const uint64_t *d_n)
{
    if (!c) return fail("qm2_probe_device: null context");
    if (n == 0) return 0;
    if (!d_kmers) return fail("qm2_probe_device: null buffer");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (use_device(c)) return 1;
    const uint64_t groups = (n + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
    const int grid = grid_for(c, groups);
    TimedLaunch t;
    if (time_begin(c, &t, QM2_K_PROBE)) return 1;
    hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(kBlock), 0, c->stream, d_kmers, n,
                       (const unsigned long long *)d_n, (const uint32_t *)nullptr, table_view(c), c->stats,
                       (const unsigned long long *)nullptr, (const unsigned long long *)nullptr);
    HIP_OK(hipGetLastError());
    return time_end(c, &t, true);
}

static int route_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup, uint64_t n_sup,
                        uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t xcd_subs, uint32_t *d_fill, uint64_t *d_call,
                        uint32_t carry = 0);

extern "C" int qm2_route_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup, uint64_t n_sup,
                                uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t *d_fill, uint64_t *d_call)
{
    return route_device(c, d_seq, n, d_sup, n_sup, d_slabs, slab_cap, n_owners, 1, d_fill, d_call);
}

extern "C" int qm2_route_device_carry(qm2_ctx *c, const void *d_seq, size_t n, uint32_t carry, const uint64_t *d_sup, uint64_t n_sup,
                                      uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t *d_fill, uint64_t *d_call)
{
    if (carry > kCarry || carry > n) return fail("qm2_route_device_carry: at most %u look-back bytes, and no more than the text", (unsigned)kCarry);
    return route_device(c, d_seq, n, d_sup, n_sup, d_slabs, slab_cap, n_owners, 1, d_fill, d_call, carry);
}

extern "C" int qm2_route_xcd_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup, uint64_t n_sup,
                                    uint64_t *d_slabs, uint64_t region_cap, uint32_t n_regions, uint32_t *d_fill, uint64_t *d_call)
{
    return route_device(c, d_seq, n, d_sup, n_sup, d_slabs, region_cap, n_regions, 8, d_fill, d_call);
}

extern "C" int qm2_route_xcd_device_carry(qm2_ctx *c, const void *d_seq, size_t n, uint32_t carry, const uint64_t *d_sup, uint64_t n_sup,
                                          uint64_t *d_slabs, uint64_t region_cap, uint32_t n_regions, uint32_t *d_fill, uint64_t *d_call)
{
    if (carry > kCarry || carry > n) return fail("qm2_route_xcd_device_carry: at most %u look-back bytes, and no more than the text", (unsigned)kCarry);
    return route_device(c, d_seq, n, d_sup, n_sup, d_slabs, region_cap, n_regions, 8, d_fill, d_call, carry);
}

static int route_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup, uint64_t n_sup,
                        uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_owners, uint32_t xcd_subs, uint32_t *d_fill, uint64_t *d_call,
                        uint32_t carry)
{
    if (!c) return fail("qm2_route_device: null context");
    if (n == 0) return 0;
    if (!d_seq || !d_slabs || !d_fill || !d_call) return fail("qm2_route_device: null buffer");
    if (n_owners == 0 || (n_owners & (n_owners - 1)) || n_owners > (uint32_t)kMaxFan || n_owners > c->H)
        return fail("qm2_route_device: owner count %u is not a power of two <= %d", n_owners, kMaxFan);
    if ((slab_cap & 1u) || ((uintptr_t)d_slabs & 15u) || slab_cap > 0xFFFFFFFFull)
        return fail("qm2_route_device: slabs must be 16-byte aligned, of an even capacity below 2^32");
    if (use_device(c)) return 1;
    const SeqView sv = seq_view((const uint8_t *)d_seq, n, carry, d_sup, n_sup);
    uint32_t lgG = 0;
    while ((1u << lgG) < n_owners) lgG++;
    SplitDst d = {d_slabs, d_fill, (uint32_t)slab_cap, nullptr, nullptr, 0, 0, xcd_subs};
    const uint64_t n_chunks = (sv.n_total + kSplitTextBytes - 1) / kSplitTextBytes;
    const int grid = (int)std::min<uint64_t>(n_chunks, (uint64_t)c->n_cu * kSplitWgPerCu);
    TimedLaunch t;
    if (time_begin(c, &t, QM2_K_ENCODE)) return 1;
    if (c->k == 30)
        hipLaunchKernelGGL((split_text_kernel<true, false>), dim3(grid), dim3(kSplitThreads), 0, c->stream, sv, c->k,
                           c->lgH - lgG, lgG, n_chunks, d, c->stats, (unsigned long long *)d_call);
    else
        hipLaunchKernelGGL((split_text_kernel<false, false>), dim3(grid), dim3(kSplitThreads), 0, c->stream, sv, c->k,
                           c->lgH - lgG, lgG, n_chunks, d, c->stats, (unsigned long long *)d_call);
    HIP_OK(hipGetLastError());
    return time_end(c, &t, true);
}

extern "C" int qm2_probe_slabs_device(qm2_ctx *c, const uint64_t *d_slabs, uint64_t slab_cap, uint32_t n_slabs,
                                      const uint32_t *d_fill)
{
    if (!c) return fail("qm2_probe_slabs_device: null context");
    if (n_slabs == 0 || slab_cap == 0) return 0;
    if (!d_slabs || !d_fill) return fail("qm2_probe_slabs_device: null buffer");
    if (slab_cap > 0xFFFFFFFFull) return fail("qm2_probe_slabs_device: slab capacity must be below 2^32 (fills are 32-bit)");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (c->keys_consumed) return fail("qm2: the key table was overwritten by qm2_chain_gather; this context cannot count any more");
    if (use_device(c)) return 1;
    if (part_plan_kind(c, 0)) return 1;                               /* (after qm2_probe_subslabs_device: the plan of our own level 0 again) */
    const uint64_t n_upper = slab_cap * n_slabs;
    bool part = c->part.levels != 0 && !(c->flags & QM2_FLAG_DIRECT) && !(slab_cap & 1u) && !((uintptr_t)d_slabs & 15u) &&
                ((c->flags & QM2_FLAG_PARTITION) || (n_upper >= (c->span_hi - c->span_lo) / 16 && n_upper >= (1u << 17)));
    if (part) {
        bool ok = false;
        if (runs_fit(c, &ok)) return 1;
        if (!ok) part = false;
    }
    if (part && part_reserve(c, n_upper)) part = false;
    if (part && table_image(c)) return 1;
    c->bytes_launched += n_upper;
    if (part) {
        c->prof.partitioned_batches++;
        PartSource src;
        src.slabs = d_slabs; src.slab_cap = slab_cap; src.n_slabs = n_slabs; src.fill = d_fill; src.n_upper = n_upper; src.n_bytes = n_upper;
        return launch_partitioned(c, src);
    }
    c->prof.direct_batches++;
    for (uint32_t g = 0; g < n_slabs; g++) {
        const uint64_t groups = (slab_cap + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
        TimedLaunch t;
        if (time_begin(c, &t, QM2_K_PROBE)) return 1;
        hipLaunchKernelGGL(probe_kernel, dim3(grid_for(c, groups)), dim3(kBlock), 0, c->stream, d_slabs + g * slab_cap, slab_cap,
                           (const unsigned long long *)nullptr, d_fill + g, table_view(c), c->stats,
                           (const unsigned long long *)nullptr, (const unsigned long long *)nullptr);
        HIP_OK(hipGetLastError());
        if (time_end(c, &t, g + 1 == n_slabs)) return 1;
    }
    return 0;
}

extern "C" int qm2_probe_subslabs_device(qm2_ctx *c, const uint64_t *d_slabs, uint64_t sub_cap, uint32_t n_peers, uint32_t n_subs,
                                         uint32_t peer_stride, const uint32_t *d_fill, uint64_t kmers_bound)
{
    if (peer_stride && peer_stride < n_subs) return fail("qm2_probe_subslabs_device: peers closer together than their sub-slabs");
    if (!peer_stride) peer_stride = n_subs;
    if (!c) return fail("qm2_probe_subslabs_device: null context");
    if (n_peers == 0 || n_subs == 0 || sub_cap == 0) return 0;
    if (!d_slabs || !d_fill) return fail("qm2_probe_subslabs_device: null buffer");
    if (c->keys_pending) return fail("qm2: the key table has not been loaded yet (qm2_keys_from_fd)");
    if (c->keys_consumed) return fail("qm2: the key table was overwritten by qm2_chain_gather; this context cannot count any more");
    if ((n_subs & (n_subs - 1)) || n_subs > (uint32_t)kMaxFan || (sub_cap & 1u) || sub_cap > 0xFFFFFFFFull || ((uintptr_t)d_slabs & 15u))
        return fail("qm2_probe_subslabs_device: sub-slabs must be a power of two of them (<= %d), 16-byte aligned, of an even capacity below 2^32", kMaxFan);
    if (c->own_hi == 0) return fail("qm2_probe_subslabs_device: the context has no own range (qm2_opts.own_lo / own_hi)");
    if (use_device(c)) return 1;
    uint32_t lg_subs = 0;
    while ((1u << lg_subs) < n_subs) lg_subs++;
    /* the plan: level 0 = the sub-slabs (its children are the senders' sub-regions), the rest as usual */
    PartPlan &p = c->part;
    if (lg_subs == 0) return fail("qm2_probe_subslabs_device: one sub-slab per peer is qm2_probe_slabs_device's case");
    if (part_plan_kind(c, (int)lg_subs)) return 1;
    if (p.levels < 2 || (c->flags & QM2_FLAG_DIRECT))
        return fail("qm2_probe_subslabs_device: %u sub-slabs leave this context's range nothing to split (or QM2_FLAG_DIRECT is set): use qm2_probe_slabs_device", n_subs);
    bool ok = false;
    if (runs_fit(c, &ok)) return 1;
    const uint64_t bound = kmers_bound ? kmers_bound : sub_cap * n_subs * (uint64_t)n_peers;
    c->bytes_launched += bound;
    if (!ok || part_reserve(c, bound)) {
        /* a table with hopelessly long runs of occupied slots, or no memory: every region probed directly */
        c->prof.direct_batches++;
        const uint64_t groups = (sub_cap + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
        TimedLaunch t;
        if (time_begin(c, &t, QM2_K_PROBE)) return 1;
        hipLaunchKernelGGL(probe_regions_kernel, dim3(grid_for(c, std::min<uint64_t>(groups * n_subs * n_peers, 1u << 16))), dim3(kBlock), 0, c->stream,
                           d_slabs, sub_cap, n_peers, n_subs, peer_stride, d_fill, table_view(c), c->stats, (const unsigned long long *)nullptr);
        HIP_OK(hipGetLastError());
        return time_end(c, &t, true);
    }
    if (table_image(c)) return 1;
    c->prof.partitioned_batches++;
    PartSource src;
    src.slabs = d_slabs; src.slab_cap = sub_cap; src.n_slabs = n_peers; src.n_subs = n_subs; src.peer_stride = peer_stride; src.fill = d_fill;
    src.n_upper = bound; src.n_bytes = bound;
    return launch_partitioned(c, src);
}

/* ---- host placement ---------------------------------------------------------- */

extern "C" int qm2_bind_near_device(int device, uint32_t *n_cpus)
{
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) { if (n_cpus) *n_cpus = 0; return 0; }
    if (n_cpus) *n_cpus = (uint32_t)CPU_COUNT(&cur);
    if (getenv("QM2_NO_BIND")) return 0;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { (void)hipGetLastError(); return 0; }   /* (a hint: whoever needs the device says so) */
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= n_dev) return 0;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return 0;
    for (char *q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    int node = -1;
    if (!f) return 0;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return 0;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return 0;
    char list[4096] = "";
    if (!fgets(list, sizeof list, f)) list[0] = 0;
    fclose(f);
    cpu_set_t want;
    CPU_ZERO(&want);
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int got = sscanf(tok, "%d-%d", &a, &b);
        if (got == 1) b = a;
        if (got < 1) continue;
        for (int c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &cur)) CPU_SET(c, &want);
    }
    if (CPU_COUNT(&want) == 0) return 0;                         /* (none of the node's CPUs is ours: leave things as they are) */
    if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
    if (n_cpus) *n_cpus = (uint32_t)CPU_COUNT(&want);
    return 0;
}

/* ---- pure host helper ------------------------------------------------------ */

extern "C" int qm2_shard_span(const uint64_t *keys, uint64_t H, uint32_t n_shards, uint32_t g,
                              uint64_t *own_lo, uint64_t *own_hi, uint64_t *span_lo,
                              uint64_t *span_hi)
{
    if (!keys || H < 2 || (H & (H - 1)) || n_shards == 0 || (n_shards & (n_shards - 1)) ||
        n_shards > H || g >= n_shards)
        return fail("qm2_shard_span: bad arguments");
    const uint64_t per = H / n_shards;
    const uint64_t lo = (uint64_t)g * per, hi = lo + per;
    uint64_t slo = lo, shi = hi;
    if (n_shards > 1) {
        /* walks go towards the middle (:94-95) and stop ON the first empty slot: extend the
         * span over the occupied run past the middle-facing edge, empty slot included */
        if (lo < (H >> 1)) while (shi < H && keys[shi - 1] != 0) shi++;
        else               while (slo > 0 && keys[slo] != 0) slo--;
        /* whole 64-slot blocks (more slots never hurt): 16-byte accesses to the span's keys and
         * counters stay aligned and never straddle its ends */
        if (H >= 64) { slo &= ~(uint64_t)63; shi = (shi + 63) & ~(uint64_t)63; }
    }
    if (own_lo) *own_lo = lo;
    if (own_hi) *own_hi = hi;
    if (span_lo) *span_lo = slo;
    if (span_hi) *span_hi = shi;
    return 0;
}
