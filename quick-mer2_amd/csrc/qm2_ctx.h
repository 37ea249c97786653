/*
 * qm2_ctx.h — the context structure shared by the translation units of
 * libqm2_hip.so (qm2_hip.hip: counting; qm2_epilogue.hip: chain gather).
 */
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>

#include "../../include/qm2.h"
#include "qm2_device.hip.h"

namespace qm2 {

constexpr int kMaxLevels = 3;

struct Slot {                               /* one pinned staging buffer */
    char       *host = nullptr;
    hipEvent_t  done = nullptr;             /* its H2D copy has finished */
    bool        busy = false;
    bool        lent = false;               /* handed out by qm2_staging_acquire */
};

struct Batch {                              /* one device batch being filled by qm2_submit */
    uint8_t    *dev = nullptr;
    size_t      cap = 0;                    /* bytes at dev: a batch of ASCII / raw text, or (QM2_PREPARE_PACKED) only of packed blocks */
    size_t      fill = 0;
    size_t      carry = 0;                  /* leading bytes that only serve as look-back */
    std::vector<uint64_t> wraps;            /* uint16-wrap positions, relative to dev[0]  */
    uint64_t   *dev_wraps = nullptr;
    size_t      wraps_cap = 0;
    hipEvent_t  done = nullptr;             /* its kernels have finished */
    hipEvent_t  copied = nullptr;           /* its staged copies have landed (the kernels wait for it) */
    bool        busy = false;
    bool        packed = false;             /* holds PACKED sequence (qm2_staging_commit_packed): fill = bytes of blocks ... */
    uint64_t    packed_nl = 0;              /* ... of whose positions this many are line ends or filler */
    bool        raw = false;                /* holds raw FASTA/FASTQ text (qm2_staging_commit_raw) ... */
    uint32_t    raw_fastq = 0, raw_skip = 0;/* ... in this line state at dev[0] */
};

struct TimedLaunch {
    hipEvent_t a, b;
    int cls;
    bool batch_end;
};

struct PartPlan {                           /* how a batch is radix-partitioned for this table */
    int      levels = 0;
    uint32_t lg_range = 0;                  /* log2 of the own range the finest buckets tile */
    uint32_t bits[kMaxLevels] = {0, 0, 0};
    uint32_t shift[kMaxLevels] = {0, 0, 0}; /* child id at level l = (home >> shift[l]) & (fan-1) */
    uint64_t buckets[kMaxLevels] = {0, 0, 0};
    /* sized for `reserved` payload bytes.  (Atomic, like qm2_ctx::keys_pending: the CLI makes the partition buffers —
     * qm2_prepare — on one thread while qm2_keys_from_fd streams the dictionary on another; each looks at the other's
     * flag to decide who takes the one-off look at the runs of occupied slots, itself under runs_mu.) */
    std::atomic<uint64_t> reserved{0};
    uint64_t refused = 0;                   /* smallest batch whose buffers did not fit in HBM */
    uint32_t cap[kMaxLevels] = {0, 0, 0};
    bool     ext0 = false;                  /* level 0 comes from outside (qm2_probe_subslabs_device: the sub-slabs a shard received ARE its level 0) */
    int      max_bits = 10;
    uint32_t subs0 = 1;                     /* level 0: sub-regions per child (qm2_partition.hip.h, sub_region0) */
    bool     seg0 = false;                  /* level 0's buffer is segmented (qm2_partition.hip.h, SplitDst) */
    uint64_t *buf[2] = {nullptr, nullptr};
    uint64_t  buf_slots[2] = {0, 0};
    uint32_t *cursors = nullptr;            /* all levels, contiguous */
    uint64_t  cursor_off[kMaxLevels] = {0, 0, 0};
    uint64_t  n_cursors = 0;
    uint64_t *spill = nullptr;
    uint64_t  spill_cap = 0;
    unsigned long long *spill_count = nullptr;
    /* occupied runs vs the staged halo (long_run_kernel): 0 not checked yet, 1 every run is shorter
     * than the halo, 2 some are longer and their buckets are marked in `shared`, 3 hopeless -> direct */
    int       runs = 0;
    uint32_t *shared = nullptr;             /* bitset over the finest buckets */
};

}  // namespace qm2

struct qm2_ctx {
    int         device = 0;
    uint32_t    k = 30;
    uint64_t    H = 0;
    uint32_t    lgH = 0;
    uint64_t    span_lo = 0, span_hi = 0;
    uint64_t    own_lo = 0, own_hi = 0;     /* home slots this context is fed (= the whole table, or a shard's own range) */
    uint32_t    flags = 0;
    uint64_t   *keys = nullptr;
    bool        own_keys = false;
    /* The counts of bucket_count's PRIVATE bucket interiors live in 2 bytes per slot (the reference's own width, :23):
     * that kernel is bound by its bytes, and 4-byte counters are a quarter of its table pass.  Everything atomic —
     * bucket edges, halo, shared buckets, the spill list, the direct kernel — adds into the 32-bit array; a tile value
     * above 65535 sends its upper part there too, so count = counters[i] + interior16[i] exactly.  qm2_fold_counters
     * adds the small array into the big one before anybody looks (qm2_finish, merge, the chain gather). */
    uint16_t   *interior16 = nullptr;
    bool        interior_dirty = false;
    /* The table image (qm2_partition.hip.h, table_image_kernel): every resident slot as a bucket record — 4 bytes of R
     * and a second word of rec_bytes - 4 — made once per key table, behind the look at the runs of occupied slots.  It
     * is what bucket_count streams instead of the keys (6 bytes a slot at hg38 scale, nothing to hash). */
    uint32_t   *img_r = nullptr;
    void       *img_w = nullptr;
    bool        img_ready = false;
    uint32_t    rec_bytes = 0;              /* 6 or 8: a bucket record (finest level of the partition, table image); 0: no plan */
    bool        keys_consumed = false;      /* the chain was uploaded over the key table */
    std::atomic<bool> keys_pending{false};  /* QM2_FLAG_KEYS_LATER: qm2_keys_from_fd has not run yet */
    std::atomic<uint64_t> n_allocations{0}; /* hipMalloc / hipHostMalloc calls so far (qm2_profile.allocations; counted from both threads) */
    uint32_t   *counters = nullptr;
    bool        own_counters = false;
    uint16_t   *narrow = nullptr;           /* H x u16 scratch for qm2_finish */
    qm2::DevStats *stats = nullptr;
    /* k-mers per input byte seen so far (sizes bucket_count's group): the stats block is copied
     * to pinned memory behind a batch now and then and read when that copy has landed */
    qm2::DevStats *stats_peek = nullptr;
    hipEvent_t  peek_evt = nullptr;
    bool        peek_pending = false;
    uint64_t    peek_bytes = 0, peek_bytes_pending = 0, bytes_launched = 0;
    qm2::DevStats peek = {};                /* latest snapshot that has landed (qm2_stats_peek) */
    double      kmers_per_byte = 1.0;
    double      kappa_hint = 1.0;           /* qm2_opts.kmers_per_kib / 1024: sizes the partition buffers */
    hipStream_t stream = nullptr;           /* every kernel of this context runs here, in order */
    bool        own_stream = false;
    hipStream_t copy_stream = nullptr;      /* H2D staging copies */
    std::vector<qm2::Slot> slots;
    size_t      staging_bytes = 0;
    uint32_t    next_slot = 0;
    qm2::Batch  batch[2];
    size_t      batch_bytes = 0;
    int         cur = 0;
    int         n_cu = 256;
    std::vector<qm2::TimedLaunch> timed;
    std::vector<qm2::TimedLaunch> pool;
    qm2_profile prof;
    qm2::PartPlan part;
    /* device-side line discipline (qm2_raw.hip.h): neutralised copy of a raw block + tile scratch */
    uint8_t    *raw_out = nullptr;
    uint64_t    raw_cap = 0;
    void       *raw_tiles = nullptr;        /* scan scratch: 40 bytes per 4 KiB tile */
    uint64_t    raw_tile_cap = 0;
    unsigned int *raw_flags = nullptr;      /* [0] refusal bits, [1] wrap count, [2..3] first unlisted desync line */
    uint64_t   *raw_wraps = nullptr;        /* where the uint16 run counter wraps (qm2_raw.hip.h), sorted */
    uint32_t    raw_shift = 0;              /* '>' lines at FASTQ sequence positions so far in this text, mod 4 */
    uint32_t    raw_verdict = 0;            /* sticky: a raw batch was refused (QM2_RAW_*); nothing raw is counted after it */
    std::mutex  runs_mu;                    /* the one-off look at the runs of occupied slots: qm2_prepare on one thread,
                                               qm2_keys_from_fd on another (QM2_PREPARE_NO_PART) */
    /* epilogue (qm2_epilogue.hip) */
    uint32_t   *chain = nullptr;            /* device copy of next[]; aliases keys when owned */
    bool        own_chain = false;
    uint16_t   *gathered = nullptr;         /* depth in chain order */
    uint64_t    n_gathered = 0;
};

int qm2_fail(const char *fmt, ...);
int qm2_fold_counters(qm2_ctx *c);           /* interior16 -> counters (on c->stream, not synchronised); no-op when nothing is pending */         /* sets qm2_last_error(), prints, returns 1 */
void qm2_part_free(qm2::PartPlan &p);
/* qm2_io.hip: nbytes of `fd` from `off` -> device memory `dst`, read by several threads through pinned buffers */
int qm2_stream_fd(qm2_ctx *c, int fd, uint64_t off, void *dst, uint64_t nbytes, const char *what);
/* qm2_hip.hip: make staging slot `s` usable (allocates its pinned buffer on first use, waits for its last copy) */
int qm2_slot_ready(qm2_ctx *c, qm2::Slot &s);

#define HIP_OK(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return qm2_fail("qm2: %s failed: %s", #call, hipGetErrorString(e_));  \
    } while (0)
