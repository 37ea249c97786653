/*
 * qm2_hip.hip — the HIP implementation of include/qm2.h (libqm2_hip.so).
 *
 * Host side of the counting phase: device residency of the dictionary
 * (QuicKmer.c:353-366), a pinned multi-buffer H2D pipeline that replaces the
 * reference's producer/worker FIFOs (:368-384, :421-439), and the drain
 * (:458-479).  The kernels live in qm2_kernels.hip.h.
 *
 * There is deliberately no CPU implementation behind this ABI: without a
 * usable HIP device every call fails loudly.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/qm2.h"
#include "qm2_kernels.hip.h"
#include "qm2_plan.h"

using namespace qm2;

/* ---- errors: puts(msg); return 1;  (QuicKmer.c:354-358) -------------------- */

static thread_local char g_err[512] = "";

static int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    puts(g_err);
    fflush(stdout);
    return 1;
}

#define HIP_OK(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail("qm2: %s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

extern "C" const char *qm2_last_error(void) { return g_err; }
extern "C" uint32_t qm2_abi_version(void) { return QM2_ABI_VERSION; }

/* ---- context --------------------------------------------------------------- */

namespace {

constexpr size_t kDevPad = 64;

struct Slot {
    char       *host = nullptr;            /* pinned */
    uint8_t    *dev = nullptr;
    uint64_t   *dev_sup = nullptr;         /* suppress list for this slot */
    size_t      sup_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t  done = nullptr;
    bool        busy = false;
};

struct TimedLaunch {
    hipEvent_t a, b;
};

}  // namespace

struct qm2_ctx {
    int         device = 0;
    uint32_t    k = 30;
    uint64_t    H = 0;
    uint64_t    span_lo = 0, span_hi = 0;
    uint32_t    flags = 0;
    uint64_t   *keys = nullptr;
    bool        own_keys = false;
    uint32_t   *counters = nullptr;
    bool        own_counters = false;
    uint16_t   *narrow = nullptr;          /* H x u16 scratch for qm2_finish */
    DevStats   *stats = nullptr;
    hipStream_t stream = nullptr;          /* device-resident submissions */
    bool        own_stream = false;
    std::vector<Slot> slots;
    size_t      staging_bytes = 0;
    uint32_t    next_slot = 0;
    int         n_cu = 256;
    std::vector<TimedLaunch> timed;        /* recorded, not yet read */
    std::vector<TimedLaunch> pool;         /* free event pairs */
    qm2_profile prof = {0, 0.0, 0.0};
    uint64_t    host_bases_hint = 0;
};

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

static int time_begin(qm2_ctx *c, hipStream_t s, TimedLaunch *t)
{
    if (!(c->flags & QM2_FLAG_TIMING)) return 0;
    if (!c->pool.empty()) {
        *t = c->pool.back();
        c->pool.pop_back();
    } else {
        HIP_OK(hipEventCreate(&t->a));
        HIP_OK(hipEventCreate(&t->b));
    }
    HIP_OK(hipEventRecord(t->a, s));
    return 0;
}

static int time_end(qm2_ctx *c, hipStream_t s, TimedLaunch *t)
{
    if (!(c->flags & QM2_FLAG_TIMING)) return 0;
    HIP_OK(hipEventRecord(t->b, s));
    c->timed.push_back(*t);
    return 0;
}

static int harvest_timing(qm2_ctx *c)
{
    for (auto &t : c->timed) {
        float ms = 0.f;
        HIP_OK(hipEventSynchronize(t.b));
        HIP_OK(hipEventElapsedTime(&ms, t.a, t.b));
        c->prof.n_launches++;
        c->prof.kernel_ms += ms;
        c->prof.last_kernel_ms = ms;
        c->pool.push_back(t);
    }
    c->timed.clear();
    return 0;
}

/* Launch the fused kernel over a device buffer. `data` points at payload byte
 * 0; `carry` bytes before emit start belong to the previous part of a line. */
static int launch_count(qm2_ctx *c, hipStream_t s, const uint8_t *data, size_t n, size_t carry,
                        const uint64_t *d_sup, uint64_t n_sup)
{
    if (n == 0) return 0;
    SeqView sv;
    const uintptr_t p = (uintptr_t)data;
    sv.lead = p & 15u;
    sv.base = (const uint8_t *)(p - sv.lead);
    sv.emit_from = sv.lead + carry;
    sv.n_total = sv.lead + n;
    sv.suppress = d_sup;
    sv.n_suppress = n_sup;
    const uint64_t n_tiles = (sv.n_total + kTileBytes - 1) / kTileBytes;
    const int grid = grid_for(c, n_tiles);
    DevStats *st = c->stats;
    TimedLaunch t;
    if (time_begin(c, s, &t)) return 1;
    if (c->k == 30)
        hipLaunchKernelGGL(count_kernel<true>, dim3(grid), dim3(kBlock), 0, s, sv, table_view(c), c->k, n_tiles, st);
    else
        hipLaunchKernelGGL(count_kernel<false>, dim3(grid), dim3(kBlock), 0, s, sv, table_view(c), c->k, n_tiles, st);
    HIP_OK(hipGetLastError());
    return time_end(c, s, &t);
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
    if (!keys && !o.ext_keys) return fail("qm2_open: no keys given");

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail("qm2_open: no usable HIP device (%s); there is no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));

    qm2_ctx *c = new qm2_ctx();
    if (o.device >= 0) c->device = o.device;
    else if (hipGetDevice(&c->device) != hipSuccess) c->device = 0;
    if (c->device >= n_dev) {
        delete c;
        return fail("qm2_open: device %d out of range (%d present)", o.device, n_dev);
    }
    c->k = k;
    c->H = hash_size;
    c->flags = o.flags;
    c->span_lo = o.span_lo;
    c->span_hi = o.span_hi ? o.span_hi : hash_size;
    if (c->span_lo >= c->span_hi || c->span_hi > hash_size) {
        delete c;
        return fail("qm2_open: bad span [%llu, %llu)", (unsigned long long)o.span_lo,
                    (unsigned long long)o.span_hi);
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
    if (o.ext_keys) c->keys = (uint64_t *)o.ext_keys;
    else {
        OPEN_OK(hipMalloc((void **)&c->keys, n_slots * sizeof(uint64_t)));
        c->own_keys = true;
        OPEN_OK(hipMemcpy(c->keys, keys, n_slots * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    if (o.ext_counters) c->counters = (uint32_t *)o.ext_counters;
    else {
        OPEN_OK(hipMalloc((void **)&c->counters, n_slots * sizeof(uint32_t)));
        c->own_counters = true;
        OPEN_OK(hipMemset(c->counters, 0, n_slots * sizeof(uint32_t)));
    }
    OPEN_OK(hipMalloc((void **)&c->stats, sizeof(DevStats)));
    OPEN_OK(hipMemset(c->stats, 0, sizeof(DevStats)));

    c->staging_bytes = o.staging_bytes ? (size_t)o.staging_bytes : ((size_t)64 << 20);
    if (c->staging_bytes < 4096) c->staging_bytes = 4096;
    const uint32_t n_staging = o.n_staging ? o.n_staging : 3;
    c->slots.resize(n_staging);          /* buffers are allocated lazily by qm2_submit */
#undef OPEN_OK
    *out = c;
    return 0;
}

extern "C" void qm2_close(qm2_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto &s : c->slots) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.dev) (void)hipFree(s.dev);
        if (s.dev_sup) (void)hipFree(s.dev_sup);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto &t : c->pool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    if (c->own_keys && c->keys) (void)hipFree(c->keys);
    if (c->own_counters && c->counters) (void)hipFree(c->counters);
    if (c->narrow) (void)hipFree(c->narrow);
    if (c->stats) (void)hipFree(c->stats);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void *qm2_counters_device(qm2_ctx *c) { return c ? c->counters : nullptr; }
extern "C" void *qm2_keys_device(qm2_ctx *c) { return c ? c->keys : nullptr; }

/* ---- host-buffer pipeline -------------------------------------------------- */

static int slot_acquire(qm2_ctx *c, Slot **out)
{
    Slot &s = c->slots[c->next_slot];
    c->next_slot = (c->next_slot + 1) % (uint32_t)c->slots.size();
    if (!s.host) {
        HIP_OK(hipHostMalloc((void **)&s.host, c->staging_bytes + kCarry, hipHostMallocDefault));
        HIP_OK(hipMalloc((void **)&s.dev, c->staging_bytes + kCarry + kDevPad));
        HIP_OK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    if (s.busy) {
        HIP_OK(hipEventSynchronize(s.done));
        s.busy = false;
    }
    *out = &s;
    return 0;
}

extern "C" int qm2_submit(qm2_ctx *c, const char *buf, size_t n)
{
    if (!c) return fail("qm2_submit: null context");
    if (n == 0) return 0;
    if (!buf) return fail("qm2_submit: null buffer");
    if (buf[n - 1] != '\n')
        return fail("qm2_submit: buffer must end with a newline (whole sequence lines only)");
    if (use_device(c)) return 1;

    std::vector<Part> parts;
    plan_parts(buf, n, c->staging_bytes, parts);
    for (const Part &pt : parts) {
        Slot *s;
        if (slot_acquire(c, &s)) return 1;
        const size_t staged = pt.carry + pt.take;
        memcpy(s->host, buf + pt.off - pt.carry, staged);
        uint64_t *d_sup = nullptr;
        const uint64_t n_sup = pt.wraps.size();
        if (n_sup) {                       /* long lines are rare: a small blocking copy is fine */
            if (s->sup_cap < n_sup) {
                if (s->dev_sup) HIP_OK(hipFree(s->dev_sup));
                s->sup_cap = std::max<size_t>(n_sup, 1024);
                HIP_OK(hipMalloc((void **)&s->dev_sup, s->sup_cap * sizeof(uint64_t)));
            }
            HIP_OK(hipMemcpy(s->dev_sup, pt.wraps.data(), n_sup * sizeof(uint64_t), hipMemcpyHostToDevice));
            d_sup = s->dev_sup;
        }
        HIP_OK(hipMemcpyAsync(s->dev, s->host, staged, hipMemcpyHostToDevice, s->stream));
        if (launch_count(c, s->stream, s->dev, staged, pt.carry, d_sup, n_sup)) return 1;
        HIP_OK(hipEventRecord(s->done, s->stream));
        s->busy = true;
    }
    return 0;
}

extern "C" int qm2_submit_device(qm2_ctx *c, const void *d_seq, size_t n, const uint64_t *d_sup,
                                 uint64_t n_sup)
{
    if (!c) return fail("qm2_submit_device: null context");
    if (n == 0) return 0;
    if (!d_seq) return fail("qm2_submit_device: null buffer");
    if (use_device(c)) return 1;
    return launch_count(c, c->stream, (const uint8_t *)d_seq, n, 0, d_sup, n_sup);
}

extern "C" int qm2_sync(qm2_ctx *c)
{
    if (!c) return fail("qm2_sync: null context");
    if (use_device(c)) return 1;
    for (auto &s : c->slots)
        if (s.busy) {
            HIP_OK(hipEventSynchronize(s.done));
            s.busy = false;
        }
    HIP_OK(hipStreamSynchronize(c->stream));
    if (harvest_timing(c)) return 1;
    DevStats st;
    HIP_OK(hipMemcpy(&st, c->stats, sizeof st, hipMemcpyDeviceToHost));
    if (st.overflow)
        return fail("qm2: %llu k-mers did not fit their exchange bucket", st.overflow);
    return 0;
}

extern "C" int qm2_finish(qm2_ctx *c, uint16_t *depth_out, qm2_stats *st)
{
    if (qm2_sync(c)) return 1;
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
    HIP_OK(hipMemset(c->counters, 0, n_slots * sizeof(uint32_t)));
    HIP_OK(hipMemset(c->stats, 0, sizeof(DevStats)));
    c->prof = {0, 0.0, 0.0};
    return 0;
}

extern "C" int qm2_get_profile(qm2_ctx *c, qm2_profile *p)
{
    if (!c || !p) return fail("qm2_get_profile: null argument");
    *p = c->prof;
    return 0;
}

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
    SeqView sv;
    const uintptr_t p = (uintptr_t)d_seq;
    sv.lead = p & 15u;
    sv.base = (const uint8_t *)(p - sv.lead);
    sv.emit_from = sv.lead;
    sv.n_total = sv.lead + n;
    sv.suppress = d_sup;
    sv.n_suppress = n_sup;
    const uint64_t n_tiles = (sv.n_total + kTileBytes - 1) / kTileBytes;
    uint32_t lgH = 0, lgB = 0;
    while ((1ull << lgH) < c->H) lgH++;
    while ((1u << lgB) < n_buckets) lgB++;
    const uint32_t owner_shift = lgH - lgB;
    const int grid = grid_for(c, n_tiles);
    TimedLaunch t;
    if (time_begin(c, c->stream, &t)) return 1;
    if (c->k == 30)
        hipLaunchKernelGGL(encode_kernel<true>, dim3(grid), dim3(kBlock), 0, c->stream, sv, c->k,
                           (uint32_t)(c->H - 1), n_tiles, d_out, bucket_cap, n_buckets, owner_shift,
                           (unsigned long long *)d_counts, c->stats);
    else
        hipLaunchKernelGGL(encode_kernel<false>, dim3(grid), dim3(kBlock), 0, c->stream, sv, c->k,
                           (uint32_t)(c->H - 1), n_tiles, d_out, bucket_cap, n_buckets, owner_shift,
                           (unsigned long long *)d_counts, c->stats);
    HIP_OK(hipGetLastError());
    return time_end(c, c->stream, &t);
}

extern "C" int qm2_probe_device(qm2_ctx *c, const uint64_t *d_kmers, uint64_t n, const uint64_t *d_n)
{
    if (!c) return fail("qm2_probe_device: null context");
    if (n == 0) return 0;
    if (!d_kmers) return fail("qm2_probe_device: null buffer");
    if (use_device(c)) return 1;
    const uint64_t groups = (n + (uint64_t)kWave * kNJ - 1) / ((uint64_t)kWave * kNJ);
    const int grid = grid_for(c, groups);
    TimedLaunch t;
    if (time_begin(c, c->stream, &t)) return 1;
    hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(kBlock), 0, c->stream, d_kmers, n,
                       (const unsigned long long *)d_n, table_view(c), c->stats);
    HIP_OK(hipGetLastError());
    return time_end(c, c->stream, &t);
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
        if (lo < (H >> 1)) {
            /* lower half: walks go up (:94-95) and may leave through `hi` */
            while (shi < H && keys[shi - 1] != 0) shi++;
            /* the walk stops ON the first empty slot: include it */
            if (shi < H && keys[shi - 1] != 0) shi++;
        } else {
            while (slo > 0 && keys[slo] != 0) slo--;
        }
    }
    if (own_lo) *own_lo = lo;
    if (own_hi) *own_hi = hi;
    if (span_lo) *span_lo = slo;
    if (span_hi) *span_hi = shi;
    return 0;
}
