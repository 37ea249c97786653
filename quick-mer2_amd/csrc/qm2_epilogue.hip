/*
 * qm2_epilogue.hip — the chain walk of the epilogue (QuicKmer.c:494-516) as a
 * parallel list ranking on the device (include/qm2.h: qm2_chain_gather /
 * qm2_chain_fetch).
 *
 * The reference does   i = first; do { out[n++] = depth[i]; i = next[i]; } while (i != first);
 * — one dependent cache miss per dictionary k-mer (~0.45 us each, ~15 min for
 * hg38).  Here:
 *   1. splitters = occupied slots whose index is a multiple of R, plus first_idx
 *      (decided from the key table before the chain overwrites it);
 *   2. walk_len_kernel: one thread per splitter walks next[] to the following
 *      splitter, records the sub-list length and the id of that splitter;
 *   3. the short list of splitters (H/R entries) is ranked on the host: an
 *      exclusive prefix of the lengths in chain order;
 *   4. walk_gather_kernel: one thread per splitter walks again and writes
 *      (uint16) counters[slot] at base + j.
 * Work: two passes of independent pointer chases, H/R of them in flight.
 */
#include <algorithm>
#include <cstdio>
#include <vector>

#include "qm2_ctx.h"

using namespace qm2;

namespace {

constexpr uint32_t kNoSplitter = 0xFFFFFFFFu;
constexpr uint32_t kMaxHops = 1u << 26;      /* a sub-list this long means the chain is corrupt */

struct ChainView {
    const uint32_t *next;
    uint32_t lg_r;          /* splitter stride R = 2^lg_r */
    uint32_t first;         /* first_idx (QuicKmer.c:494) */
    uint32_t extra_id;      /* id of first_idx when it is not a multiple of R */
};

__device__ __forceinline__ bool is_splitter(const ChainView &cv, uint32_t slot)
{
    return (slot & ((1u << cv.lg_r) - 1u)) == 0 || slot == cv.first;
}
__device__ __forceinline__ uint32_t splitter_id(const ChainView &cv, uint32_t slot)
{
    return slot == cv.first && (slot & ((1u << cv.lg_r) - 1u)) != 0 ? cv.extra_id : slot >> cv.lg_r;
}

/* valid[id] = 1 for candidate slots that hold a key (are on the chain) */
__global__ void mark_splitters_kernel(const uint64_t *keys, uint64_t H, ChainView cv, uint8_t *valid)
{
    const uint64_t n = (H >> cv.lg_r) + 1;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = id == cv.extra_id ? cv.first : id << cv.lg_r;
        bool ok = slot < H && keys[slot] != 0;
        if (id == cv.extra_id && (cv.first & ((1u << cv.lg_r) - 1u)) == 0) ok = false;   /* first_idx has a regular id */
        valid[id] = ok ? 1 : 0;
    }
}

__global__ void walk_len_kernel(ChainView cv, const uint8_t *valid, uint64_t n_ids, uint32_t *len, uint32_t *succ,
                                unsigned long long *bad)
{
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_ids; id += (uint64_t)gridDim.x * blockDim.x) {
        if (!valid[id]) { len[id] = 0; succ[id] = kNoSplitter; continue; }
        const uint32_t start = id == cv.extra_id ? cv.first : (uint32_t)(id << cv.lg_r);
        uint32_t node = cv.next[start], hops = 1;
        while (!is_splitter(cv, node) && hops < kMaxHops) { node = cv.next[node]; hops++; }
        if (hops >= kMaxHops) { atomicAdd(bad, 1ull); succ[id] = kNoSplitter; len[id] = 0; continue; }
        len[id] = hops;
        succ[id] = splitter_id(cv, node);
    }
}

__global__ void walk_gather_kernel(ChainView cv, const uint64_t *base, uint64_t n_ids, const uint32_t *len,
                                   const uint32_t *counters, uint16_t *out)
{
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_ids; id += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t at = base[id];
        if (at == ~0ull) continue;                               /* not on the cycle through first_idx */
        uint32_t node = id == cv.extra_id ? cv.first : (uint32_t)(id << cv.lg_r);
        const uint32_t n = len[id];
        for (uint32_t j = 0; j < n; j++) {
            out[at + j] = (uint16_t)counters[node];              /* QuicKmer.c:499 */
            node = cv.next[node];                                /* QuicKmer.c:500 */
        }
    }
}

}  // namespace

/* the chain comes either from host memory (`chain`) or from a file (`fd` at `file_offset`) */
static int chain_gather(qm2_ctx *c, const uint32_t *chain, int fd, uint64_t file_offset, uint64_t first_idx,
                        uint64_t *n_out)
{
    if (!c || (!chain && fd < 0)) return qm2_fail("qm2_chain_gather: null argument");
    if (c->keys_pending) return qm2_fail("qm2_chain_gather: the key table has not been loaded yet");
    if (c->span_lo != 0 || c->span_hi != c->H) return qm2_fail("qm2_chain_gather: needs the whole table");
    if (c->keys_consumed) return qm2_fail("qm2_chain_gather: already done for this context");
    if (qm2_sync(c)) return 1;
    HIP_OK(hipSetDevice(c->device));
    const uint64_t H = c->H;
    hipStream_t s = c->stream;
    if (qm2_fold_counters(c)) return 1;                          /* (the walk reads counters[]: whole before it starts; same stream) */

    ChainView cv;
    /* ~2^20 splitters: enough parallel walkers to hide the pointer-chase latency, few
     * enough that ranking them on the host is milliseconds */
    uint32_t lg_r = 6;
    while (lg_r < 16 && (H >> lg_r) > (1ull << 21)) lg_r++;
    if ((1ull << lg_r) > H) lg_r = 0;
    cv.lg_r = lg_r;
    cv.first = (uint32_t)first_idx;                              /* uint32 in the reference too, :494 */
    if (cv.first >= H) return qm2_fail("qm2_chain_gather: first index outside the table");
    const uint64_t n_ids = (H >> lg_r) + 1;
    cv.extra_id = (uint32_t)(H >> lg_r);

    uint8_t *d_valid = nullptr;
    uint32_t *d_len = nullptr, *d_succ = nullptr;
    uint64_t *d_base = nullptr;
    unsigned long long *d_bad = nullptr;
    HIP_OK(hipMalloc((void **)&d_valid, n_ids));
    HIP_OK(hipMalloc((void **)&d_len, n_ids * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&d_succ, n_ids * sizeof(uint32_t)));
    HIP_OK(hipMalloc((void **)&d_base, n_ids * sizeof(uint64_t)));
    HIP_OK(hipMalloc((void **)&d_bad, sizeof(unsigned long long)));
    HIP_OK(hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), s));

    /* 1. splitters, while the keys are still there */
    cv.next = nullptr;
    const int grid = (int)std::min<uint64_t>((n_ids + 255) / 256, 4096);
    hipLaunchKernelGGL(mark_splitters_kernel, dim3(grid), dim3(256), 0, s, c->keys, H, cv, d_valid);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(s));

    /* 2. the chain goes where the keys were (QuicKmer.c:483, :490) when we own that buffer */
    qm2_part_free(c->part);                                      /* partition buffers are not needed any more */
    if (c->own_keys) {
        c->chain = (uint32_t *)c->keys;
        c->keys_consumed = true;
    } else if (!c->chain) {
        HIP_OK(hipMalloc((void **)&c->chain, H * sizeof(uint32_t)));
        c->own_chain = true;
    }
    if (chain) HIP_OK(hipMemcpy(c->chain, chain, H * sizeof(uint32_t), hipMemcpyHostToDevice));
    else if (qm2_stream_fd(c, fd, file_offset, c->chain, H * sizeof(uint32_t), "Dictionary chain")) {
        (void)hipFree(d_valid); (void)hipFree(d_len); (void)hipFree(d_succ); (void)hipFree(d_base); (void)hipFree(d_bad);
        return 1;
    }
    cv.next = c->chain;

    /* 3. sub-list lengths */
    hipLaunchKernelGGL(walk_len_kernel, dim3(grid), dim3(256), 0, s, cv, d_valid, n_ids, d_len, d_succ, d_bad);
    HIP_OK(hipGetLastError());
    std::vector<uint32_t> len(n_ids), succ(n_ids);
    unsigned long long bad = 0;
    HIP_OK(hipMemcpyAsync(len.data(), d_len, n_ids * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(succ.data(), d_succ, n_ids * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));

    /* 4. rank the splitters on the host, in chain order from first_idx */
    int rc = 0;
    uint64_t total = 0;
    std::vector<uint64_t> base(n_ids, ~0ull);
    if (bad) rc = qm2_fail("qm2_chain_gather: the chain does not return to a splitter (corrupt .qm?)");
    else {
        const uint32_t first_id = (cv.first & ((1u << lg_r) - 1u)) == 0 ? (uint32_t)(cv.first >> lg_r) : cv.extra_id;
        uint32_t id = first_id;
        uint64_t visited = 0;
        do {
            if (id == kNoSplitter || id >= n_ids || len[id] == 0 || base[id] != ~0ull || ++visited > n_ids) {
                rc = qm2_fail("qm2_chain_gather: the chain is not one cycle through the first index (corrupt .qm?)");
                break;
            }
            base[id] = total;
            total += len[id];
            id = succ[id];
        } while (id != first_id);
    }
    if (!rc) {
        /* 5. gather */
        if (c->gathered) { HIP_OK(hipFree(c->gathered)); c->gathered = nullptr; }
        HIP_OK(hipMalloc((void **)&c->gathered, std::max<uint64_t>(total, 1) * sizeof(uint16_t)));
        HIP_OK(hipMemcpyAsync(d_base, base.data(), n_ids * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(walk_gather_kernel, dim3(grid), dim3(256), 0, s, cv, d_base, n_ids, d_len, c->counters,
                           c->gathered);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(s));
        c->n_gathered = total;
        if (n_out) *n_out = total;
    }
    (void)hipFree(d_valid); (void)hipFree(d_len); (void)hipFree(d_succ); (void)hipFree(d_base); (void)hipFree(d_bad);
    return rc;
}

extern "C" int qm2_chain_gather(qm2_ctx *c, const uint32_t *chain, uint64_t first_idx, uint64_t *n_out)
{
    if (!chain) return qm2_fail("qm2_chain_gather: null argument");
    return chain_gather(c, chain, -1, 0, first_idx, n_out);
}

extern "C" int qm2_chain_gather_fd(qm2_ctx *c, int fd, uint64_t file_offset, uint64_t first_idx, uint64_t *n_out)
{
    if (fd < 0) return qm2_fail("qm2_chain_gather_fd: bad file descriptor");
    return chain_gather(c, nullptr, fd, file_offset, first_idx, n_out);
}

namespace {

constexpr uint32_t kGcBlock = 1u << 20;      /* entries the reference freads from the .qgc at a time (:11, :502) */

/* QuicKmer.c:504-509 over entries [0, n): workgroup-private sums in LDS (the control k-mers of a genome
 * crowd into a hundred GC bins: 64-bit LDS atomics take that at a few cycles each), then one global add
 * per bin and workgroup.  acc = sum[401] | sumsq[401] | cnt[401]. */
__global__ __launch_bounds__(256) void gc_sums_kernel(const uint16_t *depth, const uint16_t *gc, uint64_t n, uint64_t n_file,
                                                      const uint16_t *stale, unsigned long long *acc)
{
    __shared__ unsigned long long s_sum[QM2_GC_BINS], s_sq[QM2_GC_BINS];
    __shared__ unsigned int s_cnt[QM2_GC_BINS];          /* (a workgroup sees n / gridDim.x entries: < 2^32) */
    for (int b = threadIdx.x; b < QM2_GC_BINS; b += blockDim.x) { s_sum[b] = 0; s_sq[b] = 0; s_cnt[b] = 0; }
    __syncthreads();
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = e < n_file ? gc[e] : stale[e & (kGcBlock - 1u)];
        if (g & 0x8000u) {                                                   /* :504 */
            const uint32_t b = g & 0x1FFu;
            if (b < QM2_GC_BINS) {
                const uint32_t v = depth[e];
                atomicAdd(&s_sum[b], (unsigned long long)v);                 /* :505 */
                atomicAdd(&s_cnt[b], 1u);                                    /* :506 */
                atomicAdd(&s_sq[b], (unsigned long long)(long long)(int32_t)(v * v));   /* :507: an `int` product, wrapped */
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < QM2_GC_BINS; b += blockDim.x) {
        if (s_cnt[b] == 0) continue;
        atomicAdd(&acc[b], s_sum[b]);
        atomicAdd(&acc[QM2_GC_BINS + b], s_sq[b]);
        atomicAdd(&acc[2 * QM2_GC_BINS + b], (unsigned long long)s_cnt[b]);
    }
}

}  // namespace

extern "C" int qm2_chain_gc_sums_fd(qm2_ctx *c, int fd, uint64_t n_file, const uint16_t *stale,
                                    int64_t *sum, int64_t *sumsq, uint64_t *cnt)
{
    if (!c || !sum || !sumsq || !cnt) return qm2_fail("qm2_chain_gc_sums_fd: null argument");
    if (!c->gathered) return qm2_fail("qm2_chain_gc_sums_fd: call qm2_chain_gather first");
    if (fd < 0) return qm2_fail("qm2_chain_gc_sums_fd: bad file descriptor");
    const uint64_t n = c->n_gathered;
    if (n_file < n && !stale) return qm2_fail("qm2_chain_gc_sums_fd: a short GC control file needs the stale block");
    HIP_OK(hipSetDevice(c->device));
    const uint64_t n_up = std::min(n_file, n);
    uint16_t *d_gc = nullptr, *d_stale = nullptr;
    unsigned long long *d_acc = nullptr;
    int rc = 0;
    auto release = [&]() { (void)hipFree(d_gc); (void)hipFree(d_stale); (void)hipFree(d_acc); };
    if (hipMalloc((void **)&d_gc, std::max<uint64_t>(n_up, 1) * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc((void **)&d_stale, (size_t)kGcBlock * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc((void **)&d_acc, 3 * QM2_GC_BINS * sizeof(unsigned long long)) != hipSuccess) {
        release();
        return qm2_fail("Memory allocation failed");
    }
    if (n_up && qm2_stream_fd(c, fd, 0, d_gc, n_up * sizeof(uint16_t), "GC control file")) { release(); return 1; }
    hipStream_t s = c->stream;
    std::vector<unsigned long long> acc(3 * QM2_GC_BINS);
    if (hipMemsetAsync(d_acc, 0, acc.size() * sizeof(unsigned long long), s) != hipSuccess ||
        (stale ? hipMemcpyAsync(d_stale, stale, (size_t)kGcBlock * sizeof(uint16_t), hipMemcpyHostToDevice, s)
               : hipMemsetAsync(d_stale, 0, (size_t)kGcBlock * sizeof(uint16_t), s)) != hipSuccess)
        rc = qm2_fail("qm2_chain_gc_sums_fd: upload failed");
    if (!rc && n) {
        const int grid = (int)std::min<uint64_t>((n + 255) / 256, (uint64_t)c->n_cu * 8);
        hipLaunchKernelGGL(gc_sums_kernel, dim3(grid), dim3(256), 0, s, c->gathered, d_gc, n, n_up, d_stale, d_acc);
        if (hipGetLastError() != hipSuccess) rc = qm2_fail("qm2_chain_gc_sums_fd: launch failed");
    }
    if (!rc && (hipMemcpyAsync(acc.data(), d_acc, acc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess))
        rc = qm2_fail("qm2_chain_gc_sums_fd: the sums did not come back (%s)", hipGetErrorString(hipGetLastError()));
    release();
    if (rc) return rc;
    for (int b = 0; b < QM2_GC_BINS; b++) {
        sum[b] = (int64_t)acc[b];
        sumsq[b] = (int64_t)acc[QM2_GC_BINS + b];
        cnt[b] = (uint64_t)acc[2 * QM2_GC_BINS + b];
    }
    return 0;
}

extern "C" int qm2_chain_fetch(qm2_ctx *c, uint16_t *out, uint64_t offset, uint64_t count)
{
    if (!c || !out) return qm2_fail("qm2_chain_fetch: null argument");
    if (!c->gathered) return qm2_fail("qm2_chain_fetch: call qm2_chain_gather first");
    if (offset > c->n_gathered || count > c->n_gathered - offset)
        return qm2_fail("qm2_chain_fetch: window outside the gathered counts");
    if (count == 0) return 0;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpy(out, c->gathered + offset, count * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return 0;
}
