/*
 * qm2_raw.hip.h — the input line discipline on the device:
 * raw FASTA / FASTQ text in, the same text out with every byte of a line the
 * reference does not count replaced by '\n' — so the count kernels, for which
 * '\n' only ends a line, see exactly the sequence lines (plus empty ones).
 *
 * The reference's rule (QuicKmer.c:397-398, :451-455): each line the loop's own
 * fgets fetches is a sequence line unless it starts with '>'; in FASTQ mode the
 * three lines after a sequence line are fetched and dropped unseen.  So, with
 * `skip` = lines to drop before the first sequence line of the block,
 *   FASTQ  line L is a sequence line  iff  L >= skip and (L - skip) % 4 == 0
 *   FASTA  line L is a sequence line  iff  its first byte is not '>'.
 * A FASTQ sequence-position line that starts with '>' is skipped WITHOUT the three
 * drops (:398 comes before :451): from there on the phase is no longer a function
 * of the line number.  That, and a line longer than 65535 bytes (the uint16 run
 * counter of :402 would wrap and this path carries no wrap list), are reported in
 * a flag word instead of being counted wrongly; the caller takes the host path for
 * such a block.
 *
 * Passes over 4 KiB tiles (256 threads x 16 bytes), 256 tiles to a group:
 *   raw_tile_kernel   per tile: number of '\n', position of the last one
 *   raw_group_kernel  the same per group of 256 tiles
 *   raw_scan_kernel   over the groups (a few thousand: one workgroup): exclusive prefix of the
 *                     counts and running maximum of the positions
 *   raw_apply_kernel  finishes the prefix inside its group (256 values, one per thread), which
 *                     gives the line number at the tile's first byte and where the line in
 *                     progress there started; then per byte: keep, or '\n'
 * All streaming: 1 B read + 1 B read + 1 B written per input byte.
 */
#pragma once
#include "qm2_device.hip.h"

namespace qm2 {

constexpr int      kRawThreads = 256;
constexpr uint32_t kRawTile = kRawThreads * 16;
constexpr uint32_t kRawDesync = 1u;          /* flag bits */
constexpr uint32_t kRawLongLine = 2u;
constexpr uint32_t kRawMaxLine = 65535;

struct RawView {
    const uint8_t *in;      /* 16-byte aligned */
    uint8_t       *out;     /* 16-byte aligned, n + 1 bytes */
    uint64_t       n;
    uint32_t       fastq, skip;
};

/* 16-bit mask of the '\n' bytes of v (bit j = byte j) */
__device__ __forceinline__ uint32_t newline_mask16(uint4 v)
{
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t z = zero_bytes(w[i] ^ 0x0A0A0A0Au);      /* 0x80 in every byte that is '\n' */
        m |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * i);
    }
    return m;
}

__device__ __forceinline__ uint4 load_raw16(const RawView &rv, uint64_t at)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at + 16 <= rv.n) v = *(const uint4 *)(rv.in + at);
    else if (at < rv.n) {
        uint8_t b[16] = {0};
        for (uint32_t j = 0; at + j < rv.n; j++) b[j] = rv.in[at + j];
        v = *(const uint4 *)b;
    }
    return v;
}

__global__ __launch_bounds__(kRawThreads) void raw_tile_kernel(RawView rv, uint32_t *tile_nl, long long *tile_last)
{
    __shared__ uint32_t s_cnt[kRawThreads / kWave];
    __shared__ long long s_last[kRawThreads / kWave];
    const uint64_t at = (uint64_t)blockIdx.x * kRawTile + threadIdx.x * 16;
    const uint32_t m = newline_mask16(load_raw16(rv, at));
    uint32_t cnt = __popc(m);
    long long last = m ? (long long)at + (31 - __clz(m)) : -1;
    for (int d = 1; d < kWave; d <<= 1) {
        cnt += __shfl_xor(cnt, d);
        const long long o = __shfl_xor(last, d);
        last = o > last ? o : last;
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) { s_cnt[wave] = cnt; s_last[wave] = last; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        long long l = -1;
        for (int w = 0; w < kRawThreads / kWave; w++) { c += s_cnt[w]; l = s_last[w] > l ? s_last[w] : l; }
        tile_nl[blockIdx.x] = c;
        tile_last[blockIdx.x] = l;
    }
}

constexpr int kRawGroup = kRawThreads;      /* tiles per group: one per thread of raw_apply_kernel */

/* sum / max of a value pair over the workgroup (all threads get the result) */
__device__ __forceinline__ void raw_block_reduce(uint32_t &cnt, long long &last, uint32_t *s_cnt, long long *s_last)
{
    for (int d = 1; d < kWave; d <<= 1) {
        cnt += __shfl_xor(cnt, d);
        const long long o = __shfl_xor(last, d);
        last = o > last ? o : last;
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();                                          /* s_* may still be read from a previous use */
    if (lane == 0) { s_cnt[wave] = cnt; s_last[wave] = last; }
    __syncthreads();
    cnt = 0;
    last = -1;
    for (int w = 0; w < kRawThreads / kWave; w++) { cnt += s_cnt[w]; last = s_last[w] > last ? s_last[w] : last; }
}

__global__ __launch_bounds__(kRawThreads) void raw_group_kernel(const uint32_t *tile_nl, const long long *tile_last, uint64_t n_tiles,
                                                                uint32_t *group_nl, long long *group_last)
{
    __shared__ uint32_t s_cnt[kRawThreads / kWave];
    __shared__ long long s_last[kRawThreads / kWave];
    const uint64_t t = (uint64_t)blockIdx.x * kRawGroup + threadIdx.x;
    uint32_t cnt = t < n_tiles ? tile_nl[t] : 0;
    long long last = t < n_tiles ? tile_last[t] : -1;
    raw_block_reduce(cnt, last, s_cnt, s_last);
    if (threadIdx.x == 0) { group_nl[blockIdx.x] = cnt; group_last[blockIdx.x] = last; }
}

/* one workgroup over the GROUPS: before[g] = newlines in groups < g; prev[g] = last newline position in groups < g (-1: none) */
__global__ __launch_bounds__(1024) void raw_scan_kernel(const uint32_t *tile_nl, const long long *tile_last, uint64_t n_tiles,
                                                        uint64_t *tile_before, long long *tile_prev)
{
    __shared__ uint64_t s_sum[1024];
    __shared__ long long s_max[1024];
    const uint64_t per = (n_tiles + 1023) / 1024;
    const uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    uint64_t sum = 0;
    long long mx = -1;
    for (uint64_t t = lo; t < hi; t++) { sum += tile_nl[t]; mx = tile_last[t] > mx ? tile_last[t] : mx; }
    s_sum[threadIdx.x] = sum;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {                                   /* 1024 partials: serial is microseconds */
        uint64_t run = 0;
        long long rmx = -1;
        for (int i = 0; i < 1024; i++) {
            const uint64_t s = s_sum[i];
            const long long m = s_max[i];
            s_sum[i] = run;
            s_max[i] = rmx;
            run += s;
            rmx = m > rmx ? m : rmx;
        }
    }
    __syncthreads();
    sum = s_sum[threadIdx.x];
    mx = s_max[threadIdx.x];
    for (uint64_t t = lo; t < hi; t++) {
        tile_before[t] = sum;
        tile_prev[t] = mx;
        sum += tile_nl[t];
        mx = tile_last[t] > mx ? tile_last[t] : mx;
    }
}

__global__ __launch_bounds__(kRawThreads) void raw_apply_kernel(RawView rv, const uint32_t *tile_nl, const long long *tile_last,
                                                                const uint64_t *group_before, const long long *group_prev,
                                                                unsigned int *flags)
{
    __shared__ uint32_t s_cnt[kRawThreads / kWave];
    __shared__ long long s_last[kRawThreads / kWave];
    /* the tiles of this group in front of this one: thread t looks at tile t of the group */
    uint64_t tile_line;
    long long tile_prev_nl;
    {
        const uint32_t g = blockIdx.x / kRawGroup, j = blockIdx.x % kRawGroup;
        const uint64_t t = (uint64_t)g * kRawGroup + threadIdx.x;
        uint32_t c = threadIdx.x < j ? tile_nl[t] : 0;
        long long l = threadIdx.x < j ? tile_last[t] : -1;
        raw_block_reduce(c, l, s_cnt, s_last);
        tile_line = group_before[g] + c;
        tile_prev_nl = group_prev[g] > l ? group_prev[g] : l;
        __syncthreads();
    }
    const uint64_t at = (uint64_t)blockIdx.x * kRawTile + threadIdx.x * 16;
    const uint4 v = load_raw16(rv, at);
    const uint32_t m = newline_mask16(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    /* inclusive scans over the wave, then made exclusive */
    uint32_t cnt = __popc(m);
    long long last = m ? (long long)at + (31 - __clz(m)) : -1;
    uint32_t icnt = cnt;
    long long ilast = last;
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t c = __shfl_up(icnt, d);
        const long long l = __shfl_up(ilast, d);
        if (lane >= d) { icnt += c; ilast = l > ilast ? l : ilast; }
    }
    if (lane == kWave - 1) { s_cnt[wave] = icnt; s_last[wave] = ilast; }
    __syncthreads();
    uint64_t line = tile_line;
    long long prev = tile_prev_nl;
    for (int w = 0; w < wave; w++) { line += s_cnt[w]; prev = s_last[w] > prev ? s_last[w] : prev; }
    {
        const uint32_t ec = __shfl_up(icnt, 1);
        const long long el = __shfl_up(ilast, 1);
        if (lane > 0) { line += ec; prev = el > prev ? el : prev; }
    }
    if (at >= rv.n) return;
    /* the line in progress at this thread's first byte started at `start`; its first byte decides FASTA lines */
    uint64_t start = (uint64_t)(prev + 1);
    const uint8_t *b = (const uint8_t *)&v;
    uint8_t first = start == at ? b[0] : rv.in[start];
    auto keep_line = [&](uint64_t l, uint8_t fc) {
        if (rv.fastq) {
            const bool seq = l >= rv.skip && ((l - rv.skip) & 3u) == 0;
            if (seq && fc == '>') atomicOr(flags, kRawDesync);
            return seq;
        }
        return fc != '>';
    };
    bool keep = keep_line(line, first);
    uint8_t o[16];
    unsigned int fl = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t p = at + j;
        if (p >= rv.n) { o[j] = '\n'; continue; }
        const uint8_t ch = b[j];
        o[j] = keep ? ch : (uint8_t)'\n';
        if (ch == '\n') {
            if (p - start > kRawMaxLine) fl |= kRawLongLine;
            line++;
            start = p + 1;
            if (j + 1 < 16 && p + 1 < rv.n) keep = keep_line(line, b[j + 1]);
        } else if (p + 1 == rv.n) {                          /* last line of the buffer, no newline */
            if (p + 1 - start > kRawMaxLine) fl |= kRawLongLine;
        }
    }
    if (fl) atomicOr(flags, fl);
    *(uint4 *)(rv.out + at) = *(const uint4 *)o;
    if (at + 16 >= rv.n && at < rv.n) rv.out[rv.n] = '\n';    /* terminator for an unterminated last line */
}

}  // namespace qm2
