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
 * drops (:398 comes before :451): the line after it is at a sequence position again,
 * i.e. every later sequence line comes one line later than the line number says.
 * Such lines are rare (malformed input), so they are found one at a time: a pass
 * that knows the first d of them (a sorted list of line numbers; line L then counts as
 * L minus the number of listed lines before it) reports the first sequence-position
 * '>' line that is not on the list; the host adds it and runs the pass again, until
 * none is found.  A batch with more than kRawMaxDesync of them is refused.
 *
 * The uint16 run counter of :402 wraps after 65535 bases without an 'N' or a line
 * end, and the k-1 windows behind the wrap are lost (:418).  The count kernels take
 * the wrap positions as a sorted list (as qm2_submit's host planner makes it,
 * qm2_plan.h); here the same list comes from a third scan channel — the position of
 * the last RESET ('N' or '\n') — so that every kept byte knows how far into its
 * N-free run it is (the few positions, appended in any order, are sorted on the host).
 * Only a batch with more wraps than the list holds is refused.
 *
 * Passes over 4 KiB tiles (256 threads x 16 bytes), 256 tiles to a group:
 *   raw_tile_kernel   per tile: number of '\n', position of the last one, of the last reset
 *   raw_group_kernel  the same per group of 256 tiles
 *   raw_scan_kernel   over the groups (a few thousand: one workgroup): exclusive prefix of the
 *                     counts and running maxima of the positions
 *   raw_apply_kernel  finishes the prefix inside its group (256 values, one per thread), which
 *                     gives the line number at the tile's first byte and where the line and
 *                     the N-free run in progress there started; then per byte: keep, or '\n'
 * All streaming: 1 B read + 1 B read + 1 B written per input byte.
 */
#pragma once
#include "qm2_device.hip.h"

namespace qm2 {

constexpr int      kRawThreads = 256;
constexpr uint32_t kRawTile = kRawThreads * 16;
constexpr int      kRawGroup = kRawThreads;  /* tiles per group: one per thread of raw_apply_kernel */
constexpr uint32_t kRawDesync = 1u;          /* flag bits: too many '>' lines at FASTQ sequence positions */
constexpr uint32_t kRawMaxDesync = 64;
constexpr uint32_t kRawTooManyWraps = 2u;
constexpr uint32_t kRawWrapCap = 1u << 20;   /* wrap positions per batch (one per 65536 bases of an N-free run: a 6 GiB batch
                                                of nothing but such runs has < 100 k) */

struct RawView {
    const uint8_t *in;      /* 16-byte aligned */
    uint8_t       *out;     /* 16-byte aligned, n + 1 bytes */
    uint64_t       n;
    uint32_t       fastq, skip;
    uint32_t       n_desync;                 /* known '>' lines at sequence positions, ascending line numbers */
    uint64_t       desync[kRawMaxDesync];
};

/* what the scans carry: newlines so far, position of the last newline, of the last reset ('N' or newline) */
struct RawAgg {
    uint32_t  cnt;
    long long nl, rst;
};
__device__ __forceinline__ RawAgg raw_join(RawAgg a, RawAgg b)
{
    RawAgg r;
    r.cnt = a.cnt + b.cnt;
    r.nl = a.nl > b.nl ? a.nl : b.nl;
    r.rst = a.rst > b.rst ? a.rst : b.rst;
    return r;
}
constexpr RawAgg kRawNone = {0u, -1, -1};

/* per-tile / per-group arrays of the three channels */
struct RawArrays {
    uint32_t  *cnt;
    long long *nl, *rst;
};
__device__ __forceinline__ RawAgg raw_get(const RawArrays &a, uint64_t i) { RawAgg r = {a.cnt[i], a.nl[i], a.rst[i]}; return r; }
__device__ __forceinline__ void raw_put(const RawArrays &a, uint64_t i, RawAgg v) { a.cnt[i] = v.cnt; a.nl[i] = v.nl; a.rst[i] = v.rst; }

/* 16-bit mask of the bytes of v that equal c (bit j = byte j) */
__device__ __forceinline__ uint32_t byte_mask16(uint4 v, uint32_t c)
{
    uint32_t m = 0;
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t z = zero_bytes(w[i] ^ (c * 0x01010101u));      /* 0x80 in every byte that is c */
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

/* what one thread's 16 bytes contribute */
__device__ __forceinline__ RawAgg raw_of16(uint4 v, uint64_t at)
{
    const uint32_t nl = byte_mask16(v, '\n'), rs = nl | byte_mask16(v, 'N');
    RawAgg a;
    a.cnt = __popc(nl);
    a.nl = nl ? (long long)at + (31 - __clz(nl)) : -1;
    a.rst = rs ? (long long)at + (31 - __clz(rs)) : -1;
    return a;
}

/* join over the workgroup (all threads get the result) */
__device__ __forceinline__ RawAgg raw_block_reduce(RawAgg v, RawAgg *s)
{
    for (int d = 1; d < kWave; d <<= 1) {
        RawAgg o;
        o.cnt = __shfl_xor(v.cnt, d);
        o.nl = __shfl_xor(v.nl, d);
        o.rst = __shfl_xor(v.rst, d);
        v = raw_join(v, o);
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();                                          /* s may still be read from a previous use */
    if (lane == 0) s[wave] = v;
    __syncthreads();
    RawAgg r = kRawNone;
    for (int w = 0; w < kRawThreads / kWave; w++) r = raw_join(r, s[w]);
    return r;
}

__global__ __launch_bounds__(kRawThreads) void raw_tile_kernel(RawView rv, RawArrays tile)
{
    __shared__ RawAgg s[kRawThreads / kWave];
    const uint64_t at = (uint64_t)blockIdx.x * kRawTile + threadIdx.x * 16;
    const RawAgg r = raw_block_reduce(raw_of16(load_raw16(rv, at), at), s);
    if (threadIdx.x == 0) raw_put(tile, blockIdx.x, r);
}

__global__ __launch_bounds__(kRawThreads) void raw_group_kernel(RawArrays tile, uint64_t n_tiles, RawArrays group)
{
    __shared__ RawAgg s[kRawThreads / kWave];
    const uint64_t t = (uint64_t)blockIdx.x * kRawGroup + threadIdx.x;
    const RawAgg r = raw_block_reduce(t < n_tiles ? raw_get(tile, t) : kRawNone, s);
    if (threadIdx.x == 0) raw_put(group, blockIdx.x, r);
}

/* one workgroup over the GROUPS, in place: group[g] becomes the join of the groups in front of g */
__global__ __launch_bounds__(1024) void raw_scan_kernel(RawArrays group, uint64_t n_groups)
{
    __shared__ RawAgg s[1024];
    const uint64_t per = (n_groups + 1023) / 1024;
    const uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < n_groups ? lo + per : n_groups;
    RawAgg acc = kRawNone;
    for (uint64_t g = lo; g < hi; g++) acc = raw_join(acc, raw_get(group, g));
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {                                   /* 1024 partials: serial is microseconds */
        RawAgg run = kRawNone;
        for (int i = 0; i < 1024; i++) {
            const RawAgg mine = s[i];
            s[i] = run;
            run = raw_join(run, mine);
        }
    }
    __syncthreads();
    acc = s[threadIdx.x];
    for (uint64_t g = lo; g < hi; g++) {
        const RawAgg mine = raw_get(group, g);
        raw_put(group, g, acc);
        acc = raw_join(acc, mine);
    }
}

__global__ __launch_bounds__(kRawThreads) void raw_apply_kernel(RawView rv, RawArrays tile, RawArrays group_before,
                                                                unsigned int *flags, uint64_t *wraps, unsigned int *n_wraps,
                                                                unsigned long long *next_desync)
{
    __shared__ RawAgg s[kRawThreads / kWave];
    /* the tiles of this group in front of this one: thread t looks at tile t of the group */
    const uint32_t g = blockIdx.x / kRawGroup, j_tile = blockIdx.x % kRawGroup;
    RawAgg before = raw_block_reduce((uint32_t)threadIdx.x < j_tile ? raw_get(tile, (uint64_t)g * kRawGroup + threadIdx.x) : kRawNone, s);
    before = raw_join(raw_get(group_before, g), before);
    __syncthreads();
    const uint64_t at = (uint64_t)blockIdx.x * kRawTile + threadIdx.x * 16;
    const uint4 v = load_raw16(rv, at);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    /* inclusive scan over the wave, then made exclusive */
    const RawAgg mine = raw_of16(v, at);
    RawAgg inc = mine;
    for (int d = 1; d < kWave; d <<= 1) {
        RawAgg o;
        o.cnt = __shfl_up(inc.cnt, d);
        o.nl = __shfl_up(inc.nl, d);
        o.rst = __shfl_up(inc.rst, d);
        if (lane >= d) inc = raw_join(o, inc);
    }
    if (lane == kWave - 1) s[wave] = inc;
    __syncthreads();
    for (int w = 0; w < wave; w++) before = raw_join(before, s[w]);
    {
        RawAgg e;
        e.cnt = __shfl_up(inc.cnt, 1);
        e.nl = __shfl_up(inc.nl, 1);
        e.rst = __shfl_up(inc.rst, 1);
        if (lane > 0) before = raw_join(before, e);
    }
    if (at >= rv.n) return;
    /* the line in progress at this thread's first byte started at `start`; its first byte decides FASTA
     * lines; the N-free run in progress started at `run` */
    uint64_t line = before.cnt;
    uint64_t start = (uint64_t)(before.nl + 1), run = (uint64_t)(before.rst + 1);
    const uint8_t *b = (const uint8_t *)&v;
    const uint8_t first = start == at ? b[0] : rv.in[start];
    auto keep_line = [&](uint64_t l, uint8_t fc) {
        if (rv.fastq) {
            uint64_t shift = 0;                               /* known desync lines in front of l; l itself on the list: dropped */
            for (uint32_t d = 0; d < rv.n_desync; d++) {
                if (rv.desync[d] == l) return false;
                shift += rv.desync[d] < l ? 1u : 0u;
            }
            const bool seq = l >= rv.skip + shift && ((l - rv.skip - shift) & 3u) == 0;
            if (seq && fc == '>') {                           /* a new one: the host takes the first and runs again */
                atomicMin(next_desync, (unsigned long long)l);
                return false;
            }
            return seq;
        }
        return fc != '>';
    };
    bool keep = keep_line(line, first);
    /* a wrap needs 65536 bases since the last reset: with a run that began less than that before
     * this thread's last byte (nearly always) there is nothing to track */
    const bool may_wrap = at + 16 - run >= 65536;
    uint8_t o[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t p = at + j;
        if (p >= rv.n) { o[j] = '\n'; continue; }
        const uint8_t ch = b[j];
        o[j] = keep ? ch : (uint8_t)'\n';
        if (ch == '\n') {
            line++;
            start = run = p + 1;
            if (j + 1 < 16 && p + 1 < rv.n) keep = keep_line(line, b[j + 1]);
        } else if (may_wrap) {
            if (ch == 'N') run = p + 1;                      /* :404-408 */
            else if (keep && ((p - run + 1) & 0xFFFFu) == 0) {
                /* base number 65536 m of its run: cur_chars wraps here (:402, :410) */
                const unsigned int w = atomicAdd(n_wraps, 1u);
                if (w < kRawWrapCap) wraps[w] = p;
                else atomicOr(flags, kRawTooManyWraps);
            }
        }
    }
    *(uint4 *)(rv.out + at) = *(const uint4 *)o;
    if (at + 16 >= rv.n) rv.out[rv.n] = '\n';                /* terminator for an unterminated last line */
}

}  // namespace qm2
