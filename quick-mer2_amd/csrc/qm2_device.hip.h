/*
 * qm2_device.hip.h — device-side building blocks of the MI355X `count` path.
 * gfx950 only (wave64).  No reference code: the behaviour follows SURVEY.md
 * Appendix B; reference lines are cited where a rule comes from.
 *
 * Data flow of one wave-step (one "tile" = 992 sequence bytes):
 *
 *   64 lanes x 16 B coalesced load  (1 KiB per wave instruction; lanes 0-1 are
 *        |                           the 32-byte look-back halo of the tile)
 *   SWAR 2-bit pack: 16 bases -> one u32 (MSB-first) + a 16-bit reset mask
 *        |                          ('N' and '\n' are the only resets, :403-408)
 *   wave shuffles: each lane also gets the packed words of lanes -1 and -2
 *        |                          => a 48-base window in three registers
 *   16 x { funnel-shift out the forward k-mer and the reverse complement of
 *          the last 30 bases (:412-416), min (:420), DJB (:66-76) -> home slot }
 *        |
 *   probe rounds: all first probes of the 16 k-mers are issued back to back
 *          (16 independent 8-byte gathers in flight per lane), later rounds
 *          only for the unresolved ones (Find_hash, :90-99)
 *        |
 *   no-return global atomic add on the 32-bit counter of each hit (:291)
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qm2 {

constexpr int      kWave          = 64;
constexpr int      kBlock         = 256;                     /* 4 waves per workgroup */
constexpr int      kBytesPerLane  = 16;
constexpr int      kHaloLanes     = 2;                       /* 32 B look-back >= 30 */
constexpr int      kTileBytes     = (kWave - kHaloLanes) * kBytesPerLane; /* 992 */
constexpr uint64_t kMask60        = (1ull << 60) - 1;

struct TableView {
    const uint64_t *keys;      /* slots [span_lo, span_hi) of the table          */
    uint32_t       *counters;  /* same span, 32-bit accumulators                 */
    uint16_t       *interior16;/* same span: the private bucket interiors of bucket_count (qm2_ctx.h); else unused */
    const uint32_t *img_r;     /* same span: the table image bucket_count stages — every slot as a bucket record, */
    const void     *img_w;     /* ... R and second word (2 or 4 bytes; qm2_partition.hip.h); else unused          */
    uint64_t        span_lo;   /* global slot index of keys[0]                   */
    uint32_t        hmask;     /* H - 1 (H <= 2^32)                              */
    uint32_t        hhalf;     /* H >> 1                                         */
};

struct SeqView {
    const uint8_t  *base;      /* 16-byte aligned pointer at or before the data  */
    uint64_t        lead;      /* bytes [0, lead) are before the buffer: '\n'    */
    uint64_t        emit_from; /* windows ending before this byte emit nothing   */
    uint64_t        n_total;   /* lead + payload bytes                           */
    const uint64_t *suppress;  /* sorted byte positions (relative to payload byte
                                  0 = base + lead) whose windows the uint16 run
                                  counter drops (:402)                           */
    uint64_t        n_suppress;
    uint32_t        packed = 0;/* the input is PACKED (include/qm2.h, "packed sequence"): base = its blocks (16-byte aligned),
                                  lead = emit_from = 0, n_total = 64 x blocks positions; the kernels' PK instantiations read it */
};

/* stats block in device memory */
struct DevStats {
    unsigned long long n_bases, n_kmers, n_probes, n_hits;
    unsigned long long overflow;    /* qm2_encode_device: k-mers that did not fit a bucket (error) */
    unsigned long long part_lost;   /* partitioned pipeline, current batch: k-mers that fit neither
                                       their bucket nor the spill list => the batch is recounted by
                                       the direct kernel instead (exact either way)              */
    unsigned long long fallbacks;   /* batches that took that fallback                            */
    unsigned long long spilled;     /* k-mers that went through the spill list, all batches       */
};

/* ---- SWAR helpers ---------------------------------------------------------- */

/* 0x80 in every byte of x that is zero (exact, no cross-byte borrow). */
__host__ __device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

/* Four ASCII bytes (little-endian in w) -> 8 bits, first byte in the top pair.
 * code = (c >> 1) & 3  (:411): A,a->0  C,c->1  T,t->2  G,g,n->3, any byte works. */
__host__ __device__ __forceinline__ uint32_t pack4(uint32_t w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* the four codes times 64, 16, 4, 1: one full-rate v_dot4_u32_u8 (the multiplier below does not fit 24 bits, so the
     * product is a quarter-rate v_mul_lo_u32) */
    return __builtin_amdgcn_udot4((w >> 1) & 0x03030303u, 0x01041040u, 0u, false);
#else
    return ((((w >> 1) & 0x03030303u) * 0x40100401u) >> 24);
#endif
}

/* Four bytes -> 4 flag bits (first byte in bit 3) from a zero_bytes() result. */
__host__ __device__ __forceinline__ uint32_t flags4(uint32_t z)
{
    return (((z >> 7) * 0x08040201u) >> 24) & 0xFu;
}

/* Sixteen bytes -> 16 flag bits (first byte in bit 15) from four zero_bytes() results. */
__host__ __device__ __forceinline__ uint32_t flags16(uint32_t zx, uint32_t zy, uint32_t zz, uint32_t zw)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* a flagged byte is 0x80: times 128 .. 16 (first dword of a pair) and 8 .. 1 (second) the pair's eight flags come
     * out as one number, times 128 — four v_dot4_u32_u8 instead of four quarter-rate multiplies and their shifts */
    const uint32_t hi = __builtin_amdgcn_udot4(zx, 0x10204080u, __builtin_amdgcn_udot4(zy, 0x01020408u, 0u, false), false);
    const uint32_t lo = __builtin_amdgcn_udot4(zz, 0x10204080u, __builtin_amdgcn_udot4(zw, 0x01020408u, 0u, false), false);
    return ((hi << 8) | lo) >> 7;
#else
    return (flags4(zx) << 12) | (flags4(zy) << 8) | (flags4(zz) << 4) | flags4(zw);
#endif
}

struct Packed {
    uint32_t fwd;   /* 16 bases, base 0 in bits 31:30                            */
    uint32_t bad;   /* 16 bits, base 0 in bit 15: 'N' or '\n'                    */
    uint32_t nl;    /* 16 bits, base 0 in bit 15: '\n' only                      */
};

__host__ __device__ __forceinline__ Packed pack16(uint4 v)
{
    Packed p;
    p.fwd = (pack4(v.x) << 24) | (pack4(v.y) << 16) | (pack4(v.z) << 8) | pack4(v.w);
    uint32_t nx = zero_bytes(v.x ^ 0x0A0A0A0Au), ny = zero_bytes(v.y ^ 0x0A0A0A0Au);
    uint32_t nz = zero_bytes(v.z ^ 0x0A0A0A0Au), nw = zero_bytes(v.w ^ 0x0A0A0A0Au);
    p.nl = flags16(nx, ny, nz, nw);
    uint32_t bx = zero_bytes(v.x ^ 0x4E4E4E4Eu), by = zero_bytes(v.y ^ 0x4E4E4E4Eu);
    uint32_t bz = zero_bytes(v.z ^ 0x4E4E4E4Eu), bw = zero_bytes(v.w ^ 0x4E4E4E4Eu);
    p.bad = p.nl | flags16(bx, by, bz, bw);
    return p;
}

/* Low 32 bits of (hi:lo) >> s, 0 <= s < 32 (one v_alignbit_b32). */
__host__ __device__ __forceinline__ uint32_t funnel_r(uint32_t lo, uint32_t hi, int s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s);     /* (written out, hipcc makes 64-bit shifts of some of them) */
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> s);
#endif
}

/* Reverse-complement image of a packed word: base i moves to bits [2i+1:2i]
 * and is complemented (comp = code ^ 2, :414). */
__host__ __device__ __forceinline__ uint32_t rc_word(uint32_t x)
{
    uint32_t r = __builtin_bitreverse32(x);
    r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
    return r ^ 0xAAAAAAAAu;
}

/* Low 32 bits of DJBHash_encode (:66-76).  Only hash & (H-1) is ever used and
 * H <= 2^32, and h*33+b mod 2^32 needs only h mod 2^32.  The recurrence is linear:
 *   h = 5381*33^8 + S(b0..b3)*33^4 + S(b4..b7),   S(x0..x3) = x0*33^3 + x1*33^2 + x2*33 + x3 < 2^24.
 * On the device S is two v_dot4_u32_u8 (the weights 35937, 1089, 33, 1 split into
 * low and high bytes) and one shift-add; *33^4 is four v_lshl_add_u32: 11 full-rate
 * instructions instead of the ~28 (with quarter-rate multiplies) hipcc makes of the loop. */
__host__ __device__ __forceinline__ uint32_t djb_s4(uint32_t w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    /* byte i of w times byte i of the weight word: 35937 = 140*256 + 97, 1089 = 4*256 + 65 */
    const uint32_t lo = __builtin_amdgcn_udot4(w, 0x01214161u, 0u, false);
    const uint32_t hi = __builtin_amdgcn_udot4(w, 0x0000048Cu, 0u, false);
    return (hi << 8) + lo;
#else
    return (w & 0xFFu) * 35937u + ((w >> 8) & 0xFFu) * 1089u + ((w >> 16) & 0xFFu) * 33u + (w >> 24);
#endif
}

__host__ __device__ __forceinline__ uint32_t djb32(uint64_t kmer)
{
    uint32_t x = djb_s4((uint32_t)kmer);
#if defined(__HIP_DEVICE_COMPILE__)
    x = __umul24(x, 1185921u);       /* * 33^4: both factors < 2^24 (S < 2^24), one full-rate v_mul_u32_u24; hipcc turns
                                        four shift-adds into a quarter-rate v_mul_lo_u32 */
#else
    x = (x << 5) + x; x = (x << 5) + x; x = (x << 5) + x; x = (x << 5) + x;      /* * 33^4 */
#endif
    return x + djb_s4((uint32_t)(kmer >> 32)) + 0x7BB34A05u;                     /* + 5381 * 33^8 mod 2^32 */
}

/* ---- bucket records: a k-mer inside its finest bucket, in 6 bytes instead of 8 ----------------------------------
 *
 * The finest level of the partition (qm2_partition.hip.h) holds the k-mers of ONE bucket of S = 2^lg_s home slots,
 * so the bucket bits of their hash (home slot, :92) are known from where they lie.  The DJB hash (:66-76) is linear in
 * the k-mer's bytes b0..b7 (b0 lowest):
 *       h = C + 33^3 * V + (1089 b5 + 33 b6 + b7)   (mod 2^32),     V = 33^4 b0 + 33^3 b1 + 33^2 b2 + 33 b3 + b4,
 * and V < 2^29 never wraps, so (33^3 being odd) the FULL hash and b5..b7 give V exactly.  V is a number in base 33
 * whose digits b1..b4 run up to 255; a digit is known from its residue (V mod 33) as soon as its top three bits are:
 * the values of one residue lie 33 apart and a top-three-bit class is 32 wide.  b0 is what is left.  Hence
 *       k-mer (60 bits at k = 30)  <->  ( h (32 bits),  R = b5, b6, low half of b7 and the top three bits of b1..b4 (32 bits) )
 * is one to one: 28 bits of the k-mer are never written.  (k = 31: the two bits above R's 20 travel beside the hash
 * as `x`.)  A bucket record is R plus what the bucket does not say about h — its low lg_s bits (which is the home
 * slot inside the bucket: bucket_count does not hash at all), and the bits above log2 H for tables below 2^32 slots:
 *       w = (h & (S - 1)) | 0 << lg_s (two code bits, see table_image_kernel) | ((x:h) >> lgH) << (lg_s + 2)
 * as 16 bits when that fits (hg38 scale: 12 + 2 + 0..2), else 32.  Checked against a decoder in tests/emul. */
__host__ __device__ __forceinline__ uint32_t rec_r(uint64_t key)
{
    const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    /* bits 0..2 b4 >> 5, 3..22 b5 b6 b7 & 15 (hi's bits 5..27 as they lie), 23..25 b2 >> 5, 26..28 b1 >> 5, 29..31 b3 >> 5 */
    return ((hi >> 5) & 0x7FFFFFu) | (((lo >> 21) & 7u) << 23) | (((lo >> 13) & 7u) << 26) | (lo & 0xE0000000u);
}
/* the record's second word for a k-mer of hash h (code bits 0) */
__host__ __device__ __forceinline__ uint32_t rec_w(uint64_t key, uint32_t h, uint32_t lgH, uint32_t lg_s)
{
    const uint32_t x = (uint32_t)(key >> 60);                               /* k = 31 only */
    const uint32_t ext = (uint32_t)((((uint64_t)x << 32) | h) >> lgH);      /* lgH <= 32 */
    const uint32_t home = lgH >= 32 ? h : h & ((1u << lgH) - 1u);
    return (home & ((1u << lg_s) - 1u)) | (ext << (lg_s + 2u));
}
/* The two CODE bits above the slot bits of a record's second word.
 *   k-mers:       0.
 *   table image:  where the key's home bucket lies from the slot's own: 0 the same, 1 the one before (lower slots),
 *                 2 the one behind, 3 = no key that any k-mer of this or a neighbouring bucket could equal: with slot
 *                 bits 0 an EMPTY slot, with slot bits 1 a key from further away (a run of > S occupied slots).
 *   staged (LDS): 0 = may equal a k-mer of the bucket being counted (home bucket = that bucket: code 0 in its own slots,
 *                 1 / 2 in the halo of a bucket that walks up / down), 1 = occupied by a key of another bucket,
 *                 3 = empty.  So one 64-bit compare of a staged record with a k-mer's is the reference's T[i] == key. */
constexpr uint32_t kCodeEmpty = 3u, kCodeOther = 1u;
/* the image's second word for `key` in table slot `slot` (R is rec_r(key), 0 for an empty slot) */
__host__ __device__ __forceinline__ uint32_t rec_image_w(uint64_t key, uint64_t slot, uint32_t hmask, uint32_t lgH, uint32_t lg_s)
{
    if (key == 0) return kCodeEmpty << lg_s;
    const uint32_t h = djb32(key);
    const uint64_t bh = (h & hmask) >> lg_s, bp = slot >> lg_s;
    const uint32_t w = rec_w(key, h, lgH, lg_s);
    if (bh == bp) return w;
    if (bh + 1 == bp) return w | (1u << lg_s);
    if (bh == bp + 1) return w | (2u << lg_s);
    return (kCodeEmpty << lg_s) | 1u;
}
/* an image word as bucket_count stages it for a bucket that walks down / up, in that bucket's own slots or its halo */
__host__ __device__ __forceinline__ uint32_t rec_staged_w(uint32_t w, bool halo, bool down, uint32_t lg_s)
{
    const uint32_t mask = 3u << lg_s, code = w & mask, want = halo ? (down ? 2u : 1u) << lg_s : 0u;
    const bool empty = code == mask && (w & ((1u << lg_s) - 1u)) == 0u;
    return empty ? mask : (w & ~mask) | (code == want ? 0u : kCodeOther << lg_s);
}
__host__ __device__ __forceinline__ bool rec_staged_empty(uint32_t staged_w, uint32_t lg_s) { return ((staged_w >> lg_s) & 3u) == kCodeEmpty; }
/* bits of w beside the slot and the code: what must fit 16 - (lg_s + 2) bits for the 2-byte form, 32 - (lg_s + 2) at all */
__host__ __device__ __forceinline__ uint32_t rec_ext_bits(uint32_t k, uint32_t lgH) { return (32u - lgH) + (k > 30 ? 2u * (k - 30u) : 0u); }

/* The 16 bytes of lane-slot `g` (byte offset from sv.base, a multiple of 16; may be negative
 * or beyond the end), in two steps so that the load can be issued early (software prefetch)
 * and stay in flight: load16_raw is a single unconditional aligned load from a clamped
 * address — any 16-byte aligned block that overlaps the payload lies inside its allocation's
 * pages, so it is always safe to read whole — and fix16, applied where the bytes are consumed,
 * turns everything outside [lead, n_total) into '\n'. */
__host__ __device__ __forceinline__ uint4 load16_raw(const SeqView &sv, int64_t g)
{
    const int64_t last = (int64_t)((sv.n_total - 1) & ~(uint64_t)15);
    const int64_t gc = g < 0 ? 0 : (g > last ? last : g);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(sv.base + gc));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__host__ __device__ __forceinline__ uint32_t keep_bytes(uint32_t w, int lo, int hi)
{
    /* keep bytes [lo, hi) of the dword (positions 0..3), the others become '\n' */
    lo = lo < 0 ? 0 : lo;
    hi = hi > 4 ? 4 : hi;
    if (lo >= hi) return 0x0A0A0A0Au;
    const uint32_t m = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & (0xFFFFFFFFu << (8 * lo));
    return (w & m) | (0x0A0A0A0Au & ~m);
}

__host__ __device__ __forceinline__ uint4 fix16(const SeqView &sv, int64_t g, uint4 v)
{
    const int64_t lo = (int64_t)sv.lead - g, hi = (int64_t)sv.n_total - g;   /* valid bytes: [lo, hi) */
    if (lo <= 0 && hi >= 16) return v;                                        /* the common case */
    if (hi <= 0 || lo >= 16) return make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
    const int l = (int)(lo < 0 ? 0 : lo), h = (int)(hi > 16 ? 16 : hi);
    return make_uint4(keep_bytes(v.x, l, h), keep_bytes(v.y, l - 4, h - 4),
                      keep_bytes(v.z, l - 8, h - 8), keep_bytes(v.w, l - 12, h - 12));
}

__host__ __device__ __forceinline__ uint4 load16(const SeqView &sv, int64_t g)
{
    return fix16(sv, g, load16_raw(sv, g));
}

/* ---- packed input (include/qm2.h, "packed sequence"; the host side is qm2_pack.c) --------------------------------
 * A lane's 16 positions are unit u = g / 16 of the stream: 4 bytes of codes and 2 bytes of reset flags inside block
 * u / 4 (24 bytes: codes of positions 0..31, of 32..63, flags).  The host has done pack16's work — (c >> 1) & 3 and
 * the reset test, QuicKmer.c:403-411 — so what is left is the bit order: the stream is first-position-lowest, the
 * window code wants base 0 on top: one v_bfrev_b32 each.  load16_packed_raw is load16_raw's counterpart (a clamped
 * address, so that it can be issued a chunk early), unpack16 that of pack16(fix16()). */
__host__ __device__ __forceinline__ uint4 load16_packed_raw(const SeqView &sv, int64_t g)
{
    const int64_t last = (int64_t)(sv.n_total >> 4) - 1;
    int64_t u = g >> 4;                                   /* (g is a multiple of 16, possibly negative) */
    u = u < 0 ? 0 : (u > last ? last : u);
    const uint8_t *blk = sv.base + (uint64_t)(u >> 2) * 24u;
    const uint32_t sub = (uint32_t)u & 3u;
    const uint32_t codes = *reinterpret_cast<const uint32_t *>(blk + 4u * sub);
    const uint32_t flags = *reinterpret_cast<const uint16_t *>(blk + 16u + 2u * sub);
    return make_uint4(codes, flags, 0u, 0u);
}

__host__ __device__ __forceinline__ Packed unpack16(const SeqView &sv, int64_t g, uint4 v)
{
    Packed p;
    /* (whole blocks: a lane's 16 positions are all inside the stream or all outside, where everything is a line end) */
    const bool in = g >= 0 && g < (int64_t)sv.n_total;
    p.fwd = __builtin_bitreverse32(v.x);                  /* code high bit in bit 2q, low bit in 2q + 1 -> base 0 in bits 31:30 */
    p.bad = in ? __builtin_bitreverse32(v.y) >> 16 : 0xFFFFu;
    p.nl = 0;                                             /* (line ends are not told from 'N' here: the host counts them, and
                                                              the launch takes them off n_bases — stats_sub_bases_kernel) */
    return p;
}

/* The 48-base window of one lane: its own 16 bases and the 32 before them. */
struct Window {
    uint32_t w0, w1, w2;     /* forward: own, lane-1, lane-2 (w2 oldest)         */
    uint32_t r0, r1, r2;     /* reverse-complement image: r0 oldest 16 bases     */
    uint64_t bad;            /* 48 flags, oldest base in bit 47                  */
    uint32_t emit;           /* 16 bits, base 0 in bit 15: this position may emit */
    uint32_t n_bases;        /* sequence characters among the own 16 that count  */
};

/* Window of the lane whose own 16 bytes start at byte g (relative to sv.base),
 * given its packed bytes p and those of the two lanes before it (p1 = lane-1,
 * p2 = lane-2).  Pure function: runs on the host too (tests/emul). */
template <bool K30>
__host__ __device__ __forceinline__ Window assemble_window(const SeqView &sv, int64_t g, bool halo_lane,
                                                           Packed p, Packed p1, Packed p2, uint32_t k)
{
    Window W;
    W.w0 = p.fwd;
    W.w1 = p1.fwd;
    W.w2 = p2.fwd;
    W.bad = ((uint64_t)p2.bad << 32) | ((uint64_t)p1.bad << 16) | p.bad;
    W.r0 = rc_word(W.w2);
    W.r1 = rc_word(W.w1);
    W.r2 = rc_word(W.w0);
    /* positions of this lane that lie in [emit_from, n_total) */
    uint32_t in_range = 0;
    {
        int64_t lo = (int64_t)sv.emit_from - g, hi = (int64_t)sv.n_total - g; /* j in [lo, hi) */
        lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
        hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
        if (hi > lo) in_range = ((0xFFFFu >> lo) & ~(0xFFFFu >> hi)) & 0xFFFFu;
    }
    if (halo_lane) in_range = 0;                   /* halo lanes never emit      */
    W.n_bases = __builtin_popcount(in_range & ~p.nl);
    /* emission: the last k characters hold no reset (run >= k, :418).  smear = OR of the reset
     * flags of the k positions ending at each position, by doubling (k <= 31). */
    uint32_t emit;
    {
        const uint32_t kk = K30 ? 30u : k;
        uint64_t sm = W.bad;                    /* width 1 */
        uint32_t w = 1;
#pragma unroll
        for (int step = 0; step < 5; step++) {
            if (2 * w <= kk) { sm |= sm >> w; w *= 2; }
        }
        if (w < kk) sm |= sm >> (kk - w);       /* width kk: bit p = OR of bad[p .. p+kk-1] */
        /* bit p of sm is set iff some reset lies in the kk positions ending at bit p
         * (older bases sit in higher bits); own position j is bit 15-j */
        emit = (uint32_t)(~sm) & 0xFFFFu;
    }
    emit &= in_range;
    /* the reference's run counter is a uint16 (:402): in an N-free run the
     * windows ending at run positions 65536*m .. 65536*m + k-1 are dropped.
     * The host (or the caller) lists the byte position of each run position
     * 65536*m; windows ending in [s, s+k-1] are suppressed. */
    if (sv.n_suppress) {
        /* list entries are relative to payload byte 0 (= base + lead) */
        const int64_t hi_pos = g + 15 - (int64_t)sv.lead;
        if (hi_pos >= 0) {
            uint64_t a = 0, b = sv.n_suppress;           /* first entry > hi_pos */
            while (a < b) {
                uint64_t m = (a + b) >> 1;
                if (sv.suppress[m] <= (uint64_t)hi_pos) a = m + 1; else b = m;
            }
            if (a > 0) {
                const int64_t s = (int64_t)sv.suppress[a - 1] + (int64_t)sv.lead;
                const int kk = K30 ? 30 : (int)k;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int64_t pos = g + j;
                    if (pos >= s && pos < s + kk) emit &= ~(0x8000u >> j);
                }
            }
        }
    }
    W.emit = emit;
    return W;
}

/* ---- where a k-mer lives in a partition level's buffer (qm2_partition.hip.h, SplitDst) ----
 * child-major (SEG = false): k-mer e of region r at r * cap + e.
 * segmented   (SEG = true):  segments of kSegKmers k-mers laid out [segment][region]:
 *                            ((e >> kSegLg) << lg_row) + (r << kSegLg) + (e & (kSegKmers - 1)),  2^lg_row = regions * kSegKmers.
 * Host-callable: tests/emul checks that both are one-to-one and that a 16-byte pair never straddles a segment. */
#ifndef QM2_SEG_LG
#define QM2_SEG_LG 13
#endif
constexpr uint32_t kSegLg = QM2_SEG_LG, kSegKmers = 1u << kSegLg;

template <bool SEG>
__host__ __device__ __forceinline__ uint64_t slot_of(uint64_t region, uint32_t e, uint32_t cap, uint32_t lg_row)
{
    if constexpr (SEG)
        return ((uint64_t)(e >> kSegLg) << lg_row) + (((uint32_t)region << kSegLg) | (e & (kSegKmers - 1u)));
    else
        return region * (uint64_t)cap + e;
}

/* Byte offset (from sv.base) of the 16 bytes lane `lane` owns in the tile that
 * starts at byte tile0. */
__host__ __device__ __forceinline__ int64_t lane_offset(uint64_t tile0, int lane)
{
    return (int64_t)tile0 - kHaloLanes * kBytesPerLane + (int64_t)lane * kBytesPerLane;
}

/* The value of the lane below (lane 0 keeps its own): DPP wave_shr:1, a vector-ALU move — __shfl_up is a ds_bpermute_b32,
 * an LDS round trip in front of everything a chunk does.  Two steps stand in for __shfl_up(x, 2): lane 1 then holds lane
 * 0's value, not its own, which nobody looks at (lanes < kHaloLanes never emit). */
__device__ __forceinline__ uint32_t lane_up1(uint32_t x)
{
#ifndef QM2_LANE_UP_DPP
#define QM2_LANE_UP_DPP 1
#endif
#if QM2_LANE_UP_DPP
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138, 0xF, 0xF, false);
#else
    return __shfl_up(x, 1);
#endif
}

/* Build the window for `lane` of a wave whose tile starts at byte tile0.
 * All 64 lanes must call (lane shifts inside).  PK: packed input. */
template <bool K30, bool PK = false>
__device__ __forceinline__ Window make_window(const SeqView &sv, uint64_t tile0, int lane, uint32_t k)
{
    const int64_t g = lane_offset(tile0, lane);
    const Packed p = PK ? unpack16(sv, g, load16_packed_raw(sv, g)) : pack16(load16(sv, g));
    Packed p1, p2;
    p1.fwd = lane_up1(p.fwd); p1.bad = lane_up1(p.bad); p1.nl = 0;
    p2.fwd = lane_up1(p1.fwd); p2.bad = lane_up1(p1.bad); p2.nl = 0;
    return assemble_window<K30>(sv, g, lane < kHaloLanes, p, p1, p2, k);
}

/* Same, with the lane's 16 bytes already fetched by load16_raw / load16_packed_raw (software prefetch). */
template <bool K30, bool PK = false>
__device__ __forceinline__ Window make_window_from(const SeqView &sv, uint64_t tile0, int lane, uint32_t k, uint4 bytes)
{
    const int64_t g = lane_offset(tile0, lane);
    const Packed p = PK ? unpack16(sv, g, bytes) : pack16(fix16(sv, g, bytes));
    Packed p1, p2;
    p1.fwd = lane_up1(p.fwd); p1.bad = lane_up1(p.bad); p1.nl = 0;
    p2.fwd = lane_up1(p1.fwd); p2.bad = lane_up1(p1.bad); p2.nl = 0;
    return assemble_window<K30>(sv, g, lane < kHaloLanes, p, p1, p2, k);
}

/* Canonical k-mer of own position j (compile-time j), rules of :412-420. */
template <bool K30, int J>
__host__ __device__ __forceinline__ uint64_t kmer_at(const Window &W, uint32_t k)
{
    constexpr int s = 2 * (15 - J);
    uint32_t flo = funnel_r(W.w0, W.w1, s);
    uint32_t fhi = funnel_r(W.w1, W.w2, s);
    uint64_t fwd = ((uint64_t)fhi << 32) | flo;
    constexpr int t = 2 * (3 + J);
    uint32_t rlo, rhi;
    if constexpr (t < 32) {
        rlo = funnel_r(W.r0, W.r1, t);
        rhi = funnel_r(W.r1, W.r2, t);
    } else {
        rlo = funnel_r(W.r1, W.r2, t - 32);
        rhi = W.r2 >> (t - 32);
    }
    uint64_t rc = (((uint64_t)rhi << 32) | rlo) & kMask60;
    if (K30) {
        fwd &= kMask60;
    } else {
        fwd &= (1ull << (2 * k)) - 1;                   /* :419 */
        /* rc holds only the bases since the last reset (:404-407 zero it):
         * keep the top 2*m bits, m = min(run, 30) */
        uint64_t b = (W.bad >> (15 - J)) | (1ull << 30);
        int m = __builtin_ctzll(b);              /* trailing clean bases */
        if (m < 30) rc &= ~((1ull << (2 * (30 - m))) - 1);
    }
    return fwd < rc ? fwd : rc;                          /* :420 */
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

}  // namespace qm2
