/*
 * qm2_pack.c — host side of the PACKED input format (include/qm2.h, "packed sequence"): sequence text as 2-bit codes
 * and a reset flag per position, 3 bits a base instead of 8 over the PCIe link.
 *
 * What the host does here is the first two operations the reference's producer thread does per character
 * (QuicKmer.c:403-411): the reset test (c == 'N', the line end) and the code (c >> 1) & 3 — nothing else.  Rolling the
 * k-mer, its reverse complement, canonicalising, hashing, probing and counting (:412-420, :66-99, :291) stay on the
 * device, which reads these blocks instead of ASCII (qm2_device.hip.h, unpack16).  Plain C, no device code; part of
 * libqm2_hip.so so that the CLI's reader threads, qm2_submit and bench.py share one implementation.
 *
 * Three builds of the same loops (qm2_pack_body.h), chosen once at run time: AVX-512BW + BMI2 (64 text bytes per
 * compare, the masks straight from vpmovb2m / vpcmpeqb, the 2-bit interleave by pdep), AVX2 + BMI2, and portable C.
 */
#define _GNU_SOURCE
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/qm2.h"

/* ---- the bit sink of the line-discipline packer: positions are appended a run at a time ---------------------------- */
typedef struct {
    uint64_t *out;                 /* blocks (3 words each) */
    size_t    cap_blocks, n_blocks;
    uint64_t  n_pos;               /* positions appended */
    uint64_t  n_newlines;
    uint64_t *wraps;
    size_t    wraps_cap, n_wraps;
} pack_sink;

/* where the reference's uint16 run counter wraps inside one taken line (qm2_plan.h, find_wraps: the byte of run position
 * 65536 m of every N-free run), as packed positions */
static int sink_wraps(pack_sink *s, const uint8_t *line, size_t len, uint64_t pos0)
{
    size_t r0 = 0;
    while (r0 < len) {
        const uint8_t *nn = memchr(line + r0, 'N', len - r0);
        const size_t r1 = nn ? (size_t)(nn - line) : len;
        for (size_t m = 65536; m <= r1 - r0; m += 65536) {
            if (s->n_wraps >= s->wraps_cap) return 1;
            s->wraps[s->n_wraps++] = pos0 + r0 + m - 1;
        }
        r0 = r1 + 1;
    }
    return 0;
}

/* ---- the three builds --------------------------------------------------------------------------------------------- */

static inline uint64_t spread_portable(uint32_t x)
{
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
/* bit i = (byte i of the 8 bytes in w has bit `bit` set) */
static inline uint64_t bits8(uint64_t w, int bit) { return (((w >> bit) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56; }
static inline uint64_t eq8(uint64_t w, uint8_t c)
{
    const uint64_t t = w ^ (0x0101010101010101ull * c), lo7 = 0x7F7F7F7F7F7F7F7Full;
    return bits8(~(((t & lo7) + lo7) | t | lo7), 7);
}
static inline void masks64_portable(const uint8_t *p, uint64_t *hi, uint64_t *lo, uint64_t *nl, uint64_t *nn)
{
    uint64_t h = 0, l = 0, a = 0, b = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        h |= bits8(w, 2) << (8 * i);
        l |= bits8(w, 1) << (8 * i);
        a |= eq8(w, '\n') << (8 * i);
        b |= eq8(w, 'N') << (8 * i);
    }
    *hi = h; *lo = l; *nl = a; *nn = b;
}
static inline uint64_t nlmask_portable(const uint8_t *p)
{
    uint64_t a = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t w;
        memcpy(&w, p + 8 * i, 8);
        a |= eq8(w, '\n') << (8 * i);
    }
    return a;
}
#define QM2_ISA portable
#define QM2_TARGET
#define QM2_MASKS64(p, hi, lo, nl, nn) masks64_portable((const uint8_t *)(p), hi, lo, nl, nn)
#define QM2_NLMASK(p) nlmask_portable((const uint8_t *)(p))
#define QM2_GTMASK(p) 0ull          /* (only the pext builds look at it) */
#define QM2_SPREAD(x) spread_portable(x)
#include "qm2_pack_body.h"
#undef QM2_ISA
#undef QM2_TARGET
#undef QM2_MASKS64
#undef QM2_NLMASK
#undef QM2_GTMASK
#undef QM2_HAVE_PEXT
#undef QM2_SPREAD

#if defined(__x86_64__)
#define AVX2_T __attribute__((target("avx2,bmi2")))
AVX2_T static inline void masks64_avx2(const uint8_t *p, uint64_t *hi, uint64_t *lo, uint64_t *nl, uint64_t *nn)
{
    const __m256i a = _mm256_loadu_si256((const __m256i *)p), b = _mm256_loadu_si256((const __m256i *)(p + 32));
    const __m256i cn = _mm256_set1_epi8('\n'), cN = _mm256_set1_epi8('N');
    /* (a 16-bit shift moves bit 2 / bit 1 of every byte into its own bit 7: nothing else of the result is looked at) */
    *hi = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(a, 5)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 5)) << 32);
    *lo = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(a, 6)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(b, 6)) << 32);
    *nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, cn)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, cn)) << 32);
    *nn = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, cN)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, cN)) << 32);
}
AVX2_T static inline uint64_t eqmask_avx2(const uint8_t *p, char ch)
{
    const __m256i cn = _mm256_set1_epi8(ch);
    return (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)p), cn)) |
           ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + 32)), cn)) << 32);
}
#define QM2_ISA avx2
#define QM2_TARGET AVX2_T
#define QM2_MASKS64(p, hi, lo, nl, nn) masks64_avx2((const uint8_t *)(p), hi, lo, nl, nn)
#define QM2_NLMASK(p) eqmask_avx2((const uint8_t *)(p), '\n')
#define QM2_GTMASK(p) eqmask_avx2((const uint8_t *)(p), '>')
#define QM2_HAVE_PEXT 1
#define QM2_SPREAD(x) _pdep_u64((uint64_t)(x), 0x5555555555555555ull)
#include "qm2_pack_body.h"
#undef QM2_ISA
#undef QM2_TARGET
#undef QM2_MASKS64
#undef QM2_NLMASK
#undef QM2_GTMASK
#undef QM2_HAVE_PEXT
#undef QM2_SPREAD

#define AVX512_T __attribute__((target("avx512f,avx512bw,bmi2")))
AVX512_T static inline void masks64_avx512(const uint8_t *p, uint64_t *hi, uint64_t *lo, uint64_t *nl, uint64_t *nn)
{
    const __m512i v = _mm512_loadu_si512((const void *)p);
    *hi = (uint64_t)_mm512_movepi8_mask(_mm512_slli_epi16(v, 5));
    *lo = (uint64_t)_mm512_movepi8_mask(_mm512_slli_epi16(v, 6));
    *nl = (uint64_t)_mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('\n'));
    *nn = (uint64_t)_mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('N'));
}
#define QM2_ISA avx512
#define QM2_TARGET AVX512_T
#define QM2_MASKS64(p, hi, lo, nl, nn) masks64_avx512((const uint8_t *)(p), hi, lo, nl, nn)
#define QM2_NLMASK(p) ((uint64_t)_mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void *)(p)), _mm512_set1_epi8('\n')))
#define QM2_GTMASK(p) ((uint64_t)_mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void *)(p)), _mm512_set1_epi8('>')))
#define QM2_HAVE_PEXT 1
#define QM2_SPREAD(x) _pdep_u64((uint64_t)(x), 0x5555555555555555ull)
#include "qm2_pack_body.h"
#undef QM2_ISA
#undef QM2_TARGET
#undef QM2_MASKS64
#undef QM2_NLMASK
#undef QM2_GTMASK
#undef QM2_HAVE_PEXT
#undef QM2_SPREAD
#endif

/* ---- dispatch -------------------------------------------------------------------------------------------------------- */

typedef struct {
    const char *name;
    size_t (*lines)(const uint8_t *, size_t, uint64_t *, uint64_t *);
    int    (*reads)(const uint8_t *, size_t, qm2_line_state *, pack_sink *);
} pack_impl;

static const pack_impl impl_portable = {"portable", pack_lines_portable, pack_reads_portable};
#if defined(__x86_64__)
static const pack_impl impl_avx2 = {"avx2", pack_lines_avx2, pack_reads_avx2};
static const pack_impl impl_avx512 = {"avx512", pack_lines_avx512, pack_reads_avx512};
#endif

static const pack_impl *by_name(const char *want)
{
    const pack_impl *c = &impl_portable;
#if defined(__x86_64__)
    __builtin_cpu_init();
    const int has2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    const int has512 = has2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    if (want && !strcmp(want, "portable")) c = &impl_portable;
    else if (want && !strcmp(want, "avx2")) c = has2 ? &impl_avx2 : &impl_portable;
    else c = has512 ? &impl_avx512 : has2 ? &impl_avx2 : &impl_portable;
#else
    (void)want;
#endif
    return c;
}

static _Atomic(const pack_impl *) chosen;      /* (the reader's threads all ask at once: every one of them finds the same answer) */
static const pack_impl *pick_impl(void)
{
    /* the widest build the CPU has; QM2_PACK_ISA = portable | avx2 | avx512 (or qm2_pack_isa_select) forces a narrower one */
    const pack_impl *c = atomic_load_explicit(&chosen, memory_order_relaxed);
    if (!c) { c = by_name(getenv("QM2_PACK_ISA")); atomic_store_explicit(&chosen, c, memory_order_relaxed); }
    return c;
}

const char *qm2_pack_isa(void) { return pick_impl()->name; }
const char *qm2_pack_isa_select(const char *name)
{
    const pack_impl *c = by_name(name);
    atomic_store_explicit(&chosen, c, memory_order_relaxed);
    return c->name;
}

size_t qm2_packed_bound(size_t n_text) { return ((n_text + 1 + 63) / 64) * (size_t)QM2_PACKED_BLOCK_BYTES; }

int qm2_pack_lines(const char *lines, size_t n, void *dst, size_t dst_bytes, uint64_t *wraps, size_t wraps_cap,
                   qm2_packed *out)
{
    if (!out || (n && (!lines || !dst))) { puts("qm2_pack_lines: null argument"); return 1; }
    memset(out, 0, sizeof *out);
    if (n == 0) return 0;
    if (lines[n - 1] != '\n') { puts("qm2_pack_lines: buffer must end with a newline (whole sequence lines only)"); return 1; }
    if ((uintptr_t)dst & 7u) { puts("qm2_pack_lines: the output must be 8-byte aligned"); return 1; }
    if (dst_bytes < ((n + 63) / 64) * (size_t)QM2_PACKED_BLOCK_BYTES) return 2;
    const pack_impl *im = pick_impl();
    /* the run-counter wraps first (lines of >= 65536 bases only: a probe of every 32 KiB block, qm2_plan.h) */
    {
        const size_t B = 32768;
        size_t blk = 0;
        while ((blk + 1) * B <= n) {
            if (memchr(lines + blk * B, '\n', B)) { blk++; continue; }
            const size_t pos = blk * B;
            const char *prev = pos ? memrchr(lines, '\n', pos) : NULL;
            const size_t ls = prev ? (size_t)(prev - lines) + 1 : 0;
            const char *nx = memchr(lines + pos, '\n', n - pos);
            const size_t le = nx ? (size_t)(nx - lines) : n;
            pack_sink s;
            memset(&s, 0, sizeof s);
            s.wraps = wraps; s.wraps_cap = wraps_cap; s.n_wraps = (size_t)out->n_wraps;
            if (sink_wraps(&s, (const uint8_t *)lines + ls, le - ls, ls)) return 2;
            out->n_wraps = s.n_wraps;
            blk = le / B;
        }
    }
    uint64_t newlines = 0;
    const size_t nb = im->lines((const uint8_t *)lines, n, (uint64_t *)dst, &newlines);
    out->n_blocks = nb;
    out->n_newlines = newlines + (64 * (uint64_t)nb - n);                        /* + the filler of the last block */
    return 0;
}

int qm2_pack_reads(const char *text, size_t n, qm2_line_state *st, void *dst, size_t dst_bytes, uint64_t *wraps,
                   size_t wraps_cap, qm2_packed *out)
{
    if (!out || !st || (n && (!text || !dst))) { puts("qm2_pack_reads: null argument"); return 1; }
    if ((uintptr_t)dst & 7u) { puts("qm2_pack_reads: the output must be 8-byte aligned"); return 1; }
    memset(out, 0, sizeof *out);
    const pack_impl *im = pick_impl();
    pack_sink s;
    memset(&s, 0, sizeof s);
    s.out = (uint64_t *)dst;
    s.cap_blocks = dst_bytes / QM2_PACKED_BLOCK_BYTES;
    s.wraps = wraps; s.wraps_cap = wraps_cap;
    if (im->reads((const uint8_t *)text, n, st, &s)) return 2;
    out->n_blocks = s.n_blocks;
    out->n_newlines = s.n_newlines;
    out->n_wraps = s.n_wraps;
    return 0;
}
