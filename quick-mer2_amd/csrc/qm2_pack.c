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
 * Builds of the same loops (qm2_pack_body.h), chosen once at run time: AVX-512BW + BMI2 (64 text bytes per compare, the
 * masks straight from vpmovb2m / vpcmpeqb, the 2-bit interleave by pdep), AVX2 + BMI2, portable C — and, where the CPU has
 * AVX-512 VBMI2, a pack_reads that compacts the kept bytes with vpcompressb first (below).
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

/* The blocks go to a pinned staging buffer that only the copy engine reads: in the x86 builds they are written with
 * non-temporal stores (movnti, combined to whole lines on the way out) — a plain store first READS the line it is about to
 * overwrite, a fifth of the DRAM traffic of a host that feeds at the memory system's limit (EXPERIMENTS.md);
 * qm2_pack_lines / qm2_pack_reads end with an sfence.  QM2_PACK_PLAIN_STORES (compile time) switches it off. */
#if defined(__x86_64__) && !defined(QM2_PACK_PLAIN_STORES)
#define QM2_STORE64(p, v) _mm_stream_si64((long long *)(p), (long long)(v))
#define QM2_STORES_DONE() _mm_sfence()
#else
#define QM2_STORE64(p, v) (*(p) = (v))
#define QM2_STORES_DONE() ((void)0)
#endif

/* ---- the three builds of qm2_pack_body.h (the VBMI2 pack_reads follows them) ----------------------------------------- */

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

#if defined(__x86_64__)
/* ---- the fourth build: AVX-512 VBMI2 (Zen 4 / 5, Ice Lake and later) -------------------------------------------------
 * pack_reads in two stages that never touch a bit accumulator: (1) the line discipline works out, per 64-byte group and
 * with the same mask arithmetic as the pext build (newline / '>' masks, pdep for every fourth line start, the borrow chain
 * of nl - starts), WHICH BYTES are kept, and vpcompressb moves exactly those to the front of a small scratch buffer that
 * stays in the L1 / L2; (2) the compacted text — sequence lines, nothing else — is packed a whole block at a time by the
 * loop of pack_lines (the masks straight from vpmovb2m / vpcmpeqb, two pdep per word).  What it saves are the three pext
 * and the accumulator's data-dependent "block full?" branch per group: 13.5 -> 16.2 GB/s of FASTQ per thread on Zen 5,
 * 9.2 -> 11.7 of FASTA (tools/micro/pack_probe 1 128 <isa> 12 3; the line discipline alone runs at 20.5). */
#define VBMI2_T __attribute__((target("avx512f,avx512bw,avx512vbmi2,bmi2")))
#define QM2_VB_SCRATCH ((size_t)4 << 10)        /* 4 KiB: 16.2 GB/s per thread on Zen 5, 16 KiB: 14.8 (EXPERIMENTS.md) */

/* whole blocks of the scratch's `fill` bytes -> the sink from block `nb` on; the rest (< 64 bytes) moves to the front (or,
 * final, is filled up with line ends).  Returns the blocks written, -1 when the output is full; *newlines grows. */
VBMI2_T static inline __attribute__((always_inline)) long vb_flush(uint8_t *buf, size_t fill, const pack_sink *s, size_t nb, int final, uint64_t *newlines)
{
    size_t i = 0;
    const size_t nb0 = nb;
    uint64_t lines = 0;
    if (nb + fill / 64 + (final && (fill & 63)) > s->cap_blocks) return -1;
    for (; i + 64 <= fill; i += 64) {
        uint64_t hi, lo, nl, nn;
        masks64_avx512(buf + i, &hi, &lo, &nl, &nn);
        lines += (uint64_t)__builtin_popcountll(nl);
        put_block_avx512(s->out + 3 * nb++, hi, lo, nl | nn);
    }
    const size_t rest = fill - i;
    if (final && rest) {                              /* the last block is filled up with line ends */
        uint64_t hi, lo, nl, nn;
        masks_tail_avx512(buf + i, rest, &hi, &lo, &nl, &nn);
        const uint64_t m = (1ull << rest) - 1;
        lines += (uint64_t)__builtin_popcountll(nl & m) + (64 - rest);
        put_block_avx512(s->out + 3 * nb++, hi & m, lo & m, nl | nn);
    } else if (rest && i) _mm512_storeu_si512((void *)buf, _mm512_loadu_si512((const void *)(buf + i)));
    *newlines += lines;
    return (long)(nb - nb0);
}

VBMI2_T static int pack_reads_vbmi2(const uint8_t *text, size_t n, qm2_line_state *st, pack_sink *s)
{
    uint32_t skip = st->skip, desync = st->desync, lines = 0;
    const uint32_t fastq = st->fastq;
    int at_line_start = 1, take = 0;
    size_t line_start = 0;
    uint64_t line_pos0 = 0;
    uint8_t *const buf = aligned_alloc(64, QM2_VB_SCRATCH + 128);     /* the kept bytes, compacted; stays in the L1 */
    if (!buf) return 2;
    size_t fill = 0, n_blocks = 0;                  /* (locals: the stores into buf must not make the compiler reload them) */
    uint64_t n_kept = 0, newlines = 0;
    int rc = 0;
    for (size_t g = 0; g < n && !rc; g += 64) {
        size_t cnt = 64;
        __m512i bytes;
        if (n - g >= 64) bytes = _mm512_loadu_si512((const void *)(text + g));
        else {
            uint8_t tmp[64];
            cnt = n - g;
            memset(tmp, '\n', sizeof tmp);
            memcpy(tmp, text + g, cnt);
            bytes = _mm512_loadu_si512((const void *)tmp);
        }
        const uint64_t valid = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
        const uint64_t nl = (uint64_t)_mm512_cmpeq_epi8_mask(bytes, _mm512_set1_epi8('\n')) & valid;
        if (!nl && !take && !at_line_start) continue;                                 /* inside a dropped line */
        const uint64_t gt = (uint64_t)_mm512_cmpeq_epi8_mask(bytes, _mm512_set1_epi8('>')) & valid;
        const uint64_t starts = ((nl << 1) | (uint64_t)(at_line_start != 0)) & valid;
        uint64_t sel;
        uint64_t kept = fastq ? _pdep_u64(0x1111111111111111ull << skip, starts) : starts & ~gt;
        if (!(fastq && (kept & gt))) {
            const uint64_t from = kept | (uint64_t)(!at_line_start && take);
            sel = ((nl - from) ^ nl) & valid;
            if (!at_line_start && take && nl) {                                       /* the open kept line ends at the lowest newline */
                const size_t line_len = g + (size_t)__builtin_ctzll(nl) - line_start;
                if (line_len >= 65536 && sink_wraps(s, text + line_start, line_len, line_pos0)) { rc = 2; break; }
            }
            const unsigned n_starts = (unsigned)__builtin_popcountll(starts);
            lines += n_starts;
            if (fastq) skip = (skip + 4u - (n_starts & 3u)) & 3u;
            const int ends_with_nl = (int)((nl >> (cnt - 1)) & 1), last_kept = (int)((sel >> (cnt - 1)) & 1);
            if (!ends_with_nl && last_kept && kept && (!nl || (63 - __builtin_clzll(kept)) > (63 - __builtin_clzll(nl)))) {
                const unsigned b = 63u - (unsigned)__builtin_clzll(kept);
                line_start = g + b;
                line_pos0 = n_kept + (uint64_t)__builtin_popcountll(sel & ((1ull << b) - 1));
            }
            at_line_start = ends_with_nl;
            take = ends_with_nl ? take : last_kept;
        } else {
            /* a FASTQ sequence-position line that starts with '>' (:398 drops it WITHOUT the three discards): line by line */
            sel = 0;
            size_t pos = 0;
            while (pos < cnt) {
                if (at_line_start) {
                    at_line_start = 0;
                    lines++;
                    if (skip) { skip--; take = 0; }
                    else if ((gt >> pos) & 1) { desync = 1; take = 0; }
                    else {
                        take = 1;
                        skip = 3;
                        line_start = g + pos;
                        line_pos0 = n_kept + (uint64_t)__builtin_popcountll(sel);
                    }
                }
                const uint64_t rest = nl >> pos;
                size_t len;
                int ends;
                if (rest) { len = (size_t)__builtin_ctzll(rest) + 1; ends = 1; }
                else { len = cnt - pos; ends = 0; }
                if (take) {
                    sel |= (len >= 64 ? ~0ull : (1ull << len) - 1) << pos;
                    if (ends) {
                        const size_t line_len = g + pos + len - 1 - line_start;
                        if (line_len >= 65536 && sink_wraps(s, text + line_start, line_len, line_pos0)) { rc = 2; break; }
                    }
                }
                pos += len;
                if (ends) at_line_start = 1;
            }
            if (rc) break;
        }
        /* the kept bytes of the group to the front of what the scratch holds (a whole vector is stored: what lies behind
         * the kept bytes is overwritten by the next group) */
        _mm512_storeu_si512((void *)(buf + fill), _mm512_maskz_compress_epi8((__mmask64)sel, bytes));
        const unsigned m = (unsigned)__builtin_popcountll(sel);
        fill += m;
        n_kept += m;
        if (fill >= QM2_VB_SCRATCH) {
            const long w = vb_flush(buf, fill, s, n_blocks, 0, &newlines);
            if (w < 0) { rc = 2; break; }
            n_blocks += (size_t)w;
            fill &= 63;
        }
    }
    if (!rc && !at_line_start && take) {            /* the text's last line had no '\n': supplied */
        buf[fill++] = '\n';
        if (n - line_start >= 65536 && sink_wraps(s, text + line_start, n - line_start, line_pos0)) rc = 2;
    }
    if (!rc) {
        const long w = vb_flush(buf, fill, s, n_blocks, 1, &newlines);
        if (w < 0) rc = 2;
        else n_blocks += (size_t)w;
    }
    if (!rc) {
        s->n_blocks = n_blocks;
        s->n_newlines = newlines;
        s->n_pos = 64 * (uint64_t)n_blocks;
        st->skip = skip;
        st->desync = desync;
        st->lines += lines;
    }
    free(buf);
    return rc;
}
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
static const pack_impl impl_vbmi2 = {"avx512vbmi2", pack_lines_avx512, pack_reads_vbmi2};
#endif

static const pack_impl *by_name(const char *want)
{
    const pack_impl *c = &impl_portable;
#if defined(__x86_64__)
    __builtin_cpu_init();
    const int has2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    const int has512 = has2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    const int hasvb = has512 && __builtin_cpu_supports("avx512vbmi2");
    if (want && !strcmp(want, "portable")) c = &impl_portable;
    else if (want && !strcmp(want, "avx2")) c = has2 ? &impl_avx2 : &impl_portable;
    else if (want && !strcmp(want, "avx512")) c = has512 ? &impl_avx512 : has2 ? &impl_avx2 : &impl_portable;
    else c = hasvb ? &impl_vbmi2 : has512 ? &impl_avx512 : has2 ? &impl_avx2 : &impl_portable;
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
    QM2_STORES_DONE();
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
    const int rrc = im->reads((const uint8_t *)text, n, st, &s);
    QM2_STORES_DONE();
    if (rrc) return 2;
    out->n_blocks = s.n_blocks;
    out->n_newlines = s.n_newlines;
    out->n_wraps = s.n_wraps;
    return 0;
}
