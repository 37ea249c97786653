/*
 * qm2_pack_body.h — the packers of qm2_pack.c, compiled once per instruction set: the including file defines
 *   QM2_ISA            suffix of the functions made here (avx512 / avx2 / portable)
 *   QM2_TARGET         the function attribute that selects the instruction set ("" for the portable build)
 *   QM2_MASKS64(p, hi, lo, nl, nn)   the four 64-bit byte masks of the 64 text bytes at p (bit i = byte i):
 *                      hi = bit 2 of the byte, lo = bit 1 (code = (c >> 1) & 3 = hi:lo, QuicKmer.c:411), nl = c == '\n', nn = c == 'N'
 *   QM2_NLMASK(p)      the nl mask alone (a group inside a dropped line needs nothing else)
 *   QM2_GTMASK(p)      c == '>' (lines are dropped for their first byte, QuicKmer.c:398)
 *   QM2_HAVE_PEXT      defined where BMI2 is there: pack_reads then handles a whole 64-byte group at once
 *   QM2_SPREAD(x)      bit q of the 32-bit x -> bit 2q of a 64-bit word
 *   QM2_STORE64(p, v)  one word of a block into the output (qm2_pack.c: non-temporal in the x86 builds)
 */
#define QM2_CAT2(a, b) a##_##b
#define QM2_CAT(a, b) QM2_CAT2(a, b)
#define QM2_FN(name) QM2_CAT(name, QM2_ISA)

/* one block = 64 positions: codes of positions 0..31, of 32..63, reset flags (include/qm2.h, "packed sequence") */
QM2_TARGET static inline void QM2_FN(put_block)(uint64_t *dst, uint64_t hi, uint64_t lo, uint64_t bad)
{
    QM2_STORE64(dst, QM2_SPREAD((uint32_t)hi) | (QM2_SPREAD((uint32_t)lo) << 1));
    QM2_STORE64(dst + 1, QM2_SPREAD((uint32_t)(hi >> 32)) | (QM2_SPREAD((uint32_t)(lo >> 32)) << 1));
    QM2_STORE64(dst + 2, bad);
}

/* positions appended a run at a time: cnt in 1..64, the bits above cnt zero.  The block being filled lives in a LOCAL
 * accumulator of pack_reads (registers): through the sink in memory every group waited for the stores of the one before. */
typedef struct { uint64_t hi, lo, bad, n_pos, n_newlines; size_t n_blocks; unsigned fill; } QM2_FN(acc_t);
QM2_TARGET static inline __attribute__((always_inline)) int QM2_FN(acc_append)(QM2_FN(acc_t) *a, const pack_sink *s, uint64_t hi, uint64_t lo, uint64_t bad, unsigned cnt)
{
    const unsigned f = a->fill;
    a->hi |= hi << f; a->lo |= lo << f; a->bad |= bad << f;
    a->n_pos += cnt;
    if (f + cnt < 64) { a->fill = f + cnt; return 0; }
    if (a->n_blocks >= s->cap_blocks) return 1;
    QM2_FN(put_block)(s->out + 3 * a->n_blocks++, a->hi, a->lo, a->bad);
    a->fill = f + cnt - 64;
    if (f) { a->hi = hi >> (64 - f); a->lo = lo >> (64 - f); a->bad = bad >> (64 - f); }
    else a->hi = a->lo = a->bad = 0;
    return 0;
}

/* the masks of the n < 64 bytes at p (the rest read as '\n') */
QM2_TARGET static inline void QM2_FN(masks_tail)(const uint8_t *p, size_t n, uint64_t *hi, uint64_t *lo, uint64_t *nl, uint64_t *nn)
{
    uint8_t tmp[64];
    memset(tmp, '\n', sizeof tmp);
    memcpy(tmp, p, n);
    QM2_MASKS64(tmp, hi, lo, nl, nn);
}

QM2_TARGET static inline uint64_t QM2_FN(gt_tail)(const uint8_t *p, size_t n)
{
    uint8_t tmp[64];
    memset(tmp, '\n', sizeof tmp);
    memcpy(tmp, p, n);
    return QM2_GTMASK(tmp);
}

/* n bytes of sequence lines -> blocks; every byte is a position.  Returns the blocks written. */
QM2_TARGET static size_t QM2_FN(pack_lines)(const uint8_t *text, size_t n, uint64_t *dst, uint64_t *n_newlines)
{
    size_t b = 0, i = 0;
    uint64_t newlines = 0;
    for (; i + 64 <= n; i += 64, b++) {
        uint64_t hi, lo, nl, nn;
        QM2_MASKS64(text + i, &hi, &lo, &nl, &nn);
        newlines += (uint64_t)__builtin_popcountll(nl);
        QM2_FN(put_block)(dst + 3 * b, hi, lo, nl | nn);
    }
    if (i < n) {                                   /* the last block is filled up with line ends */
        uint64_t hi, lo, nl, nn;
        QM2_FN(masks_tail)(text + i, n - i, &hi, &lo, &nl, &nn);
        const uint64_t m = (1ull << (n - i)) - 1;                                        /* (the filler reads as line ends, code 0) */
        newlines += (uint64_t)__builtin_popcountll(nl & m);
        QM2_FN(put_block)(dst + 3 * b, hi & m, lo & m, nl | nn);
        b++;
    }
    *n_newlines = newlines;
    return b;
}

/* The FASTA / FASTQ line discipline (QuicKmer.c:393-398, :451-455) and the packing in one pass: `text` is whole lines
 * (a last one without its '\n' gets one); the bits of the sequence lines are appended to the sink a piece of a 64-byte
 * group at a time.  Returns 0, or 2 when the output or the wrap list does not hold it. */
QM2_TARGET static int QM2_FN(pack_reads)(const uint8_t *text, size_t n, qm2_line_state *st, pack_sink *s)
{
    uint32_t skip = st->skip, desync = st->desync, lines = 0;
    const uint32_t fastq = st->fastq;
    int at_line_start = 1, take = 0;
    size_t line_start = 0;                          /* of the line being taken: where it starts in the text ... */
    uint64_t line_pos0 = 0;                         /* ... and in the packed positions */
    QM2_FN(acc_t) a = {0, 0, 0, 0, 0, 0, 0};
    for (size_t g = 0; g < n; g += 64) {
        uint64_t hi = 0, lo = 0, nl, nn = 0;
        size_t cnt = 64;
        int have = 0;                               /* hi / lo / nn of this group are worked out: only for a group that holds kept text */
        if (n - g >= 64) {
            nl = QM2_NLMASK(text + g);
            if (!nl && !take && !at_line_start) continue;                             /* inside a dropped line */
        } else {
            cnt = n - g;
            QM2_FN(masks_tail)(text + g, cnt, &hi, &lo, &nl, &nn);
            nl &= (1ull << cnt) - 1;                /* (the filler is not text) */
            have = 1;
        }
#ifdef QM2_HAVE_PEXT
        /* A whole group at once, no branch per line.  Which bytes belong to kept lines follows from three masks: where
         * lines start (behind every newline; at byte 0 when the group opens at a line start), which of those starts
         * are kept — FASTQ: every fourth, counted from `skip` (pdep deals the pattern 0001 0001 ... onto the start bits);
         * FASTA: those whose byte is not '>' — and the borrow chain of (nl - kept starts), which runs from every kept
         * start up to the newline that ends its line (or off the top: the line goes on in the next group).  The kept
         * bytes' code and reset bits are then gathered with pext and appended in one piece.  The one thing this cannot
         * follow is a FASTQ sequence-position line that starts with '>' (:398 drops it WITHOUT the three discards, so
         * the lines behind it count differently): such a group takes the line-by-line loop below. */
        {
            const uint64_t valid = cnt >= 64 ? ~0ull : (1ull << cnt) - 1;
            const uint64_t starts = ((nl << 1) | (uint64_t)(at_line_start != 0)) & valid;
            uint64_t kept;
            int fast = 1;
            {
                /* (no test for "no line starts here": pdep of nothing is nothing, and the branch would be a coin toss) */
                const uint64_t gt = cnt >= 64 ? QM2_GTMASK(text + g) : QM2_FN(gt_tail)(text + g, cnt);
                if (fastq) {
                    kept = _pdep_u64(0x1111111111111111ull << skip, starts);
                    if (kept & gt) fast = 0;
                } else kept = starts & ~gt;
            }
            if (fast) {
                const uint64_t from = kept | (uint64_t)(!at_line_start && take);      /* a kept line that is already open starts "at byte 0" */
                const uint64_t sel = ((nl - from) ^ nl) & valid;
                if (sel) {
                    if (!have) { uint64_t nl2; QM2_MASKS64(text + g, &hi, &lo, &nl2, &nn); have = 1; }
                    const unsigned m = (unsigned)__builtin_popcountll(sel);
                    /* the open kept line ends here (its newline is the lowest one): was it long enough for the run counter to wrap? */
                    if (!at_line_start && take && nl) {
                        const size_t line_len = g + (size_t)__builtin_ctzll(nl) - line_start;
                        if (line_len >= 65536 && sink_wraps(s, text + line_start, line_len, line_pos0)) return 2;
                    }
                    if (QM2_FN(acc_append)(&a, s, _pext_u64(hi, sel), _pext_u64(lo, sel), _pext_u64(nl | nn, sel), m)) return 2;
                    a.n_newlines += (uint64_t)__builtin_popcountll(sel & nl);
                }
                const unsigned n_starts = (unsigned)__builtin_popcountll(starts);
                lines += n_starts;
                if (fastq) skip = (skip + 4u - (n_starts & 3u)) & 3u;                 /* one step of :451-455 per line start */
                const int ends_with_nl = (int)((nl >> (cnt - 1)) & 1);
                const int last_kept = (int)((sel >> (cnt - 1)) & 1);
                if (!ends_with_nl && last_kept && kept && (!nl || (63 - __builtin_clzll(kept)) > (63 - __builtin_clzll(nl)))) {
                    /* a kept line that started in this group stays open: where it starts, in the text and in the positions */
                    const unsigned b = 63u - (unsigned)__builtin_clzll(kept);
                    line_start = g + b;
                    line_pos0 = a.n_pos - (uint64_t)__builtin_popcountll(sel >> b);
                }
                at_line_start = ends_with_nl;
                take = ends_with_nl ? take : last_kept;
                continue;
            }
        }
#endif
        size_t pos = 0;
        while (pos < cnt) {
            if (at_line_start) {
                at_line_start = 0;
                lines++;
                const uint8_t first = text[g + pos];
                if (skip) { skip--; take = 0; }                                       /* :452-454 */
                else if (first == '>') { if (fastq) desync = 1; take = 0; }           /* :398 (before :451: no discards) */
                else {
                    take = 1;
                    if (fastq) skip = 3;
                    line_start = g + pos;
                    line_pos0 = a.n_pos;
                }
            }
            const uint64_t rest = pos < 64 ? nl >> pos : 0;
            size_t len;                             /* positions of this line inside the group, its '\n' included */
            int ends;
            if (rest) { len = (size_t)__builtin_ctzll(rest) + 1; ends = 1; }
            else { len = cnt - pos; ends = 0; }
            if (take) {
                if (!have) { uint64_t nl2; QM2_MASKS64(text + g, &hi, &lo, &nl2, &nn); have = 1; }
                const uint64_t m = len >= 64 ? ~0ull : (1ull << len) - 1;
                if (QM2_FN(acc_append)(&a, s, (hi >> pos) & m, (lo >> pos) & m, ((nl | nn) >> pos) & m, (unsigned)len)) return 2;
                if (ends) {
                    a.n_newlines++;
                    const size_t line_len = g + pos + len - 1 - line_start;           /* without the '\n' */
                    if (line_len >= 65536 && sink_wraps(s, text + line_start, line_len, line_pos0)) return 2;
                }
            }
            pos += len;
            if (ends) at_line_start = 1;
        }
    }
    if (!at_line_start && take) {                   /* the text's last line had no '\n': supplied (a divergence where the reference is undefined) */
        if (QM2_FN(acc_append)(&a, s, 0, 0, 1, 1)) return 2;
        a.n_newlines++;
        if (n - line_start >= 65536 && sink_wraps(s, text + line_start, n - line_start, line_pos0)) return 2;
    }
    if (a.fill) {                                   /* fill the last block up with line ends */
        const unsigned pad = 64 - a.fill;
        if (QM2_FN(acc_append)(&a, s, 0, 0, pad >= 64 ? ~0ull : (1ull << pad) - 1, pad)) return 2;
        a.n_newlines += pad;
    }
    s->n_blocks = a.n_blocks;
    s->n_newlines = a.n_newlines;
    s->n_pos = a.n_pos;
    st->skip = skip;
    st->desync = desync;
    st->lines += lines;
    return 0;
}

#undef QM2_FN
#undef QM2_CAT
#undef QM2_CAT2
