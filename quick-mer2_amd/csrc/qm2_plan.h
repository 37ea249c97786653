/*
 * qm2_plan.h — host-only planning for qm2_submit: where the reference's uint16
 * run counter wraps, and how a buffer of sequence lines is cut into staging
 * parts.  Header-only so that the CPU emulation test (tests/emul) exercises the
 * very same code as libqm2_hip.so.
 */
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace qm2 {

constexpr size_t kCarry = 32;   /* look-back bytes re-sent when a line is cut mid-way
                                   (>= max(k,31)-1: forward window and the 30-base
                                   reverse-complement register, QuicKmer.c:415-416) */

/* Byte positions (in buf) of run positions 65536*m of every N-free run: where
 * cur_chars (a uint16_t, QuicKmer.c:402, :410) wraps to 0 and the next k-1
 * windows are lost too (:418).  Only lines of >= 65536 characters can contain
 * one; such a line always covers a whole aligned 32 KiB block, so probing each
 * block for a '\n' finds them at almost no cost on short-read input. */
inline void find_wraps(const char *buf, size_t n, std::vector<uint64_t> &out)
{
    const size_t B = 32768;
    size_t blk = 0;
    while ((blk + 1) * B <= n) {
        if (memchr(buf + blk * B, '\n', B)) { blk++; continue; }
        const size_t pos = blk * B;
        const char *prev = pos ? (const char *)memrchr(buf, '\n', pos) : nullptr;
        const size_t ls = prev ? (size_t)(prev - buf) + 1 : 0;
        const char *nx = (const char *)memchr(buf + pos, '\n', n - pos);
        const size_t le = nx ? (size_t)(nx - buf) : n;
        size_t r0 = ls;                      /* N-free runs inside the line [ls, le) */
        while (r0 < le) {
            const char *nn = (const char *)memchr(buf + r0, 'N', le - r0);
            const size_t r1 = nn ? (size_t)(nn - buf) : le;
            for (size_t m = 65536; m <= r1 - r0; m += 65536) out.push_back((uint64_t)(r0 + m - 1));
            r0 = r1 + 1;
        }
        blk = le / B;                        /* >= blk + 1: the block held no newline */
    }
}

/* find_wraps for text that arrives in pieces which may start and end in the MIDDLE of a line (qm2_shards.hip: a line longer
 * than a staging buffer): `run` = bases of the current N-free run in front of the piece, carried on to the next call. */
inline void find_wraps_stream(const char *buf, size_t n, uint64_t &run, std::vector<uint64_t> &out)
{
    size_t pos = 0;
    while (pos < n) {
        const char *nl = (const char *)memchr(buf + pos, '\n', n - pos);
        const size_t end = nl ? (size_t)(nl - buf) : n;
        size_t r0 = pos;
        while (r0 < end) {
            if (run + (end - r0) < 65536) {                  /* no wrap in the rest of this line: only the run's length is wanted */
                const char *ln = (const char *)memrchr(buf + r0, 'N', end - r0);
                run = ln ? (uint64_t)(end - (size_t)(ln - buf) - 1) : run + (end - r0);
                break;
            }
            const char *nn = (const char *)memchr(buf + r0, 'N', end - r0);
            const size_t r1 = nn ? (size_t)(nn - buf) : end;
            for (uint64_t m = (run / 65536 + 1) * 65536; m <= run + (r1 - r0); m += 65536) out.push_back((uint64_t)r0 + (m - run) - 1);
            if (nn) { run = 0; r0 = r1 + 1; } else { run += r1 - r0; r0 = r1; }
        }
        if (nl) { run = 0; pos = end + 1; } else pos = end;
    }
}

/* How much of `lines` (n bytes of whole lines left, n > 0) the next staging buffer of `cap` bytes takes, and whether that
 * piece ends with a line end: whole lines where possible; of a line longer than the buffer (or the rest of one that is
 * already open) as much as fits, up to its end. */
inline size_t next_piece(const char *lines, size_t n, size_t cap, bool line_open, bool *ends_line)
{
    size_t take = cap < n ? cap : n;
    *ends_line = true;
    if (take < n) {
        const void *nl = line_open ? nullptr : memrchr(lines, '\n', take);
        if (nl) take = (size_t)((const char *)nl - lines) + 1;
        else {
            const void *e = memchr(lines, '\n', take);
            if (e) take = (size_t)((const char *)e - lines) + 1;
            else *ends_line = false;
        }
    }
    return take;
}

/* The state of a stream of sequence lines that is staged piece by piece into batches of text (qm2_shards.hip): where the
 * uint16 run counter wraps in every piece, and — while a line is open across pieces — its last kCarry bytes with the
 * wraps among them, which become the look-back of the next batch should this one end under the line. */
struct LineStream {
    bool     open = false;          /* the last staged byte was not a line end */
    uint64_t run = 0;               /* bases of the current N-free run so far (:402's counter, unwrapped) */
    char     tail[kCarry];          /* the open line's last bytes */
    size_t   tail_len = 0;
    uint64_t abs = 0;               /* bytes staged so far */
    std::vector<uint64_t> recent;   /* wraps within the last kCarry bytes, as stream positions */

    /* the next piece; `w` receives its wrap positions relative to its first byte */
    void piece(const char *src, size_t n, bool ends_line, std::vector<uint64_t> &w)
    {
        if (!open && ends_line) find_wraps(src, n, w);         /* whole lines: the block-probing scan */
        else find_wraps_stream(src, n, run, w);
        for (uint64_t p : w) recent.push_back(abs + p);
        if (!ends_line) {
            const char *ls = (const char *)memrchr(src, '\n', n);
            const char *from = ls ? ls + 1 : src;
            const size_t m = n - (size_t)(from - src);         /* bytes of the open line in this piece */
            if (ls || !open) tail_len = 0;                     /* (it starts in this piece) */
            if (m >= kCarry) { memcpy(tail, from + m - kCarry, kCarry); tail_len = kCarry; }
            else {
                const size_t old = tail_len < kCarry - m ? tail_len : kCarry - m;
                memmove(tail, tail + tail_len - old, old);
                memcpy(tail + old, from, m);
                tail_len = old + m;
            }
            open = true;
        } else { open = false; run = 0; tail_len = 0; }
        abs += n;
        while (!recent.empty() && recent.front() + kCarry < abs) recent.erase(recent.begin());
    }
    /* a fresh batch under an open line: its look-back bytes into dst (kCarry bytes of room), the wraps among them
     * (relative to dst) appended to `wraps`; returns how many bytes (0: no line is open) */
    size_t carry_into(char *dst, std::vector<uint64_t> &wraps) const
    {
        if (!open || !tail_len) return 0;
        memcpy(dst, tail, tail_len);
        for (uint64_t w : recent) if (w >= abs - tail_len) wraps.push_back(w - (abs - tail_len));
        return tail_len;
    }
};

struct Part {
    size_t off;       /* first payload byte of this part in the caller's buffer    */
    size_t take;      /* payload bytes; 0 = "launch what is pending, then go on"   */
    size_t carry;     /* bytes before `off` staged again as look-back (0 or kCarry) */
    std::vector<uint64_t> wraps;   /* wrap positions relative to the first STAGED byte */
};

/* Cut [0, n) into parts: the first may hold at most `first_room` payload bytes
 * (what is left in the batch being filled), the others `capacity`.  Cuts fall on
 * line boundaries where possible; every part but the last fills its batch.  A
 * line longer than a whole batch is cut mid-way and the next part re-stages its
 * last kCarry bytes as look-back. */
inline void plan_parts(const char *buf, size_t n, size_t first_room, size_t capacity, std::vector<Part> &parts)
{
    std::vector<uint64_t> wraps;
    find_wraps(buf, n, wraps);
    size_t wi = 0, off = 0, carry = 0;
    bool first = true;
    while (off < n) {
        const size_t cap = first ? first_room : capacity;
        size_t take = cap < n - off ? cap : n - off;
        size_t next_carry = 0;
        if (off + take < n) {
            const char *nl = take ? (const char *)memrchr(buf + off, '\n', take) : nullptr;
            if (nl) take = (size_t)(nl - (buf + off)) + 1;
            else if (first && first_room < capacity) take = 0;   /* nothing fits behind the pending bytes */
            else next_carry = kCarry;                            /* one line longer than a batch */
        }
        Part p;
        p.off = off; p.take = take; p.carry = carry;
        /* wraps inside the staged bytes [off - carry, off + take).  (One further
         * back cannot reach this part: its k windows end within k-1 <= 30 bytes.) */
        const uint64_t lo = off - carry, hi = off + take;
        while (wi < wraps.size() && wraps[wi] < lo) wi++;
        for (size_t t = wi; t < wraps.size() && wraps[t] < hi; t++) p.wraps.push_back(wraps[t] - lo);
        parts.push_back(std::move(p));
        off += take;
        carry = next_carry;
        first = false;
    }
}

}  // namespace qm2
