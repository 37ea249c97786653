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
