/*
 * qm2_io.hip — file -> device streaming for the two big arrays of a .qm (the key table,
 * QuicKmer.c:353-359, and the chain, :483): reader threads pread 64 MiB pieces into a small
 * ring of pinned buffers, the calling thread copies each piece up as it arrives.  The order of
 * the pieces does not matter (each knows its destination), so nobody waits for a straggler.
 * Replaces malloc(H * 8) + one fread + one hipMemcpy: no 32 GiB host array, and the page-cache
 * copy (the slow part, ~6 GB/s per thread) runs on several cores while the DMA is under way.
 * The ring is made of the context's own staging buffers (idle at both times this runs: before
 * the first read is submitted and after the last one is counted): pinning and unpinning extra
 * buffers here was measured to cost the counting phase that follows 50-70 ms.
 */
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "qm2_ctx.h"

namespace {

constexpr int    kRing = 6;
constexpr int    kReaders = 4;

enum class State { Free, Filling, Full, Copying };

struct Piece {
    char      *host = nullptr;
    size_t     len = 0;
    uint64_t   index = 0;
    State      state = State::Free;
    hipEvent_t copied = nullptr;
};

struct Ring {
    std::mutex              m;
    std::condition_variable cv;
    Piece                   piece[kRing];
    uint64_t                next = 0, total = 0;      /* pieces handed to readers / in all */
    size_t                  piece_bytes = 0;
    bool                    failed = false, short_file = false;
};

void reader(Ring *r, int fd, uint64_t off, uint64_t nbytes)
{
    for (;;) {
        Piece *p = nullptr;
        {
            std::unique_lock<std::mutex> lk(r->m);
            for (;;) {
                if (r->failed || r->next >= r->total) return;
                for (auto &q : r->piece)
                    if (q.state == State::Free) { p = &q; break; }
                if (p) break;
                r->cv.wait(lk);
            }
            p->index = r->next++;
            p->state = State::Filling;
        }
        const uint64_t at = p->index * r->piece_bytes;
        const size_t want = (size_t)std::min<uint64_t>(r->piece_bytes, nbytes - at);
        size_t got = 0;
        bool bad = false;
        while (got < want) {
            const ssize_t n = pread(fd, p->host + got, want - got, (off_t)(off + at + got));
            if (n < 0) { if (errno == EINTR) continue; bad = true; break; }
            if (n == 0) break;
            got += (size_t)n;
        }
        std::lock_guard<std::mutex> lk(r->m);
        p->len = got;
        if (bad) r->failed = true;
        if (got < want) r->short_file = r->failed = true;
        p->state = State::Full;
        r->cv.notify_all();
    }
}

}  // namespace

int qm2_stream_fd(qm2_ctx *c, int fd, uint64_t off, void *dst, uint64_t nbytes, const char *what)
{
    if (nbytes == 0) return 0;
    Ring r;
    r.piece_bytes = c->staging_bytes;
    r.total = (nbytes + r.piece_bytes - 1) / r.piece_bytes;
    const int n_bufs = (int)std::min<uint64_t>(std::min<size_t>(kRing, c->slots.size()), r.total);
    int rc = 0;
    for (int i = 0; i < n_bufs && !rc; i++) {
        qm2::Slot &slot = c->slots[i];
        if (slot.lent) rc = qm2_fail("qm2: staging buffers are held while %s is loaded", what);
        else if (qm2_slot_ready(c, slot)) rc = 1;
        else if (hipEventCreateWithFlags(&r.piece[i].copied, hipEventDisableTiming) != hipSuccess)
            rc = qm2_fail("Memory allocation failed");
        r.piece[i].host = slot.host;
    }
    for (int i = n_bufs; i < kRing; i++) r.piece[i].state = State::Copying;   /* never used: neither Free nor Full */
    std::vector<std::thread> readers;
    if (!rc)
        for (int t = 0; t < std::min<int>(kReaders, n_bufs); t++) readers.emplace_back(reader, &r, fd, off, nbytes);
    uint64_t done = 0;
    while (!rc && done < r.total) {
        Piece *p = nullptr;
        {
            std::unique_lock<std::mutex> lk(r.m);
            for (;;) {
                if (r.failed) break;
                for (int i = 0; i < n_bufs; i++)
                    if (r.piece[i].state == State::Full) { p = &r.piece[i]; break; }
                if (p) break;
                /* nothing to copy: retire finished copies, or wait for a reader */
                bool retired = false;
                for (int i = 0; i < n_bufs; i++)
                    if (r.piece[i].state == State::Copying && hipEventQuery(r.piece[i].copied) == hipSuccess) {
                        r.piece[i].state = State::Free;
                        retired = true;
                        done++;
                    }
                if (retired) { r.cv.notify_all(); if (done >= r.total) break; continue; }
                bool copying = false;
                for (int i = 0; i < n_bufs; i++) copying |= r.piece[i].state == State::Copying;
                if (copying) { lk.unlock(); std::this_thread::yield(); lk.lock(); }
                else r.cv.wait(lk);
            }
            if (p) p->state = State::Copying;
        }
        if (!p) break;
        if (hipMemcpyAsync((char *)dst + p->index * r.piece_bytes, p->host, p->len, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess ||
            hipEventRecord(p->copied, c->copy_stream) != hipSuccess) {
            rc = qm2_fail("qm2: upload of %s failed", what);
            std::lock_guard<std::mutex> lk(r.m);
            r.failed = true;
            r.cv.notify_all();
        }
    }
    {
        std::lock_guard<std::mutex> lk(r.m);
        if (rc) r.failed = true;
        r.cv.notify_all();
    }
    for (auto &t : readers) t.join();
    (void)hipStreamSynchronize(c->copy_stream);
    if (!rc && r.failed) rc = qm2_fail(r.short_file ? "%s is truncated" : "Cannot read %s", what);
    for (int i = 0; i < n_bufs; i++)
        if (r.piece[i].copied) (void)hipEventDestroy(r.piece[i].copied);
    return rc;
}
