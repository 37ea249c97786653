/*
 * qm2_shards.hip — BASELINE config 4 driven from C (include/qm2_shards.h): hash-range shards of the dictionary on the
 * GPUs of one node, k-mers routed to their owners by RCCL over xGMI.  libqm2_shards.so = this file + librccl on top of
 * the C-ABI of libqm2_hip.so (qm2_route_device, qm2_probe_slabs_device, qm2_merge_counters); no kernel of its own.
 *
 * The reference has one table and N probing threads in one address space (QuicKmer.c:256-296, :368-384); here every
 * GPU holds a slice of it and a batch's k-mers travel to the slice that owns their home slot (:92).  Single process,
 * single host thread, one communicator per GPU (ncclCommInitAll) — the host knows every rank's slab fills, so only the
 * filled parts of the slabs go over the wire, in one ncclGroupStart / ncclGroupEnd per round.
 *
 * EXPERIMENTAL for more than one GPU — NOT YET RUN ON TWO PHYSICAL GPUs (no lease so far had more than one): the round
 * logic, the dealing, the overflow re-route and the fold are exercised on one GPU with QM2_SHARDS_FAKE_EXCHANGE=1 (the
 * ncclSend / ncclRecv pairs replaced by device-to-device copies, G = 2 / 4 logical ranks on one device) and with G = 1;
 * the RCCL calls themselves only by a G = 1 run.  Until a run on two cards is on record, qm2_shards_open refuses G > 1
 * over RCCL unless QM2_SHARDS_EXPERIMENTAL=1 is set (the CLI's help says the same).
 *
 * Dealing: the reader's commits (whole sequence lines, at most a staging buffer each) go to the ranks ROUND-ROBIN, so a
 * small input is routed by all GPUs, not by the first one alone; when the rank whose turn it is cannot take a commit,
 * the round runs with what every rank holds.  A line longer than a staging buffer (a chromosome per line) arrives
 * through qm2_shards_submit, is cut into staging-sized pieces that stay on ONE rank, back to back — cuts of a round only
 * ever fall on line ends — and, if the rank's batch fills in the middle of it, goes on behind a round with its last
 * bytes as look-back (qm2_route_device_carry), like a line that qm2_submit has to cut.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/qm2_shards.h"
#include "qm2_plan.h"

namespace {

int fail(const char *fmt, const char *a = "", const char *b = "")
{
    printf(fmt, a, b);
    putchar('\n');
    fflush(stdout);
    return 1;
}

#define S_OK(call)                                                                       \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) return fail("qm2 shards: %s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define N_OK(call)                                                                       \
    do {                                                                                 \
        ncclResult_t e_ = (call);                                                        \
        if (e_ != ncclSuccess) return fail("qm2 shards: %s failed: %s", #call, ncclGetErrorString(e_)); \
    } while (0)

struct Slot {
    char      *host = nullptr;
    hipEvent_t done = nullptr;          /* its H2D copy has finished */
    bool       busy = false, lent = false;
};

struct ExSet {                          /* one set of exchange buffers of a rank */
    uint64_t *slabs = nullptr, *recv = nullptr;       /* G x cap k-mers each */
    uint32_t *fill = nullptr, *incoming = nullptr;    /* G */
    unsigned long long *call = nullptr;               /* {bases, k-mers, k-mers that fitted nowhere} of the last route */
    hipEvent_t probed = nullptr;        /* the probe that read this set has finished */
    bool       busy = false;
};

struct Rank {
    int         dev = 0;
    hipStream_t s_route = nullptr, s_probe = nullptr, s_copy = nullptr;
    qm2_ctx    *rctx = nullptr, *ctx = nullptr;
    uint64_t    own_lo = 0, own_hi = 0, lo = 0, hi = 0;
    uint8_t    *text = nullptr;         /* this round's sequence lines */
    size_t      fill = 0;
    size_t      carry0 = 0;             /* the first bytes of `text` are the look-back of a line cut by the previous round */
    std::vector<size_t> cuts;           /* line ends at the ends of commits in `text`: where a round may be cut */
    std::vector<uint64_t> wraps;        /* where the reference's uint16 run counter wraps (:402; lines of >= 65536 bases), in `text` */
    uint64_t   *d_wraps = nullptr;
    size_t      wraps_cap = 0;
    hipEvent_t  copied = nullptr, routed = nullptr;
    ExSet       set[2];
    ncclComm_t  comm = nullptr;
    unsigned long long *call_host = nullptr;          /* pinned: 3 */
    uint32_t   *fill_host = nullptr, *incoming_host = nullptr;   /* pinned: G each */
};

}  // namespace

struct qm2_shards {
    int       G = 0, dict_fd = -1;
    uint64_t  H = 0;
    uint32_t  k = 30;
    size_t    batch_bytes = 0, staging_bytes = (size_t)64 << 20, cap = 0;
    bool      fake = false, comms = false;
    /* The exchange layout (QM2_SHARDS_ROUTE = xcd | flat).
     *   flat  one slab per owner (qm2_route_device); only the filled part of a slab travels; the receiver splits what it
     *         got twice (qm2_probe_slabs_device) — rounds 1-4's layout, and the fall-back for tables below 2^23 slots.
     *   xcd   the senders' 10-bit route (qm2_route_xcd_device): 1024 regions = G owners x n_subs sub-slabs, each in 8
     *         per-XCD sub-regions of a fixed capacity `cap`, x-major; an owner's part of it — 8 blocks of n_subs x cap
     *         k-mers per sender — travels whole, fills alongside, and IS the first level of the receiver's own partition
     *         (qm2_probe_subslabs_device: one split instead of two; 8 logical shards on one GPU: 94 % of a replica's
     *         rate against 78 %, profiles/r04_c3/bench_sharded_8_logical_20M.json).  Which of the two the wire
     *         prefers — whole blocks with ~13 % of room in them, or filled parts only — is for the first run on more
     *         than one GPU to measure; both are exact. */
    bool      xcd = false, xcd_now = false;   /* xcd_now / cap_now: the round being run (round_range's fall-back to one slab per owner) */
    size_t    cap_now = 0, cap_flat = 0;
    uint32_t  n_subs = 1;              /* xcd: sub-slabs per owner (1024 / G) */
    size_t    n_fill = 0;              /* fill entries of a rank's route: G (flat) or 8 x 1024 (xcd) */
    std::vector<Rank> r;
    std::vector<Slot> slots;
    uint32_t  next_slot = 0;
    int       feed = 0;                 /* the rank whose turn it is (round-robin by commit) */
    int       turn = 0;                 /* exchange set of the next round */
    qm2::LineStream ls;             /* an over-long line across staged pieces: run-counter wraps, look-back */
    qm2_ctx  *result = nullptr;
    unsigned long long n_bases = 0, n_kmers = 0;
    qm2_shards_info info = {0, 0, 0, 0, 0};
};

/* qm2_shard_span's rule with the keys read from the file: the owned range extended past its middle-facing edge over
 * the run of occupied slots, the empty slot that ends it included, in whole 64-slot blocks */
static int span_from_fd(int fd, uint64_t H, uint32_t G, uint32_t g, uint64_t *own_lo, uint64_t *own_hi, uint64_t *slo, uint64_t *shi)
{
    const uint64_t per = H / G, lo = (uint64_t)g * per, hi = lo + per;
    uint64_t a = lo, b = hi;
    auto key_at = [&](uint64_t slot, uint64_t *v) {
        return pread(fd, v, sizeof *v, (off_t)(24 + slot * sizeof(uint64_t))) == (ssize_t)sizeof *v;
    };
    if (G > 1) {
        uint64_t v = 0;
        if (lo < (H >> 1)) {
            while (b < H) { if (!key_at(b - 1, &v)) return fail("Dictionary file is truncated"); if (v == 0) break; b++; }
        } else {
            while (a > 0) { if (!key_at(a, &v)) return fail("Dictionary file is truncated"); if (v == 0) break; a--; }
        }
        if (H >= 64) { a &= ~(uint64_t)63; b = (b + 63) & ~(uint64_t)63; }
    }
    *own_lo = lo; *own_hi = hi; *slo = a; *shi = b;
    return 0;
}

extern "C" int qm2_shards_open(qm2_shards **out, int dict_fd, uint64_t hash_size, uint32_t k, const int *devices,
                               int n_devices, uint64_t batch_bytes, uint32_t n_staging)
{
    if (!out || !devices) return fail("qm2_shards_open: null argument");
    *out = nullptr;
    if (n_devices < 1 || (n_devices & (n_devices - 1)) || (uint64_t)n_devices > hash_size / 64)
        return fail("qm2_shards_open: the number of shards must be a power of two (and a shard at least 64 slots)");
    qm2_shards *s = new qm2_shards();
    s->G = n_devices;
    s->dict_fd = dict_fd;
    s->H = hash_size;
    s->k = k;
    s->fake = getenv("QM2_SHARDS_FAKE_EXCHANGE") != nullptr;
    if (n_devices > 1 && !s->fake && !getenv("QM2_SHARDS_EXPERIMENTAL")) {
        delete s;
        return fail("--shards on more than one GPU is experimental: its RCCL exchange has not been run on two physical GPUs yet "
                    "(set QM2_SHARDS_EXPERIMENTAL=1 to use it; --devices a,b,.. is the tested multi-GPU path)");
    }
    for (int i = 0; i < n_devices && !s->fake; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) {
                delete s;
                return fail("qm2_shards_open: a device is listed twice (RCCL wants one rank per GPU)");
            }
    const uint64_t per = hash_size / (uint64_t)n_devices;
    s->batch_bytes = batch_bytes ? (size_t)batch_bytes
                                 : (size_t)std::min<uint64_t>(std::max<uint64_t>(per + per / 2, 1ull << 30), 6ull << 30);
    if (s->batch_bytes < 2 * s->staging_bytes) s->staging_bytes = std::max<size_t>(s->batch_bytes / 2, 4096);
    {
        /* the 10-bit route needs a shard whose own range leaves a level to split behind the sub-slabs (2^12-slot buckets,
         * 10 routed bits: tables of 2^23 slots and more) and at most 1024 / 2 owners */
        const char *mode = getenv("QM2_SHARDS_ROUTE");
        const bool can = hash_size >= (1ull << 23) && n_devices <= 512 && k <= 30;
        if (mode && !strcmp(mode, "xcd") && !can) { delete s; return fail("QM2_SHARDS_ROUTE=xcd needs a table of at least 2^23 slots (and k <= 30)"); }
        s->xcd = can && !(mode && !strcmp(mode, "flat"));
        s->n_subs = s->xcd ? 1024u / (uint32_t)n_devices : 1u;
        s->n_fill = s->xcd ? (size_t)8 * 1024 : (size_t)n_devices;
    }
    if (s->xcd) {
        /* a sub-region takes its share of a batch's k-mers (at most one per byte) with 12 % head-room and the spread of
         * uniform hashing — what qm2_prepare gives the text level's own sub-regions; a round that overflows one is cut
         * and routed again like a flat one */
        const double mean = (double)s->batch_bytes / (8.0 * 1024.0);
        s->cap = ((size_t)(mean * 1.12 + 8.0 * std::sqrt(mean + 1.0)) + 256) & ~(size_t)1;
    } else
    /* a slab takes a rank's share of one batch with 30 % head-room (uniform hashing spreads the k-mers evenly) and at
     * least one whole commit, so that cutting a round — at commit ends, then rank by rank — always ends in pieces that fit */
    s->cap = (std::max<size_t>((size_t)((double)s->batch_bytes / n_devices * 1.3), s->staging_bytes) + 4096) & ~(size_t)1;
    s->xcd_now = s->xcd;
    s->cap_now = s->cap;
    s->info.ten_bit_route = s->xcd ? 1u : 0u;
    /* (one slab per owner, a whole commit wide: what a piece falls back to that overflows the 10-bit route however it is cut) */
    s->cap_flat = (s->staging_bytes + 4096) & ~(size_t)1;
    s->r.resize((size_t)n_devices);
    *out = s;                            /* from here on qm2_shards_close cleans up */
    for (int g = 0; g < n_devices; g++) {
        Rank &R = s->r[(size_t)g];
        R.dev = devices[g];
        S_OK(hipSetDevice(R.dev));
        S_OK(hipStreamCreateWithFlags(&R.s_route, hipStreamNonBlocking));
        S_OK(hipStreamCreateWithFlags(&R.s_probe, hipStreamNonBlocking));
        S_OK(hipStreamCreateWithFlags(&R.s_copy, hipStreamNonBlocking));
        S_OK(hipEventCreateWithFlags(&R.copied, hipEventDisableTiming));
        S_OK(hipEventCreateWithFlags(&R.routed, hipEventDisableTiming));
        if (span_from_fd(dict_fd, hash_size, (uint32_t)n_devices, (uint32_t)g, &R.own_lo, &R.own_hi, &R.lo, &R.hi)) return 1;
        qm2_opts o;
        memset(&o, 0, sizeof o);
        o.struct_size = sizeof o;
        o.device = R.dev;
        o.flags = QM2_FLAG_KEYS_LATER;
        o.span_lo = 0; o.span_hi = std::min<uint64_t>(64, hash_size);
        o.stream = R.s_route;
        if (qm2_open(&R.rctx, nullptr, hash_size, k, &o)) return 1;      /* encodes and routes: no table */
        o.span_lo = R.lo; o.span_hi = R.hi; o.own_lo = R.own_lo; o.own_hi = R.own_hi;
        o.stream = R.s_probe;
        if (qm2_open(&R.ctx, nullptr, hash_size, k, &o)) return 1;
        if (qm2_keys_from_fd(R.ctx, dict_fd, 24)) return 1;
        S_OK(hipSetDevice(R.dev));
        S_OK(hipMalloc((void **)&R.text, s->batch_bytes + 64));
        for (ExSet &e : R.set) {
            /* (xcd: 8 x 1024 regions out, 8 G "peers" x n_subs sub-slabs in — the same number either way) */
            const size_t n_regions = s->xcd ? (size_t)8 * 1024 : (size_t)n_devices;
            const size_t elems = std::max(n_regions * s->cap, s->xcd ? (size_t)n_devices * s->cap_flat : (size_t)0);
            S_OK(hipMalloc((void **)&e.slabs, elems * sizeof(uint64_t)));
            if (n_devices > 1) S_OK(hipMalloc((void **)&e.recv, elems * sizeof(uint64_t)));
            S_OK(hipMalloc((void **)&e.fill, s->n_fill * sizeof(uint32_t)));
            S_OK(hipMalloc((void **)&e.incoming, s->n_fill * sizeof(uint32_t)));
            S_OK(hipMalloc((void **)&e.call, 3 * sizeof(unsigned long long)));
            S_OK(hipEventCreateWithFlags(&e.probed, hipEventDisableTiming));
        }
        S_OK(hipHostMalloc((void **)&R.call_host, 3 * sizeof(unsigned long long), hipHostMallocPortable));
        S_OK(hipHostMalloc((void **)&R.fill_host, s->n_fill * sizeof(uint32_t), hipHostMallocPortable));
        S_OK(hipHostMalloc((void **)&R.incoming_host, s->n_fill * sizeof(uint32_t), hipHostMallocPortable));
        /* the shard's partition buffers, ahead of the first round (xcd: the plan behind the sub-slabs is made by the first
         * qm2_probe_subslabs_device, with the k-mers that really arrived as its bound) */
        if (!s->xcd && qm2_prepare(R.ctx, (uint64_t)s->cap * (uint64_t)n_devices, 0)) return 1;
    }
    if (n_devices > 1 && !s->fake) {
        std::vector<ncclComm_t> comms((size_t)n_devices);
        N_OK(ncclCommInitAll(comms.data(), n_devices, devices));
        for (int g = 0; g < n_devices; g++) s->r[(size_t)g].comm = comms[(size_t)g];
        s->comms = true;
    }
    s->slots.resize(std::max<uint32_t>(n_staging, 3));
    for (Slot &sl : s->slots) {
        S_OK(hipHostMalloc((void **)&sl.host, s->staging_bytes, hipHostMallocPortable));
        S_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    return 0;
}

/* One round over the bytes [lo[g], hi[g]) of every rank's text.  Returns 0, or 1 after a message. */
static int round_range(qm2_shards *s, const std::vector<size_t> &lo, const std::vector<size_t> &hi, int depth)
{
    const int G = s->G;
    const int w = s->turn;
    /* (the layout of THIS round: the context's, or — a piece that overflows the 10-bit route's small regions however it is
     * cut — one slab per owner in the same buffers, see below) */
    const bool xcd = s->xcd_now;
    const size_t cap = s->cap_now, n_fill = xcd ? s->n_fill : (size_t)G;
    bool any = false;
    for (int g = 0; g < G; g++) any = any || hi[(size_t)g] > lo[(size_t)g];
    if (!any) return 0;
    /* route: text -> slabs by owner, on every rank's router stream.  The probe of two rounds ago read this set: the
     * router stream waits for it before anything overwrites the set (RCCL: a rank's receive buffer is written by its
     * own ncclRecv, on its own router stream; rehearsal: by the SENDERS' copies, so every sender waits for every probe) */
    for (int g = 0; g < G; g++) {
        Rank &R = s->r[(size_t)g];
        S_OK(hipSetDevice(R.dev));
        for (int q = 0; q < G; q++)
            if ((q == g || s->fake) && s->r[(size_t)q].set[w].busy) S_OK(hipStreamWaitEvent(R.s_route, s->r[(size_t)q].set[w].probed, 0));
    }
    for (int g = 0; g < G; g++) {
        Rank &R = s->r[(size_t)g];
        ExSet &e = R.set[w];
        e.busy = false;
        S_OK(hipSetDevice(R.dev));
        S_OK(hipStreamWaitEvent(R.s_route, R.copied, 0));
        S_OK(hipMemsetAsync(e.fill, 0, n_fill * sizeof(uint32_t), R.s_route));
        S_OK(hipMemsetAsync(e.call, 0, 3 * sizeof(unsigned long long), R.s_route));
        if (hi[(size_t)g] > lo[(size_t)g]) {
            /* the piece's wrap positions, relative to its first byte (long lines are rare: a small blocking copy) */
            std::vector<uint64_t> sup;
            for (uint64_t p : R.wraps) if (p >= lo[(size_t)g] && p < hi[(size_t)g]) sup.push_back(p - lo[(size_t)g]);
            if (sup.size() > R.wraps_cap) {
                if (R.d_wraps) S_OK(hipFree(R.d_wraps));
                R.wraps_cap = std::max<size_t>(sup.size(), 1024);
                S_OK(hipMalloc((void **)&R.d_wraps, R.wraps_cap * sizeof(uint64_t)));
            }
            if (!sup.empty()) {
                S_OK(hipStreamSynchronize(R.s_route));             /* (an earlier route may still read the list) */
                S_OK(hipMemcpy(R.d_wraps, sup.data(), sup.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
            }
            /* (a piece that starts at the text's first byte may start with the look-back of a cut line) */
            const uint32_t carry = lo[(size_t)g] == 0 ? (uint32_t)R.carry0 : 0u;
            if (xcd ? qm2_route_xcd_device_carry(R.rctx, R.text + lo[(size_t)g], hi[(size_t)g] - lo[(size_t)g], carry, sup.empty() ? nullptr : R.d_wraps,
                                                    sup.size(), e.slabs, cap, 1024u, e.fill, (uint64_t *)e.call)
                       : qm2_route_device_carry(R.rctx, R.text + lo[(size_t)g], hi[(size_t)g] - lo[(size_t)g], carry,
                                                sup.empty() ? nullptr : R.d_wraps, sup.size(), e.slabs, cap, (uint32_t)G, e.fill, (uint64_t *)e.call))
                return 1;
        }
        S_OK(hipSetDevice(R.dev));
        S_OK(hipMemcpyAsync(R.call_host, e.call, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.s_route));
        S_OK(hipMemcpyAsync(R.fill_host, e.fill, n_fill * sizeof(uint32_t), hipMemcpyDeviceToHost, R.s_route));
    }
    bool unfit = false;
    for (int g = 0; g < G; g++) {
        Rank &R = s->r[(size_t)g];
        S_OK(hipSetDevice(R.dev));
        S_OK(hipStreamSynchronize(R.s_route));                 /* the one word a round makes the host wait for */
        unfit = unfit || R.call_host[2] != 0;
    }
    if (unfit) {
        /* skew (poly-A sends everything to one owner): cut every rank's piece at the commit end nearest its middle
         * and route the halves again.  A single commit always fits a slab (cap >= a staging buffer), so this ends. */
        if (depth > 48) return fail("qm2 shards: a round keeps overflowing its slabs");
        s->info.rerouted++;
        std::vector<size_t> mid((size_t)G);
        bool cut_somewhere = false;
        for (int g = 0; g < G; g++) {
            const Rank &R = s->r[(size_t)g];
            const size_t a = lo[(size_t)g], b = hi[(size_t)g], want = a + (b - a) / 2;
            size_t best = b;                                   /* no interior cut: everything stays in the first half */
            for (size_t c : R.cuts)
                if (c > a && c < b && (best == b || (c > want ? c - want : want - c) < (best > want ? best - want : want - best))) best = c;
            mid[(size_t)g] = best;
            cut_somewhere = cut_somewhere || best != b;
        }
        if (!cut_somewhere) {
            /* every rank is down to one commit and together they still overflow an owner's slabs (G commits of the same
             * k-mer): half of the ranks now, the others in a round of their own; one rank's one commit always fits */
            std::vector<int> live;
            for (int g = 0; g < G; g++) if (hi[(size_t)g] > lo[(size_t)g]) live.push_back(g);
            if (live.size() < 2 && xcd) {
                /* one commit of one rank, and a sub-region of the 10-bit route still overflows (a flood of one k-mer: poly-A):
                 * this piece goes round with ONE slab per owner — in the same buffers, each slab a whole commit wide, so it
                 * fits whatever it holds — and is split twice on the other side; the rounds behind it route by 10 bits again */
                s->xcd_now = false;
                s->cap_now = s->cap_flat;
                s->info.flat_fallbacks++;
                const int rc = round_range(s, lo, hi, depth + 1);
                s->xcd_now = true;
                s->cap_now = s->cap;
                return rc;
            }
            if (live.size() < 2) return fail("qm2 shards: the k-mers between two line ends do not fit an owner's slab (one line of a single repeated k-mer, longer than a rank's share of a batch?)");
            std::vector<size_t> hi_a(hi), lo_b(lo);
            for (size_t i = 0; i < live.size(); i++) {
                const size_t g = (size_t)live[i];
                if (i < live.size() / 2) lo_b[g] = hi[g];     /* first half: only in round a */
                else hi_a[g] = lo[g];                         /* second half: only in round b */
            }
            if (round_range(s, lo, hi_a, depth + 1)) return 1;
            return round_range(s, lo_b, hi, depth + 1);
        }
        if (round_range(s, lo, mid, depth + 1)) return 1;
        return round_range(s, mid, hi, depth + 1);
    }
    s->info.rounds++;
    for (int g = 0; g < G; g++) { s->n_bases += s->r[(size_t)g].call_host[0]; s->n_kmers += s->r[(size_t)g].call_host[1]; }
    /* exchange: slab p of rank g -> slot g of rank p's receive buffer; the fills are known here, so only what is
     * filled travels and the receivers get their counts from the host */
    if (G == 1) {
        Rank &R = s->r[0];
        S_OK(hipSetDevice(R.dev));
        S_OK(hipEventRecord(R.routed, R.s_route));
        S_OK(hipStreamWaitEvent(R.s_probe, R.routed, 0));
        if (xcd) {
            /* what was routed is what is probed: 8 "peers" (the router's XCD sub-regions) of 1024 sub-slabs, 1024 regions apart */
            uint64_t bound = 0;
            for (size_t i = 0; i < s->n_fill; i++) bound += std::min<uint64_t>(R.fill_host[i], cap);
            if (qm2_probe_subslabs_device(R.ctx, R.set[w].slabs, cap, 8, 1024, 1024, R.set[w].fill, bound + 64)) return 1;
        } else if (qm2_probe_slabs_device(R.ctx, R.set[w].slabs, cap, 1, R.set[w].fill)) return 1;
    } else if (xcd) {
        /* exchange, 10-bit route: sender g's region (x, p * n_subs + sub) -> receiver p's sub-slab ((x * G + g), sub): for
         * every (sender, receiver, XCD) one block of n_subs x cap k-mers, whole; the fills travel through the host, which
         * has them anyway (the fit verdict) */
        const size_t ns = s->n_subs, blk = ns * cap;
        std::vector<uint64_t> bound((size_t)G, 0);
        for (int p = 0; p < G; p++)
            for (int g = 0; g < G; g++)
                for (int x = 0; x < 8; x++)
                    for (size_t q = 0; q < ns; q++) {
                        const uint32_t f = s->r[(size_t)g].fill_host[(size_t)x * 1024 + (size_t)p * ns + q];
                        s->r[(size_t)p].incoming_host[((size_t)x * (size_t)G + (size_t)g) * ns + q] = f;
                        bound[(size_t)p] += std::min<uint64_t>(f, cap);
                    }
        if (!s->fake) N_OK(ncclGroupStart());
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            ExSet &e = R.set[w];
            S_OK(hipSetDevice(R.dev));
            S_OK(hipMemcpyAsync(e.incoming, R.incoming_host, n_fill * sizeof(uint32_t), hipMemcpyHostToDevice, R.s_route));
            for (int p = 0; p < G; p++)
                for (int x = 0; x < 8; x++) {
                    const uint64_t *out = e.slabs + ((size_t)x * 1024 + (size_t)p * ns) * cap;
                    if (p == g || s->fake) {
                        S_OK(hipMemcpyAsync(s->r[(size_t)p].set[w].recv + ((size_t)x * (size_t)G + (size_t)g) * blk, out, blk * sizeof(uint64_t),
                                            hipMemcpyDeviceToDevice, R.s_route));
                        continue;
                    }
                    N_OK(ncclSend(out, blk, ncclUint64, p, R.comm, R.s_route));
                    N_OK(ncclRecv(e.recv + ((size_t)x * (size_t)G + (size_t)p) * blk, blk, ncclUint64, p, R.comm, R.s_route));
                    s->info.exchanged_bytes += blk * sizeof(uint64_t);
                }
        }
        if (!s->fake) N_OK(ncclGroupEnd());
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            S_OK(hipSetDevice(R.dev));
            S_OK(hipEventRecord(R.routed, R.s_route));
        }
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            S_OK(hipSetDevice(R.dev));
            for (int q = 0; q < G; q++)
                if (q == g || s->fake) S_OK(hipStreamWaitEvent(R.s_probe, s->r[(size_t)q].routed, 0));
            /* 8 G "peers" (sender x XCD) of n_subs sub-slabs, side by side: the first level of this shard's own partition */
            if (qm2_probe_subslabs_device(R.ctx, R.set[w].recv, cap, 8u * (uint32_t)G, (uint32_t)ns, 0, R.set[w].incoming, bound[(size_t)g] + 64)) return 1;
        }
    } else {
        for (int p = 0; p < G; p++)
            for (int g = 0; g < G; g++) s->r[(size_t)p].incoming_host[g] = s->r[(size_t)g].fill_host[p];
        if (!s->fake) N_OK(ncclGroupStart());
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            ExSet &e = R.set[w];
            S_OK(hipSetDevice(R.dev));
            S_OK(hipMemcpyAsync(e.incoming, R.incoming_host, (size_t)G * sizeof(uint32_t), hipMemcpyHostToDevice, R.s_route));
            for (int p = 0; p < G; p++) {
                const size_t n_out = R.fill_host[p], n_in = s->r[(size_t)p].fill_host[g];
                if (p == g || s->fake) {
                    /* own slab (or, in the one-GPU rehearsal, every slab): a device-to-device copy into the receiver's buffer */
                    if (n_out)
                        S_OK(hipMemcpyAsync(s->r[(size_t)p].set[w].recv + (size_t)g * cap, e.slabs + (size_t)p * cap, n_out * sizeof(uint64_t),
                                            hipMemcpyDeviceToDevice, R.s_route));
                    continue;
                }
                if (n_out) N_OK(ncclSend(e.slabs + (size_t)p * cap, n_out, ncclUint64, p, R.comm, R.s_route));
                if (n_in) N_OK(ncclRecv(e.recv + (size_t)p * cap, n_in, ncclUint64, p, R.comm, R.s_route));
                s->info.exchanged_bytes += n_out * sizeof(uint64_t);
            }
        }
        if (!s->fake) N_OK(ncclGroupEnd());
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            S_OK(hipSetDevice(R.dev));
            S_OK(hipEventRecord(R.routed, R.s_route));
        }
        /* probe what arrived: the prober stream waits for its own rank's exchange (RCCL: the receives are on its
         * router stream; rehearsal: the senders' copies are on THEIR router streams, so for all of them) */
        for (int g = 0; g < G; g++) {
            Rank &R = s->r[(size_t)g];
            S_OK(hipSetDevice(R.dev));
            for (int q = 0; q < G; q++)
                if (q == g || s->fake) S_OK(hipStreamWaitEvent(R.s_probe, s->r[(size_t)q].routed, 0));
            if (qm2_probe_slabs_device(R.ctx, R.set[w].recv, cap, (uint32_t)G, R.set[w].incoming)) return 1;
        }
    }
    for (int g = 0; g < G; g++) {
        Rank &R = s->r[(size_t)g];
        S_OK(hipSetDevice(R.dev));
        S_OK(hipEventRecord(R.set[w].probed, R.s_probe));
        R.set[w].busy = true;
    }
    s->turn ^= 1;
    return 0;
}

/* all the text collected so far */
static int run_round(qm2_shards *s)
{
    std::vector<size_t> lo((size_t)s->G, 0), hi((size_t)s->G);
    for (int g = 0; g < s->G; g++) {
        Rank &R = s->r[(size_t)g];
        hi[(size_t)g] = R.fill;
        S_OK(hipSetDevice(R.dev));
        S_OK(hipEventRecord(R.copied, R.s_copy));
    }
    if (round_range(s, lo, hi, 0)) return 1;
    /* (the routes have finished — the host waited for their verdict — so the text buffers are free again) */
    for (Rank &R : s->r) { R.fill = 0; R.carry0 = 0; R.cuts.clear(); R.wraps.clear(); }
    s->feed = 0;
    return 0;
}

static int slot_ready(Slot &sl)
{
    if (sl.busy) { S_OK(hipEventSynchronize(sl.done)); sl.busy = false; }
    return 0;
}

extern "C" int qm2_shards_acquire(qm2_shards *s, char **buf, size_t *capacity)
{
    if (!s || !buf || !capacity) return fail("qm2_shards_acquire: null argument");
    for (size_t tries = 0; tries < s->slots.size(); tries++) {
        Slot &sl = s->slots[s->next_slot];
        s->next_slot = (s->next_slot + 1) % (uint32_t)s->slots.size();
        if (sl.lent) continue;
        if (slot_ready(sl)) return 1;
        sl.lent = true;
        *buf = sl.host;
        *capacity = s->staging_bytes;
        return 0;
    }
    return fail("qm2_shards_acquire: every staging buffer is held");
}

/* n bytes at `src` (pinned: `owner`) into the batch of the rank whose turn it is.  ends_line: the last byte is a line
 * end (every commit of the reader; the last piece of an over-long line) — only there may a round be cut, and only then
 * does the turn pass on. */
static int stage(qm2_shards *s, const char *src, size_t n, Slot *owner, bool ends_line)
{
    if (n > s->staging_bytes) return fail("qm2 shards: more bytes than a staging buffer holds");
    if (s->r[(size_t)s->feed].fill + n > s->batch_bytes) {
        /* this rank's batch is full: the round runs with what every rank holds (they were dealt the same amounts) */
        if (run_round(s)) return 1;
        /* ... under a line: it goes on in the empty batch behind its last bytes, which only serve as look-back */
        Rank &R0 = s->r[(size_t)s->feed];
        char look[qm2::kCarry];
        const size_t carry = s->ls.carry_into(look, R0.wraps);
        if (carry) {
            S_OK(hipSetDevice(R0.dev));
            S_OK(hipMemcpy(R0.text, look, carry, hipMemcpyHostToDevice));    /* (rare, 32 bytes; the text is free: run_round waited for the routes) */
            R0.fill = R0.carry0 = carry;
        }
    }
    Rank &R = s->r[(size_t)s->feed];
    S_OK(hipSetDevice(R.dev));
    S_OK(hipMemcpyAsync(R.text + R.fill, src, n, hipMemcpyHostToDevice, R.s_copy));
    S_OK(hipEventRecord(owner->done, R.s_copy));
    owner->busy = true;
    {
        std::vector<uint64_t> w;
        s->ls.piece(src, n, ends_line, w);
        for (uint64_t p : w) R.wraps.push_back(p + R.fill);
    }
    R.fill += n;
    if (ends_line) {
        R.cuts.push_back(R.fill);
        s->feed = (s->feed + 1) % s->G;
    }
    return 0;
}

extern "C" int qm2_shards_commit(qm2_shards *s, char *buf, size_t n)
{
    if (!s || !buf) return fail("qm2_shards_commit: null argument");
    Slot *owner = nullptr;
    for (Slot &sl : s->slots) if (sl.lent && sl.host == buf) owner = &sl;
    if (!owner) return fail("qm2_shards_commit: not a buffer from qm2_shards_acquire");
    owner->lent = false;
    if (n == 0) return 0;
    if (buf[n - 1] != '\n') return fail("qm2_shards_commit: buffer must end with a newline (whole sequence lines only)");
    if (s->ls.open) return fail("qm2_shards_commit: a line is still open (qm2_shards_submit)");   /* (cannot happen: submit closes what it opens) */
    return stage(s, buf, n, owner, true);
}

extern "C" int qm2_shards_submit(qm2_shards *s, const char *lines, size_t n)
{
    if (!s || (!lines && n)) return fail("qm2_shards_submit: null argument");
    if (n && lines[n - 1] != '\n') return fail("qm2_shards_submit: buffer must end with a newline (whole sequence lines only)");
    /* a pageable buffer: through the staging ring, cut at line ends; a line longer than a staging buffer in pieces that
     * stay on one rank (stage) */
    size_t at = 0;
    while (at < n) {
        char *buf;
        size_t cap;
        if (qm2_shards_acquire(s, &buf, &cap)) return 1;
        Slot *owner = nullptr;
        for (Slot &sl : s->slots) if (sl.lent && sl.host == buf) owner = &sl;
        owner->lent = false;
        bool ends_line = true;
        const size_t take = qm2::next_piece(lines + at, n - at, cap, s->ls.open, &ends_line);
        memcpy(buf, lines + at, take);
        if (stage(s, buf, take, owner, ends_line)) return 1;
        at += take;
    }
    return 0;
}

extern "C" int qm2_shards_finish(qm2_shards *s, qm2_stats *st)
{
    if (!s) return fail("qm2_shards_finish: null argument");
    if (run_round(s)) return 1;
    qm2_stats tot;
    memset(&tot, 0, sizeof tot);
    for (Rank &R : s->r) {
        S_OK(hipSetDevice(R.dev));
        S_OK(hipStreamSynchronize(R.s_route));
        S_OK(hipStreamSynchronize(R.s_probe));
        for (ExSet &e : R.set) e.busy = false;
        qm2_stats one;
        if (qm2_finish(R.ctx, nullptr, &one)) return 1;
        tot.n_probes += one.n_probes;
        tot.n_hits += one.n_hits;
    }
    tot.n_bases = s->n_bases;
    tot.n_kmers = s->n_kmers;
    /* the job's counters: every shard's span (own slots and overhang alike, slot for slot) added into one whole table
     * on the first device — where the epilogue walks the chain (:483-516) */
    if (!s->result) {
        qm2_opts o;
        memset(&o, 0, sizeof o);
        o.struct_size = sizeof o;
        o.device = s->r[0].dev;
        o.flags = QM2_FLAG_KEYS_LATER;
        if (qm2_open(&s->result, nullptr, s->H, s->k, &o)) return 1;
        /* (the epilogue's list ranking starts from the occupied slots: the whole key table, once, for it) */
        if (qm2_keys_from_fd(s->result, s->dict_fd, 24)) return 1;
    } else if (qm2_reset(s->result)) return 1;
    for (Rank &R : s->r)
        if (qm2_merge_counters(s->result, R.ctx)) return 1;
    if (st) *st = tot;
    return 0;
}

extern "C" qm2_ctx *qm2_shards_result(qm2_shards *s) { return s ? s->result : nullptr; }

extern "C" int qm2_shards_get_info(qm2_shards *s, qm2_shards_info *info)
{
    if (!s || !info) return fail("qm2_shards_get_info: null argument");
    *info = s->info;
    return 0;
}

extern "C" void qm2_shards_close(qm2_shards *s)
{
    if (!s) return;
    for (Rank &R : s->r) {
        (void)hipSetDevice(R.dev);
        if (R.s_route) (void)hipStreamSynchronize(R.s_route);
        if (R.s_probe) (void)hipStreamSynchronize(R.s_probe);
        if (R.s_copy) (void)hipStreamSynchronize(R.s_copy);
        if (R.comm) (void)ncclCommDestroy(R.comm);
        if (R.ctx) qm2_close(R.ctx);
        if (R.rctx) qm2_close(R.rctx);
        (void)hipSetDevice(R.dev);
        if (R.text) (void)hipFree(R.text);
        if (R.d_wraps) (void)hipFree(R.d_wraps);
        for (ExSet &e : R.set) {
            if (e.slabs) (void)hipFree(e.slabs);
            if (e.recv) (void)hipFree(e.recv);
            if (e.fill) (void)hipFree(e.fill);
            if (e.incoming) (void)hipFree(e.incoming);
            if (e.call) (void)hipFree(e.call);
            if (e.probed) (void)hipEventDestroy(e.probed);
        }
        if (R.call_host) (void)hipHostFree(R.call_host);
        if (R.fill_host) (void)hipHostFree(R.fill_host);
        if (R.incoming_host) (void)hipHostFree(R.incoming_host);
        if (R.copied) (void)hipEventDestroy(R.copied);
        if (R.routed) (void)hipEventDestroy(R.routed);
        if (R.s_route) (void)hipStreamDestroy(R.s_route);
        if (R.s_probe) (void)hipStreamDestroy(R.s_probe);
        if (R.s_copy) (void)hipStreamDestroy(R.s_copy);
    }
    for (Slot &sl : s->slots) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (s->result) qm2_close(s->result);
    delete s;
}
