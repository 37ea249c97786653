"""Multi-GPU drivers for the count path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways the path shards; counts are sums of independent integer
increments, so both are exact:

  ReplicaJob  (E1, default)  every rank holds the whole dictionary (it always
      fits one MI355X: <= 32 GiB keys + 16 GiB counters) and counts its own
      share of the reads.  No data-path collective; ONE all-reduce (sum, int32)
      of the counters at the end.

  ShardedJob  (E2, BASELINE config 4)  the table is range-partitioned by the
      high bits of the home slot; each rank encodes its share of the reads and
      buckets the k-mers by owner into fixed-capacity slabs (qm2_route_device),
      exchanges fills and slabs with asynchronous all-to-alls while the next
      batch is already being encoded, and probes what it receives against its
      own span with the partitioned pipeline (qm2_probe_slabs_device).  A span =
      the owned range plus the run of occupied
      slots past its middle-facing edge (qm2_shard_span), so every probe walk
      stays local; counts that land in that overhang are folded into the
      neighbour that owns those slots at the end.

The compute legs go through an `engine` object.  The product engine is
HipEngine (the C-ABI, no CPU fallback).  The CPU test-suite passes its own
stand-in engine built on the oracle — that code lives in tests/, never here.
"""
import numpy as np

from . import pyqm2


def rank_slice(buf, rank, world):
    """[start, end) of `buf` (bytes of whole '\\n'-terminated lines) that rank `rank` of `world`
    counts: equal byte shares moved forward to the next line boundary."""
    n = len(buf)

    def cut(i):
        if i <= 0:
            return 0
        if i >= n:
            return n
        j = buf.find(b"\n", i - 1)
        return n if j < 0 else j + 1

    return cut(n * rank // world), cut(n * (rank + 1) // world)


class HipEngine:
    """The product's compute legs on one GPU (device pointers are torch tensors' data_ptr()).

    Two HIP streams per rank: the ROUTER stream carries encode + bucket-by-owner (qm2_route_device) and
    the collectives that follow it; the PROBER stream carries the probe of what has arrived
    (qm2_probe_slabs_device).  The prober waits for an exchange through its work handle, the router never
    does: while batch i is on the wire and being probed, batch i+1 is already being encoded."""

    def __init__(self, torch, device):
        self.torch = torch
        self.device = device
        self.s_route = torch.cuda.Stream(device)
        self.s_probe = torch.cuda.Stream(device)

    def open(self, keys_span, hash_size, k, span, counters, own=None, ext_keys=None, flags=0):
        """keys_span: host keys of the span, or None with ext_keys = device pointer to them"""
        self._torch_done()
        return pyqm2.Counter(keys_span, hash_size=hash_size, k=k, device=self.device.index or 0, flags=flags,
                             span=span, own=own, ext_counters=counters.data_ptr(), ext_keys=ext_keys,
                             stream=self.s_probe.cuda_stream if span is not None else None)

    def open_router(self, hash_size, k, flags=0):
        """A context that only encodes and routes (no table: a 64-slot stand-in span)."""
        return pyqm2.Counter(None, hash_size=hash_size, k=k, device=self.device.index or 0, flags=pyqm2.FLAG_KEYS_LATER | flags,
                             span=(0, min(64, hash_size)), stream=self.s_route.cuda_stream)

    def new_counters(self, n):
        return self.torch.zeros(n, dtype=self.torch.int32, device=self.device)

    def to_device(self, lines: bytes):
        return self.torch.frombuffer(bytearray(lines), dtype=self.torch.uint8).to(self.device)

    def _torch_done(self):
        # the library runs on its own non-blocking stream: torch-side fills / copies that produce its
        # inputs must have finished before it is called
        self.torch.cuda.current_stream(self.device).synchronize()

    def count(self, ctx, d_lines):
        self._torch_done()
        ctx.submit_device(d_lines.data_ptr(), d_lines.numel())
        ctx.sync()

    def on_router(self):
        return self.torch.cuda.stream(self.s_route)

    def new_set(self, n_owners, cap):
        """One set of exchange buffers: (slabs, fill, call, recv, incoming); allocated once, reused."""
        t = self.torch
        return (t.empty(n_owners * cap, dtype=t.int64, device=self.device), t.zeros(n_owners, dtype=t.int32, device=self.device),
                t.zeros(3, dtype=t.int64, device=self.device), t.empty(n_owners * cap, dtype=t.int64, device=self.device),
                t.zeros(n_owners, dtype=t.int32, device=self.device))

    def route(self, rctx, d_lines, n_owners, cap, bufs, after=None):
        """Fills bufs = (slabs int64[n_owners * cap], fill int32[n_owners], call int64[3] = bases, k-mers, not
        fitted, ...) on the router stream (the caller is inside on_router()); `after`: what probe_after returned
        for the round that used these buffers last — the router waits for it before it overwrites them."""
        if after is not None:
            self.s_route.wait_event(after)
        slabs, fill, call = bufs[0], bufs[1], bufs[2]
        fill.zero_()
        call.zero_()
        if d_lines is not None and d_lines.numel():
            rctx.route_device(d_lines.data_ptr(), d_lines.numel(), slabs.data_ptr(), cap, n_owners, fill.data_ptr(),
                              call.data_ptr())
        return slabs, fill, call

    def new_set_xcd(self, n_regions, cap):
        """Exchange buffers of the 10-bit route: (slabs [8][n_regions][cap], fill [8][n_regions], call, recv, incoming) — x-major,
        as qm2_route_xcd_device writes them; recv / incoming take [8][world][n_subs] with world * n_subs = n_regions."""
        t = self.torch
        return (t.empty(8 * n_regions * cap, dtype=t.int64, device=self.device), t.zeros(8 * n_regions, dtype=t.int32, device=self.device),
                t.zeros(3, dtype=t.int64, device=self.device), t.empty(8 * n_regions * cap, dtype=t.int64, device=self.device),
                t.zeros(8 * n_regions, dtype=t.int32, device=self.device))

    def route_xcd(self, rctx, d_lines, n_regions, cap, bufs, after=None):
        """route() by 10 bits: n_regions = owners x sub-slabs (<= 1024), every region in 8 per-XCD sub-regions of `cap` k-mers"""
        if after is not None:
            self.s_route.wait_event(after)
        slabs, fill, call = bufs[0], bufs[1], bufs[2]
        fill.zero_()
        call.zero_()
        if d_lines is not None and d_lines.numel():
            rctx.route_xcd_device(d_lines.data_ptr(), d_lines.numel(), slabs.data_ptr(), cap, n_regions, fill.data_ptr(), call.data_ptr())
        return slabs, fill, call

    def probe_subslabs_after(self, ctx, works, slabs, cap, n_peers, n_subs, fill, peer_stride=0):
        """probe_after() for sub-slabs that are the first level of the shard's own partition (qm2_probe_subslabs_device)"""
        with self.torch.cuda.stream(self.s_probe):
            for w in works:
                if w is not None:
                    w.wait()
            ctx.probe_subslabs_device(slabs.data_ptr(), cap, n_peers, n_subs, fill.data_ptr(), kmers_bound=0, peer_stride=peer_stride)
            ev = self.torch.cuda.Event()
            ev.record(self.s_probe)
        return ev

    class _After:
        """what a work handle is to a collective: the prober stream waits for the router stream's work so far"""
        def __init__(self, torch, stream):
            self.ev = torch.cuda.Event()
            self.ev.record(stream)
            self.torch = torch

        def wait(self):
            self.torch.cuda.current_stream().wait_event(self.ev)

    def routed(self):
        return self._After(self.torch, self.s_route)

    def probe_after(self, ctx, works, slabs, cap, n_slabs, fill):
        """Probe the received slabs once the exchanges `works` have completed — on the prober stream.
        Returns an event that fires when the probe is done with the buffers."""
        with self.torch.cuda.stream(self.s_probe):
            for w in works:
                if w is not None:
                    w.wait()                    # the prober stream waits; the host and the router stream do not
            ctx.probe_slabs_device(slabs.data_ptr(), cap, n_slabs, fill.data_ptr())
            ev = self.torch.cuda.Event()
            ev.record(self.s_probe)
        return ev

    def split_lines(self, d_lines):
        """Two halves of a device text buffer, cut behind a line end."""
        n = d_lines.numel()
        nl = (d_lines[n // 2:] == 10).nonzero()
        cut = n // 2 + int(nl[0]) + 1 if nl.numel() else n
        return d_lines[:cut], d_lines[cut:]

    def drain(self, ctx):
        self.s_route.synchronize()
        self.s_probe.synchronize()
        # qm2_finish, not qm2_sync: the counter tensor this job owns (ext_counters) is complete only behind it
        # (the counts of the bucket interiors are folded in there: include/qm2.h)
        ctx.finish(want_depth=False)


class ReplicaJob:
    """E1: dictionary replicated, reads partitioned, one all-reduce at the end."""

    def __init__(self, dist, engine, keys, k):
        self.dist, self.engine = dist, engine
        self.H = len(keys)
        self.counters = engine.new_counters(self.H)
        self.ctx = engine.open(keys, self.H, k, None, self.counters)

    def count(self, lines: bytes):
        """Every rank passes the SAME buffer; each counts its own line-aligned share."""
        a, b = rank_slice(lines, self.dist.get_rank(), self.dist.get_world_size())
        if b > a:
            self.engine.count(self.ctx, self.engine.to_device(lines[a:b]))

    def count_local(self, d_lines):
        """Rank-private reads already on the device."""
        self.engine.count(self.ctx, d_lines)

    def finish(self):
        """-> uint16[H] on every rank (mod-2^16 truncation commutes with the sum)."""
        self.engine.drain(self.ctx)          # (the counter tensor is complete behind qm2_finish)
        if self.dist.get_world_size() > 1:
            step = 1 << 28                   # 1 GiB pieces: no 2^31-element count reaches the collective
            for a in range(0, self.counters.numel(), step):
                self.dist.all_reduce(self.counters[a:a + step])
        return (self.counters.cpu().numpy().view(np.uint32) & 0xFFFF).astype(np.uint16)

    def close(self):
        self.ctx.close()
        if getattr(self, "rctx", None) is not None:
            self.rctx.close()


class ShardedJob:
    """E2: table range-partitioned by home slot; k-mers routed by all-to-all in fixed-capacity slabs.

    Per batch and rank: route (encode + bucket by owner, device-side fills) -> ONE small all-reduce that
    makes the "did every slab fit" decision collective -> all-to-all of the fills and of the slabs
    (fixed size: no host-side split sizes, no device-to-host copy of counts) -> probe of the received
    slabs with their device-side fills.  The collectives and the probe are asynchronous; the only thing
    the host ever waits for is the one-word fit flag of the batch it is about to send.  A batch that
    does not fit (skewed input: poly-A sends everything to one owner) is cut in two behind a line end
    and routed again, by every rank alike — exact for any input, never an error and never a hang.
    Every rank must call count / count_local the same number of times (an empty call is fine)."""

    def __init__(self, dist, engine, keys, k, torch, batch_bytes=None, resident=None, flags=0, route=None):
        """keys: the host key table; or None with resident = (H, (own_lo, own_hi, lo, hi), device pointer to the
        span's keys) when the table is already on the device (bench.py)."""
        self.dist, self.engine, self.torch = dist, engine, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert self.world & (self.world - 1) == 0, "shard count must be a power of two"
        self.k = k
        if resident is None:
            self.H = len(keys)
            self.own_lo, self.own_hi, self.lo, self.hi = pyqm2.shard_span(keys, self.world, self.rank)
        else:
            self.H, (self.own_lo, self.own_hi, self.lo, self.hi) = resident[0], resident[1]
        self.counters = engine.new_counters(self.hi - self.lo)
        # the router needs no table; the prober holds only this rank's span
        if resident is None:
            self.ctx = engine.open(np.ascontiguousarray(keys[self.lo:self.hi]), self.H, k, (self.lo, self.hi), self.counters,
                                   own=(self.own_lo, self.own_hi))
            self.rctx = engine.open_router(self.H, k)
        else:
            self.ctx = engine.open(None, self.H, k, (self.lo, self.hi), self.counters, own=(self.own_lo, self.own_hi),
                                   ext_keys=resident[2], flags=flags)
            self.rctx = engine.open_router(self.H, k, flags=flags)
        self.batch_bytes = int(batch_bytes) if batch_bytes else 1 << 26
        # a slab takes a rank's share of one batch with 30 % head-room (uniform hashing spreads the k-mers evenly)
        self.cap = (int(self.batch_bytes / self.world * 1.3) + 64) & ~1
        # The exchange layout (route = "xcd" | "flat"; None: QM2_SHARDS_ROUTE, else xcd where the library can run it — tables
        # of 2^23 slots and more).  flat: one slab per owner, split twice by the receiver.  xcd: the senders' 10-bit route
        # (qm2_route_xcd_device) — owners x 1024 / owners sub-slabs, each in 8 per-XCD sub-regions of a fixed capacity,
        # x-major — exchanged with one all-to-all per XCD slice (a slice is owner-major: equal splits), and what a shard
        # receives IS the first level of its own partition (qm2_probe_subslabs_device: one split).  Both exact; which one
        # the wire prefers (whole blocks with ~13 % of room in them, or less data split once more) is for the first run on
        # more than one GPU to say.
        import os as _os
        if route is None:
            route = _os.environ.get("QM2_SHARDS_ROUTE")
        can = self.H >= (1 << 23) and self.world <= 512 and k <= 30 and hasattr(engine, "route_xcd")
        assert route in (None, "xcd", "flat"), route
        self.xcd = (route == "xcd") or (route is None and can)
        self.n_subs = 1024 // self.world if self.xcd else 1
        if self.xcd:
            mean = self.batch_bytes / (8.0 * 1024.0)
            self.sub_cap = (int(mean * 1.12 + 8.0 * (mean + 1.0) ** 0.5) + 256) & ~1
        self.n_bases = self.n_kmers = 0
        self.rounds = self.retries = self.flat_fallbacks = 0
        self._xsets = None                    # the 10-bit route's two sets of exchange buffers
        self._xbusy = [None, None]
        self._keep = []
        self._sets = None                     # two sets of exchange buffers, used in turn
        self._busy = [None, None]             # per set: the probe that reads it has finished (engine event)
        self._turn = 0
        self._acc = None                      # device-side {bases, k-mers, -} of the accepted rounds

    def count(self, lines: bytes):
        a, b = rank_slice(lines, self.rank, self.world)
        self.count_local(self.engine.to_device(lines[a:b]) if b > a else None)

    def count_local(self, d_lines):
        t, dist = self.torch, self.dist
        n = d_lines.numel() if d_lines is not None else 0
        # the same number of rounds on every rank: the largest share decides
        rounds = t.tensor([(n + self.batch_bytes - 1) // self.batch_bytes], dtype=t.int64, device=self.counters.device)
        dist.all_reduce(rounds, op=dist.ReduceOp.MAX)
        rounds = max(1, int(rounds.item()))
        pieces = [d_lines] if rounds == 1 else self._cut(d_lines, rounds)
        for piece in pieces:
            self._round(piece)

    def _cut(self, d_lines, rounds):
        out, rest = [], d_lines
        for left in range(rounds, 1, -1):
            if rest is None or rest.numel() == 0:
                out.append(None)
                continue
            n = rest.numel()
            want = (n + left - 1) // left
            nl = (rest[want - 1:] == 10).nonzero()
            cut = want + int(nl[0]) if nl.numel() else n
            out.append(rest[:cut])
            rest = rest[cut:]
        out.append(rest)
        return out

    def _round_xcd(self, d_lines, depth=0):
        """One round by the 10-bit route.  A piece that overflows its small sub-regions is cut like a flat one; once every
        rank's piece is small enough for one slab per owner to hold it whatever it is made of (a flood of one k-mer),
        it goes round the flat way."""
        t, dist, W, eng = self.torch, self.dist, self.world, self.engine
        have = d_lines is not None and d_lines.numel() > 0
        self.rounds += 1
        nr, cap, ns = W * self.n_subs, self.sub_cap, self.n_subs
        if self._xsets is None:
            self._xsets = [eng.new_set_xcd(nr, cap), eng.new_set_xcd(nr, cap)]
            if self._acc is None:
                self._acc = self._xsets[0][2].clone()
        which = self._turn
        bufs = self._xsets[which]
        with eng.on_router():
            slabs, fill, call = eng.route_xcd(self.rctx, d_lines if have else None, nr, cap, bufs, after=self._xbusy[which])
            self._xbusy[which] = None
            flag = t.stack([call[2], t.tensor(d_lines.numel() if have else 0, dtype=call.dtype, device=call.device)])
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            fits, largest = int(flag[0].item()) == 0, int(flag[1].item())
            works = []
            if fits:
                if W == 1:
                    recv, incoming, works = slabs, fill, [eng.routed()]
                else:
                    recv, incoming = bufs[3], bufs[4]
                    blk = nr * cap                       # one XCD slice: [owner][sub][cap], owner-major
                    for x in range(8):
                        works.append(dist.all_to_all_single(incoming[x * nr:(x + 1) * nr], fill[x * nr:(x + 1) * nr], async_op=True))
                        works.append(dist.all_to_all_single(recv[x * blk:(x + 1) * blk], slabs[x * blk:(x + 1) * blk], async_op=True))
                self._acc += call
        if not fits:
            assert depth < 48, "a batch keeps overflowing its sub-slabs"
            self.retries += 1
            if largest <= self.cap:                   # (the same number on every rank) one slab per owner holds any such piece
                self.flat_fallbacks += 1
                self._round(d_lines, depth + 1, flat=True)
                return
            halves = eng.split_lines(d_lines) if have else (None, None)
            self._round_xcd(halves[0], depth + 1)
            self._round_xcd(halves[1], depth + 1)
            return
        # received: slice x holds [sender][sub][cap] — 8 W "peers" of n_subs sub-slabs side by side (W = 1: what was routed,
        # 8 peers 1024 regions apart)
        self._xbusy[which] = eng.probe_subslabs_after(self.ctx, works, recv, cap, 8 * W if W > 1 else 8, ns if W > 1 else nr, incoming,
                                                      peer_stride=0 if W > 1 else nr)
        self._keep = (self._keep + [d_lines])[-2:]
        self._turn ^= 1

    def _round(self, d_lines, depth=0, flat=False):
        if self.xcd and not flat:
            return self._round_xcd(d_lines, depth)
        t, dist, W, eng = self.torch, self.dist, self.world, self.engine
        have = d_lines is not None and d_lines.numel() > 0
        self.rounds += 1
        if self._sets is None:
            self._sets = [eng.new_set(W, self.cap), eng.new_set(W, self.cap)]
            if self._acc is None:
                self._acc = self._sets[0][2].clone()
                self._acc.zero_()
        which = self._turn
        bufs = self._sets[which]
        with eng.on_router():
            slabs, fill, call = eng.route(self.rctx, d_lines if have else None, W, self.cap, bufs, after=self._busy[which])
            self._busy[which] = None
            # collective decision: did every slab on every rank take what was routed to it?
            unfit = call[2:3].clone()
            dist.all_reduce(unfit, op=dist.ReduceOp.MAX)
            fits = int(unfit.item()) == 0
            if fits:
                if W == 1:                        # nothing to exchange: what was routed is what is probed
                    recv, incoming, w1, w2 = slabs, fill, eng.routed(), None
                else:
                    recv, incoming = bufs[3], bufs[4]
                    w1 = dist.all_to_all_single(incoming, fill, async_op=True)
                    w2 = dist.all_to_all_single(recv, slabs, async_op=True)
                self._acc += call
        if not fits:
            # every rank takes this branch together; a piece of at most `cap` bytes cannot overflow a slab
            assert depth < 48, "a batch keeps overflowing its slabs"
            self.retries += 1
            halves = eng.split_lines(d_lines) if have else (None, None)
            self._round(halves[0], depth + 1, flat)
            self._round(halves[1], depth + 1, flat)
            return
        self._busy[which] = eng.probe_after(self.ctx, (w1, w2), recv, self.cap, W, incoming)
        self._keep = (self._keep + [d_lines])[-2:]      # (the text of the rounds still being routed)
        self._turn ^= 1

    def finish(self):
        """-> uint16[H] on every rank, assembled from the owners' slices."""
        t, dist, W = self.torch, self.dist, self.world
        self.engine.drain(self.ctx)
        self._keep = []
        if self._acc is not None:
            self.n_bases, self.n_kmers = int(self._acc[0]), int(self._acc[1])
        c = self.counters
        owned = c[self.own_lo - self.lo:self.own_hi - self.lo].clone()
        # overhang past the middle-facing edge belongs to the next shard towards the middle
        over_hi = c[self.own_hi - self.lo:] if self.hi > self.own_hi else c[:0]
        over_lo = c[:self.own_lo - self.lo] if self.lo < self.own_lo else c[:0]
        sizes = t.tensor([over_lo.numel(), over_hi.numel()], dtype=t.int64, device=c.device)
        all_sizes = [t.empty_like(sizes) for _ in range(W)]
        dist.all_gather(all_sizes, sizes)
        width = int(max(int(s.max()) for s in all_sizes))
        pad = t.zeros(2 * max(1, width), dtype=c.dtype, device=c.device)
        pad[:over_lo.numel()] = over_lo
        pad[max(1, width):max(1, width) + over_hi.numel()] = over_hi
        gathered = [t.empty_like(pad) for _ in range(W)]
        dist.all_gather(gathered, pad)
        per = self.H // W
        for g in range(W):
            n_lo, n_hi = int(all_sizes[g][0]), int(all_sizes[g][1])
            if n_hi and g + 1 == self.rank:       # g's overhang above its range starts at my own_lo
                assert n_hi <= per, "overhang longer than a shard"
                owned[:n_hi] += gathered[g][max(1, width):max(1, width) + n_hi]
            if n_lo and g - 1 == self.rank:       # g's overhang below its range ends at my own_hi
                assert n_lo <= per, "overhang longer than a shard"
                owned[per - n_lo:] += gathered[g][:n_lo]
        slices = [t.empty_like(owned) for _ in range(W)]
        dist.all_gather(slices, owned)
        full = t.cat(slices).cpu().numpy().view(np.uint32)
        return (full & 0xFFFF).astype(np.uint16)

    def close(self):
        self.ctx.close()
        if getattr(self, "rctx", None) is not None:
            self.rctx.close()
