"""Multi-GPU drivers for the count path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Two ways the path shards; counts are sums of independent integer
increments, so both are exact:

  ReplicaJob  (E1, default)  every rank holds the whole dictionary (it always
      fits one MI355X: <= 32 GiB keys + 16 GiB counters) and counts its own
      share of the reads.  No data-path collective; ONE all-reduce (sum, int32)
      of the counters at the end.

  ShardedJob  (E2, BASELINE config 4)  the table is range-partitioned by the
      high bits of the home slot; each rank encodes its share of the reads,
      buckets the k-mers by owner (qm2_encode_device), exchanges them with one
      all-to-all per batch, and probes what it receives against its own span
      (qm2_probe_device).  A span = the owned range plus the run of occupied
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
    """The product's compute legs on one GPU (device pointers are torch tensors' data_ptr())."""

    def __init__(self, torch, device):
        self.torch = torch
        self.device = device

    def open(self, keys_span, hash_size, k, span, counters):
        self._torch_done()
        return pyqm2.Counter(keys_span, hash_size=hash_size, k=k, device=self.device.index or 0,
                             span=span, ext_counters=counters.data_ptr())

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

    def encode(self, ctx, d_lines, n_buckets):
        """-> (kmers int64[n_buckets * cap] on device, fills int64[n_buckets] on device, cap)"""
        t = self.torch
        n = d_lines.numel()
        # uniform hashing spreads the <= n k-mers evenly over the owners: 30 % head-room per bucket
        # (a bucket that still overflows is reported by the library as an error, never dropped silently)
        cap = max(16, n if n_buckets == 1 else int(n / n_buckets * 1.3) + 4096)
        out = t.empty(n_buckets * cap, dtype=t.int64, device=self.device)
        fills = t.zeros(n_buckets, dtype=t.int64, device=self.device)
        self._torch_done()
        ctx.encode_device(d_lines.data_ptr(), d_lines.numel(), out.data_ptr(), cap, n_buckets, fills.data_ptr())
        ctx.sync()
        return out, fills, cap

    def probe(self, ctx, d_kmers):
        if d_kmers.numel():
            self._torch_done()
            ctx.probe_device(d_kmers.data_ptr(), d_kmers.numel())
            ctx.sync()


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
        if self.dist.get_world_size() > 1:
            step = 1 << 28                   # 1 GiB pieces: no 2^31-element count reaches the collective
            for a in range(0, self.counters.numel(), step):
                self.dist.all_reduce(self.counters[a:a + step])
        return (self.counters.cpu().numpy().view(np.uint32) & 0xFFFF).astype(np.uint16)

    def close(self):
        self.ctx.close()


class ShardedJob:
    """E2: table range-partitioned by home slot; k-mers routed by all-to-all."""

    def __init__(self, dist, engine, keys, k, torch):
        self.dist, self.engine, self.torch = dist, engine, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert self.world & (self.world - 1) == 0, "shard count must be a power of two"
        self.H, self.k = len(keys), k
        self.own_lo, self.own_hi, self.lo, self.hi = pyqm2.shard_span(keys, self.world, self.rank)
        self.counters = engine.new_counters(self.hi - self.lo)
        # the encoder needs no table; the prober holds only this rank's span
        self.ctx = engine.open(np.ascontiguousarray(keys[self.lo:self.hi]), self.H, k,
                               (self.lo, self.hi), self.counters)

    def count(self, lines: bytes):
        a, b = rank_slice(lines, self.rank, self.world)
        self.count_local(self.engine.to_device(lines[a:b]) if b > a else None)

    def count_local(self, d_lines):
        t, dist, W = self.torch, self.dist, self.world
        if d_lines is not None and d_lines.numel():
            out, fills, cap = self.engine.encode(self.ctx, d_lines, W)
        else:
            cap = 16
            out = t.empty(W * cap, dtype=t.int64, device=self.counters.device)
            fills = t.zeros(W, dtype=t.int64, device=self.counters.device)
        # 1) tell every peer how many k-mers it is about to get
        incoming = t.empty_like(fills)
        dist.all_to_all_single(incoming, fills)
        send = [int(x) for x in fills.cpu()]
        recv = [int(x) for x in incoming.cpu()]
        # 2) the k-mers themselves: one all-to-all of ragged int64 slabs
        packed = t.cat([out[g * cap:g * cap + send[g]] for g in range(W)]) if sum(send) else out[:0]
        got = t.empty(sum(recv), dtype=t.int64, device=out.device)
        dist.all_to_all_single(got, packed, output_split_sizes=recv, input_split_sizes=send)
        # 3) probe + increment locally
        self.engine.probe(self.ctx, got)

    def finish(self):
        """-> uint16[H] on every rank, assembled from the owners' slices."""
        t, dist, W = self.torch, self.dist, self.world
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
