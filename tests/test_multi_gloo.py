"""The N>1 host logic (quick-mer2_amd/multi.py) under torch.distributed with the
gloo backend, world_size 2 and 4, on the CPU.  The compute legs are replaced by
a stand-in engine built on the oracle (test infrastructure, defined HERE);
everything else — read partitioning, owner bucketing contract, all-to-all of
ragged k-mer slabs, overhang folding, counter reduction, final assembly — is the
product's code.  Results must equal the single-process oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def djb_np(kmers):
    """DJBHash_encode (QuicKmer.c:66-76) of a uint64 array, vectorised (checked against the oracle's in OracleEngine.route)"""
    x = np.asarray(kmers, dtype=np.uint64)
    h = np.full(x.shape, 5381, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(8):
            h = h * np.uint64(33) + ((x >> np.uint64(8 * i)) & np.uint64(0xFF))
    return h


class OracleEngine:
    """CPU stand-in for HipEngine: same method contract, oracle arithmetic."""

    def __init__(self):
        import oracle_lib as ol
        self.ol = ol

    class Ctx:
        pass

    def open(self, keys_span, hash_size, k, span, counters, own=None):
        c = self.Ctx()
        c.keys = np.ascontiguousarray(keys_span, dtype=np.uint64)
        c.H, c.k = hash_size, k
        c.lo = 0 if span is None else span[0]
        c.counters = counters
        c.close = lambda: None
        return c

    def open_router(self, hash_size, k):
        c = self.Ctx()
        c.H, c.k = hash_size, k
        c.close = lambda: None
        return c

    def new_counters(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def to_device(self, lines):
        return torch.frombuffer(bytearray(lines), dtype=torch.uint8)

    def _probe_np(self, c, kmers):
        import ctypes as C
        L = self.ol.lib()
        cnt = c.counters.numpy()
        idx = C.c_uint64()
        mask, half = c.H - 1, c.H >> 1
        for km in kmers:
            km = int(km)
            i = L.qm2o_djb(km) & mask
            step = -1 if (i & half) else 1
            while True:
                v = int(c.keys[i - c.lo])        # IndexError here = a walk left the span
                if v == km:
                    cnt[i - c.lo] += 1
                    break
                if v == 0:
                    break
                i += step

    def count(self, c, d_lines):
        kmers, _ = self.ol.extract(d_lines.numpy().tobytes(), c.k)
        self._probe_np(c, kmers)

    def on_router(self):
        import contextlib
        return contextlib.nullcontext()

    def new_set(self, n_owners, cap):
        return (torch.zeros(n_owners * cap, dtype=torch.int64), torch.zeros(n_owners, dtype=torch.int32),
                torch.zeros(3, dtype=torch.int64), torch.zeros(n_owners * cap, dtype=torch.int64),
                torch.zeros(n_owners, dtype=torch.int32))

    def route(self, c, d_lines, n_owners, cap, bufs, after=None):
        """the contract of qm2_route_device: fixed slabs, fill = what was ROUTED, call = bases, k-mers, not fitted"""
        slabs, fill, call = bufs[0], bufs[1], bufs[2]
        fill.zero_()
        call.zero_()
        if d_lines is None or d_lines.numel() == 0:
            return slabs, fill, call
        kmers, nb = self.ol.extract(d_lines.numpy().tobytes(), c.k)
        L = self.ol.lib()
        lgH = int(np.log2(c.H))
        shift = lgH - int(np.log2(n_owners))
        owner = ((djb_np(kmers) & np.uint64(c.H - 1)) >> np.uint64(shift)).astype(np.int64) if n_owners > 1 else np.zeros(len(kmers), dtype=np.int64)
        for x in kmers[:50]:
            assert L.qm2o_djb(int(x)) == int(djb_np(np.array([x]))[0])
        lost = 0
        for g in range(n_owners):
            sel = kmers[owner == g].view(np.int64)
            fill[g] = len(sel)
            keep = min(len(sel), cap)
            lost += len(sel) - keep
            slabs[g * cap:g * cap + keep] = torch.from_numpy(sel[:keep].copy())
        call[0], call[1], call[2] = nb, len(kmers), lost
        return slabs, fill, call

    def new_set_xcd(self, n_regions, cap):
        return (torch.zeros(8 * n_regions * cap, dtype=torch.int64), torch.zeros(8 * n_regions, dtype=torch.int32),
                torch.zeros(3, dtype=torch.int64), torch.zeros(8 * n_regions * cap, dtype=torch.int64),
                torch.zeros(8 * n_regions, dtype=torch.int32))

    def route_xcd(self, c, d_lines, n_regions, cap, bufs, after=None):
        """the contract of qm2_route_xcd_device: region = the top bits of the home slot, every region in 8 sub-regions
        (which of the 8 a k-mer lands in is the device's business: any is right), x-major, fill = what was ROUTED"""
        slabs, fill, call = bufs[0], bufs[1], bufs[2]
        fill.zero_()
        call.zero_()
        if d_lines is None or d_lines.numel() == 0:
            return slabs, fill, call
        kmers, nb = self.ol.extract(d_lines.numpy().tobytes(), c.k)
        L = self.ol.lib()
        shift = int(np.log2(c.H)) - int(np.log2(n_regions))
        region = ((djb_np(kmers) & np.uint64(c.H - 1)) >> np.uint64(shift)).astype(np.int64)
        xs = (np.arange(len(kmers)) * 5 + 3) % 8
        at_all = xs * n_regions + region
        order = np.argsort(at_all, kind="stable")
        ats, starts, counts = np.unique(at_all[order], return_index=True, return_counts=True)
        km = kmers[order].view(np.int64)
        lost = 0
        for at, st, n in zip(ats.tolist(), starts.tolist(), counts.tolist()):
            fill[at] = n
            keep = min(n, cap)
            lost += n - keep
            slabs[at * cap:at * cap + keep] = torch.from_numpy(km[st:st + keep].copy())
        call[0], call[1], call[2] = nb, len(kmers), lost
        return slabs, fill, call

    def probe_subslabs_after(self, c, works, slabs, cap, n_peers, n_subs, fill, peer_stride=0):
        for w in works:
            if w is not None:
                w.wait()
        stride = peer_stride or n_subs
        for p in range(n_peers):
            for q in range(n_subs):
                at = p * stride + q
                n = min(int(fill[at]), cap)
                if n:
                    self._probe_np(c, slabs[at * cap:at * cap + n].numpy().view(np.uint64))
        return None

    def probe_after(self, c, works, slabs, cap, n_slabs, fill):
        for w in works:
            if w is not None:
                w.wait()
        for g in range(n_slabs):
            n = min(int(fill[g]), cap)
            self._probe_np(c, slabs[g * cap:g * cap + n].numpy().view(np.uint64))
        return None

    def routed(self):
        return None

    def split_lines(self, d_lines):
        b = d_lines.numpy().tobytes()
        cut = b.find(b"\n", len(b) // 2)
        cut = len(b) if cut < 0 else cut + 1
        return d_lines[:cut], d_lines[cut:]

    def drain(self, c):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, qm_path, lines, out_dir, batch_bytes=None, route="flat"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from quick_mer2_amd import multi
    import oracle_lib as ol
    qm = ol.load_qm(qm_path)
    eng = OracleEngine()
    if mode == "replica":
        job = multi.ReplicaJob(dist, eng, qm["keys"], qm["k"])
    else:
        job = multi.ShardedJob(dist, eng, qm["keys"], qm["k"], torch, batch_bytes=batch_bytes, route=route)
        assert job.xcd == (route == "xcd")
    half = lines.rfind(b"\n", 0, len(lines) // 2) + 1
    job.count(lines[:half])              # two "batches"
    job.count(lines[half:])
    depth = job.finish()
    np.save(os.path.join(out_dir, "depth_%s_%d.npy" % (mode, rank)), depth)
    if mode == "sharded":
        np.save(os.path.join(out_dir, "info_%d.npy" % rank), np.array([job.rounds, job.retries, job.n_bases, job.n_kmers, job.flat_fallbacks]))
    job.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world,route", [("replica", 2, "flat"), ("sharded", 2, "flat"), ("sharded", 4, "flat"), ("sharded", 4, "xcd")])
def test_multi_rank_counts_equal_single_process(mode, world, route, tmp_path):
    import oracle_lib as ol
    import synth
    qm_path = os.path.join(HERE, "golden", "kat", "ref.fa.qm")
    qm = ol.load_qm(qm_path)
    cases = synth.kat_cases(synth.kat_reference())
    p = tmp_path / "r.fa"
    p.write_bytes(cases["reads_fa"][1][:120_000] + b"\n" + cases["polyA"][1])
    lines = ol.strip_reads(str(p))
    exp, _ = ol.count_lines(qm["keys"], qm["k"], lines)
    # (xcd: the 10-bit route — one all-to-all per XCD slice, what arrives is the receivers' first partition level;
    #  small batches so that the poly-A lines overflow a sub-region and go round the flat way)
    mp.spawn(_worker, args=(world, _free_port(), mode, qm_path, lines, str(tmp_path), 20000 if route == "xcd" else 400000, route), nprocs=world, join=True)
    for r in range(world):
        got = np.load(str(tmp_path / ("depth_%s_%d.npy" % (mode, r))))
        assert np.array_equal(got, exp), (mode, world, r)


@pytest.mark.parametrize("world,route", [(2, "flat"), (4, "flat"), (2, "xcd")])
def test_sharded_skewed_input_is_cut_and_rerouted_not_hung(world, route, tmp_path):
    """Poly-A (every k-mer goes to ONE owner) with slabs sized for an even spread: the ranks decide
    together that the batch does not fit, cut it and route it again — exact counts, no error, no hang
    (round 1 raised on one rank only and left the others in the collective).  Small batches also
    exercise the several-rounds-per-call path."""
    import oracle_lib as ol
    import synth
    qm_path = os.path.join(HERE, "golden", "kat", "ref.fa.qm")
    qm = ol.load_qm(qm_path)
    cases = synth.kat_cases(synth.kat_reference())
    p = tmp_path / "r.fa"
    poly = b"".join(b">a\n" + b"A" * 150 + b"\n" for _ in range(300))          # 45 kb of k-mer 0: one owner gets it all
    p.write_bytes(poly + cases["reads_fa"][1][:40_000] + b"\n" + poly)
    lines = ol.strip_reads(str(p))
    exp, exp_st = ol.count_lines(qm["keys"], qm["k"], lines)
    mp.spawn(_worker, args=(world, _free_port(), "sharded", qm_path, lines, str(tmp_path), 6000, route), nprocs=world, join=True)
    retries = fallbacks = 0
    bases = kmers = 0
    for r in range(world):
        got = np.load(str(tmp_path / ("depth_sharded_%d.npy" % r)))
        assert np.array_equal(got, exp), (world, r)
        info = np.load(str(tmp_path / ("info_%d.npy" % r)))
        retries += int(info[1])
        bases += int(info[2])
        kmers += int(info[3])
        fallbacks += int(info[4])
    assert retries > 0                                  # the skew did overflow the slabs
    assert fallbacks == 0 or route == "xcd"             # (the 10-bit route's small sub-regions may send a flood round the flat way)
    assert bases == exp_st["n_bases"] and kmers == exp_st["n_kmers"]   # nothing counted twice, nothing lost


def test_rank_slices_partition_the_stream():
    import __graft_entry__ as ge
    ge.load_package()
    from quick_mer2_amd import multi
    rng = np.random.default_rng(3)
    blob = b"".join(bytes(rng.integers(65, 70, int(rng.integers(0, 90)), dtype=np.uint8)) + b"\n" for _ in range(500))
    for world in (1, 2, 3, 8):
        cuts = [multi.rank_slice(blob, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(blob)
        for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
            assert b == c
        for a, b in cuts:
            assert a == b or (blob[b - 1:b] == b"\n" and (a == 0 or blob[a - 1:a] == b"\n"))


def test_shard_spans_contain_every_walk():
    """qm2_shard_span (pure host C): a probe walk that starts in the owned range never leaves the span."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import oracle_lib as ol
    for name in ("kat", "mini"):
        keys = ol.load_qm(os.path.join(HERE, "golden", name, "ref.fa.qm"))["keys"]
        H = len(keys)
        for G in (1, 2, 4, 8, 64):
            prev_hi = 0
            for g in range(G):
                own_lo, own_hi, lo, hi = pkg.pyqm2.shard_span(keys, G, g)
                assert own_lo == prev_hi and own_hi - own_lo == H // G and lo <= own_lo and hi >= own_hi
                prev_hi = own_hi
                if G > 1:
                    # spans are whole 64-slot blocks; a walk that leaves the owned range through its
                    # middle-facing edge meets an empty slot before it leaves the span
                    assert lo % 64 == 0 and hi % 64 == 0
                    if own_lo < H // 2:
                        assert lo == own_lo and np.any(keys[own_hi - 1:hi] == 0)
                        assert hi - own_hi < 64 or np.all(keys[own_hi - 1:hi - 64] != 0)   # ... and not a block more than that
                    else:
                        assert hi == own_hi and np.any(keys[lo:own_lo + 1] == 0)
                        assert own_lo - lo < 64 or np.all(keys[lo + 64:own_lo + 1] != 0)
