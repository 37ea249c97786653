#!/usr/bin/env python3
"""bench.py — input bases/sec counted by the MI355X `count` hot path.

One "step" = one pass of the hot path (through the C-ABI, qm2_submit_device) over one batch of
synthetic reads that is already resident in HBM: `value` is that rate (inputs in HBM when the
timed region starts).  The same JSON line also carries
  * streamed_bases_per_s — the SURVEY 8(d) wall-clock metric: the same batches fed from host memory
    through the pinned staging buffers (qm2_staging_acquire / commit, as the CLI's reader does),
    first commit -> return of qm2_finish, every allocation made beforehand (qm2_prepare); once as
    stripped sequence lines and once as raw FASTQ text with the line discipline on the device;
  * roofline — the WHOLE path: SURVEY 8(d)'s algorithmic bytes of a step / the summed kernel time of
    a step (HIP events on the kernels' own stream) against the 8 TB/s HBM peak, with the per-kernel
    table (and the dominant kernel's own figures) beneath it;
  * cpu_baseline — the real reference binary (oracle/_ref/quicKmer2, `count -t 6`) timed on this
    box's cores on the same dictionary, beside the oracle port and its thread sweep.

Workloads (BASELINE.json `configs`):

  c3  (default; BASELINE.json's metric is quoted on the hg38 k=30 dictionary, which fits one
      GPU) hg38-like dictionary: 3.1 Gb synthetic reference, k=30, H=2^32, 2.27 G unique
      k-mers (32 GiB keys + 16 GiB u32 counters), batches of 60 M x 150 bp reads
  c2  chr22-like dictionary: 50.8 Mb synthetic reference, k=30, H=2^28
      (2 GiB keys + 1 GiB counters), batches of 10 M x 150 bp reads
  c5  c3's dictionary, HiFi-like reads U(15k, 20k) bp

N > 1 (launched by torch.distributed.run, one rank per GPU): replicas — every
rank holds the whole dictionary (it always fits: <= 48 GiB)
and counts its own batches with no data-path collective; the only exchange is
ONE RCCL reduce of the u32 counters onto rank 0 per job, after the last step
(timed separately: "final_reduce_ms").  scaling = weak.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# VALU issue: a wave64 instruction occupies a SIMD's 16 lanes for 4 cycles; 4 SIMDs per CU (MI355X_MICROARCH.md)
VALU_WAVE_INSTS_PER_CU_CYCLE = 1.0


def kernel_source_sha16():
    """Identifies the kernels a PMC file was measured on: sha256 over the device code of libqm2_hip.so — the kernel headers
    csrc/*.hip.h (16 hex digits; the launch parameters in qm2_hip.hip are not part of it).  tools/summarize_profile.py stamps
    profiles/traffic_<workload>.json with it; a bench line only quotes the file's counters (roofline.traffic, roofline.valu)
    when the stamp is that of the tree it runs from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "quick-mer2_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(gpus, json_out):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start N ranks of this very script under
    torch.distributed.run as a CHILD process (nothing here has touched the GPU yet, and nothing will), relay the one JSON
    line rank 0 prints and the exit code.  (The reference's own N-way split is its worker pool, QuicKmer.c:368-384.)"""
    import shlex
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launcher = shlex.split(os.environ["QM2_BENCH_LAUNCHER"]) if os.environ.get("QM2_BENCH_LAUNCHER") else [sys.executable, "-m", "torch.distributed.run"]
    cmd = launcher + ["--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                      os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a rank environment: starting %s\n" % (gpus, " ".join(cmd)))
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    line = None
    for raw in pr.stdout:
        txt = raw.decode(errors="replace")
        if txt.startswith("{"):
            line = txt
        else:
            sys.stderr.write(txt)
    rc = pr.wait()
    if line is not None:
        json_out.write(line if line.endswith("\n") else line + "\n")
        json_out.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no JSON line\n")
        rc = 1
    return rc

WORKLOADS = {
    "c2": dict(genome=50_818_468, lgH=28, reads=10_000_000, len_lo=150, len_hi=150, sub=0.005,
               desc="chr22-like 50.8 Mb synthetic reference, k=30, H=2^28; 10M x 150 bp reads per step"),
    "c3": dict(genome=3_100_000_000, lgH=32, reads=60_000_000, len_lo=150, len_hi=150, sub=0.005,
               desc="hg38-like 3.1 Gb synthetic reference, k=30, H=2^32; 60M x 150 bp reads per step"),
    "c5": dict(genome=3_100_000_000, lgH=32, reads=400_000, len_lo=15_000, len_hi=20_000, sub=0.001,
               desc="hg38-like dictionary (H=2^32), HiFi-like reads U(15k,20k) bp, 400k reads per step"),
    "tiny": dict(genome=2_000_000, lgH=22, reads=200_000, len_lo=150, len_hi=150, sub=0.005,
                 desc="2 Mb synthetic reference, H=2^22 (plumbing check only)"),
}


def genome_layout(n):
    """N blocks and duplicated segments of the synthetic reference: ~23 % N in a
    few blocks (chr22: 11.6 of 50.8 Mb), 2 % of the sequence in duplicated 1-20 kb segments."""
    n_blocks = [(0, int(0.205 * n)), (int(0.59 * n), int(0.012 * n)), (n - int(0.0102 * n), int(0.0102 * n))]
    dups = []
    target, at, src, i = int(0.02 * n), int(0.75 * n), int(0.25 * n), 0
    while target > 0:
        ln = 1000 + (i * 7919) % 19000
        dups.append((at, src, ln))
        at += ln + 50_000
        src += ln + 31_337
        target -= ln
        i += 1
        if at + 20_000 >= n - int(0.0102 * n) or src + 20_000 >= int(0.58 * n):
            break
    gaps = sorted(n_blocks)
    covered, at = [], 0
    for st, ln in gaps:
        if st > at:
            covered.append((at, st - at))
        at = st + ln
    if at < n:
        covered.append((at, n - at))
    return n_blocks, dups, covered


def cpu_legs(args, wl, pyqm2, torch, keys, buf0, nbytes0, H, k, local, n_unique):
    """Everything timed on the host cores (rank 0, N = 1 only): the oracle port at the reference's
    recommended -t 6 on a bounded sample + the same sample through the GPU (bit-exact check), the port
    at the other worker counts of SURVEY 8(d), and the REAL reference binary on the same dictionary."""
    import numpy as np
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    hk = keys.cpu().numpy().view(np.uint64)
    ncpu = os.cpu_count() or 1

    def sample(nb):
        host = buf0[:min(nbytes0, int(nb))].cpu().numpy().tobytes()
        return host[:host.rfind(b"\n") + 1]

    def timed(host, thr):
        tA = time.perf_counter()
        dep, st = ol.count_lines(hk, k, host, threads=thr)
        return dep, st, time.perf_counter() - tA

    # calibrate on a small slice, then size the -t 6 sample for ~cpu_seconds
    host = sample(400_000 * 151)
    _, st, sec = timed(host, args.cpu_threads)
    rate6 = st["n_bases"] / max(1e-6, sec)
    host = sample(max(len(host), rate6 * 1.0067 * args.cpu_seconds))
    cdepth, cst, cpu_s = timed(host, args.cpu_threads)
    cdepth = ol.remove_drain_padding(cdepth, hk, cst["n_kmers"])
    chk = pyqm2.Counter(None, hash_size=H, k=k, device=local, ext_keys=keys.data_ptr())
    chk.submit(host)
    gdepth, gst = chk.finish()
    chk.close()
    same = bool(np.array_equal(gdepth, cdepth) and gst["n_kmers"] == cst["n_kmers"])
    del gdepth, cdepth
    port = {"value": round(cst["n_bases"] / cpu_s, 1), "unit": "bases/s", "cores": args.cpu_threads + 1,
            "sample": "first %d reads (%d bases) of batch 0, oracle threaded model: 1 producer + %d probe workers, %.1f s"
                      % (host.count(b"\n"), cst["n_bases"], args.cpu_threads, cpu_s),
            "gpu_counts_identical_on_sample": same}
    # -t 0, -t 6 and the best of {12, 24, nproc-1}
    sweep = {str(args.cpu_threads): port["value"]}
    for thr in [int(x) for x in args.cpu_sweep.split(",") if x != ""]:
        if thr == args.cpu_threads or thr >= ncpu:
            continue
        est = rate6 * (0.12 if thr == 0 else min(2.0, thr / 6.0))
        sub = sample(max(151 * 20000, est * args.sweep_seconds))
        _, sst, ssec = timed(sub, thr)
        sweep[str(thr)] = round(sst["n_bases"] / ssec, 1)
    port["sweep_bases_per_s_by_workers"] = sweep
    port["best_workers"] = max(sweep, key=lambda t: sweep[t])
    out = dict(port, kind="port", host_cores=ncpu)

    # the reference itself: `quicKmer2 count -t 6` on a .qm of THIS dictionary (header + keys; it is
    # stopped at its "Counting elapse" line, before it would read and walk the chain, :481-:516)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "quicKmer2")
    import shutil
    need = 24 + 8 * H + 2.2e9
    roots = [r for r in (args.scratch, "/dev/shm") if os.path.isdir(r) and os.access(r, os.W_OK)]
    root = max(roots, key=lambda r: shutil.disk_usage(r).free) if roots else args.scratch
    work = tempfile.mkdtemp(prefix="qm2_bench_", dir=root)
    try:
        if args.no_reference_binary or not os.path.exists(ref_bin) or shutil.disk_usage(work).free < need * 1.1:
            out["reference_binary"] = "skipped (%s)" % ("--no-reference-binary" if args.no_reference_binary else
                                                         "oracle/_ref/quicKmer2 not shipped" if not os.path.exists(ref_bin)
                                                         else "scratch space below %.0f GB" % (need * 1.1 / 1e9))
        else:
            with open(os.path.join(work, "ref.fa.qm"), "wb") as f:
                f.write(b"QM11" + bytes([k, 0, 100, 100]) + int(H).to_bytes(8, "little") + (0).to_bytes(8, "little"))
                hk.tofile(f)
            host = sample(rate6 * 1.0067 * args.ref_seconds)
            n_lines = host.count(b"\n")
            with open(os.path.join(work, "reads.fa"), "wb") as f:
                f.write(b">r\n")
                f.write(host)
            tA = time.perf_counter()
            # The reference printf()s through stdio, which holds its lines back on a pipe: give it a pseudo-terminal,
            # or (no pty devices on the box) preload a three-line shim that makes its stdout line-buffered — what
            # `stdbuf -oL` does; the reference binary itself is untouched.
            import pty
            cmd = [ref_bin, "count", "-t", str(args.cpu_threads), os.path.join(work, "ref.fa"),
                   os.path.join(work, "reads.fa"), os.path.join(work, "out")]
            env = dict(os.environ)
            try:
                master, slave = pty.openpty()
            except OSError:
                shim_dir = tempfile.mkdtemp(prefix="qm2_lb_")            # (not under /dev/shm: it may be mounted noexec)
                shim_c, shim_so = os.path.join(shim_dir, "lb.c"), os.path.join(shim_dir, "lb.so")
                open(shim_c, "w").write('#include <stdio.h>\n__attribute__((constructor)) static void lb(void) { setvbuf(stdout, NULL, _IOLBF, 0); }\n')
                subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", shim_so, shim_c], check=True)
                env["LD_PRELOAD"] = shim_so + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
                master, slave = os.pipe()
            pr = subprocess.Popen(cmd, stdout=slave, stderr=slave, stdin=subprocess.DEVNULL, close_fds=True, env=env)
            os.close(slave)
            t_loaded = t_done = None
            nk_ref = None
            pending = b""
            deadline = tA + 600
            while t_done is None and time.perf_counter() < deadline:
                try:
                    chunk = os.read(master, 4096)
                except OSError:
                    break
                if not chunk:
                    break
                now = time.perf_counter()
                pending += chunk
                while b"\n" in pending:
                    line, pending = pending.split(b"\n", 1)
                    line = line.decode(errors="replace").strip()
                    if line.startswith("Read 0x") and t_loaded is None:
                        t_loaded = now                           # keys are in (:359): the timed loop starts (:386)
                    if line.startswith("Counting elapse"):
                        t_done = now
                        nk_ref = int(line.split("total")[1].split()[0])
            pr.kill()
            pr.wait()
            os.close(master)
            if t_loaded and t_done:
                # the threaded drain sleeps one fixed second (:469); it is not counting time
                sec = (t_done - t_loaded) - (1.0 if args.cpu_threads > 0 else 0.0)
                nb = len(host) - n_lines
                out = {"value": round(nb / sec, 1), "unit": "bases/s", "cores": args.cpu_threads + 1, "kind": "reference",
                       "sample": "the reference binary itself (oracle/_ref/quicKmer2 count -t %d, gcc -O3 of QuicKmer.c) on a .qm of this "
                                 "dictionary: first %d reads (%d bases) of batch 0 as FASTA, %.1f s between its 'Read 0x.. hash' and "
                                 "'Counting elapse' lines minus the fixed sleep(1) of its drain; %d k-mers (port on the same reads: %s)"
                                 % (args.cpu_threads, n_lines, nb, sec, nk_ref,
                                    "identical" if nk_ref == ol.count_lines(hk[:1], k, b"", threads=0)[1]["n_kmers"] + ol.extract(host, k)[0].size else "DIFFERENT"),
                       "host_cores": ncpu, "dict_load_s": round(t_loaded - tA, 1), "port": port}
            else:
                out["reference_binary"] = "did not reach its 'Counting elapse' line"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    del hk
    return out


def part_like(p1, p0):
    """the timed steps ran the partitioned pipeline (the realistic batch is only meaningful on the same path)"""
    return p1["partitioned_batches"] - p0["partitioned_batches"] > 0


def sum64(torch, counters, piece=1 << 28):
    """Sum of a large int32 tensor in int64 without a full-size int64 temporary."""
    return sum(int(counters[a:a + piece].sum(dtype=torch.int64)) for a in range(0, counters.numel(), piece))


def shard_spans_device(pyqm2, np, keys, H, G):
    """qm2_shard_span needs host keys but only looks at the slots around the shard edges: give it a sparse
    host image (untouched pages stay unmapped) with windows copied from the device table."""
    W = 1 << 18
    sparse = np.zeros(H, dtype=np.uint64)
    per = H // G
    for g in range(1, G):
        e = g * per
        sparse[e - W:e + W] = keys[e - W:e + W].cpu().numpy().view(np.uint64)
    spans = [pyqm2.shard_span(sparse, G, g) for g in range(G)]
    for own_lo, own_hi, lo, hi in spans:
        assert own_lo - lo < W - 64 and hi - own_hi < W - 64, "a run of occupied slots longer than the copied window"
    return spans


def sharded_bench(args, wl, pyqm2, torch, np, dev, local, rank, world, use_dist, keys, batches, n_batches, H, k, n_unique,
                  mode_flag, t_setup):
    """BASELINE config 4.  N ranks: quick-mer2_amd/multi.py's ShardedJob (route -> asynchronous all-to-all of
    fixed slabs over RCCL -> probe of the rank's span, overlapped with the next batch's encode).
    N = 1: the same two device legs on --shards logical shards of the one GPU, the exchange a no-op —
    what hash-range sharding costs per base against a replica, before an 8-GPU box ever sees it."""
    import ctypes as _C
    G = world if world > 1 else args.shards
    spans = shard_spans_device(pyqm2, np, keys, H, G)
    flags = pyqm2.FLAG_TIMING | mode_flag
    max_bytes = max(b[1] for b in batches)
    t_class = {}

    def add_prof(p0, p1):
        for c in p1["class_ms"]:
            t_class[c] = t_class.get(c, 0.0) + p1["class_ms"][c] - p0["class_ms"][c]

    job_path = world > 1 or (use_dist and os.environ.get("QM2_SHARDED_JOB") == "1")     # (1-rank rehearsal of the N>1 path)
    if job_path:
        G = world
        spans = shard_spans_device(pyqm2, np, keys, H, G) if G > 1 else [(0, H, 0, H)]
        import torch.distributed as dist
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        ge.load_package()
        from quick_mer2_amd import multi
        eng = multi.HipEngine(torch, dev)
        own_lo, own_hi, lo, hi = spans[rank]
        job = multi.ShardedJob(dist, eng, None, k, torch, batch_bytes=max_bytes, flags=flags,
                               resident=(H, spans[rank], keys.data_ptr() + 8 * lo))
        job.ctx.prepare(job.cap * world)
        for i in range(args.warmup):
            job.count_local(batches[i % n_batches][0])
        eng.drain(job.ctx)
        _, s0 = job.ctx.finish(want_depth=False)
        p0, r0 = job.ctx.profile(), job.rctx.profile()
        acc0 = [int(x) for x in job._acc] if job._acc is not None else [0, 0, 0]
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            job.count_local(batches[(args.warmup + i) % n_batches][0])
        eng.drain(job.ctx)
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        _, s1 = job.ctx.finish(want_depth=False)
        add_prof(p0, job.ctx.profile())
        add_prof(r0, job.rctx.profile())
        tot = torch.tensor([int(job._acc[0]) - acc0[0], int(job._acc[1]) - acc0[1],
                            s1["n_probes"] - s0["n_probes"], s1["n_hits"] - s0["n_hits"],
                            sum64(torch, job.counters), s1["n_hits"]], dtype=torch.int64, device=dev)
        dist.all_reduce(tot)
        n_bases, n_kmers, n_probes, n_hits, csum, hsum = [int(x) for x in tot]
        conserved = csum == hsum
        spilled = job.ctx.profile()["spilled_kmers"]
        retries = job.retries
        par = "hash-range shards x%d: one per rank, k-mers routed by asynchronous RCCL all-to-all of fixed slabs" % world
        job.close()
    else:
        # the senders' 10-bit route (default): G owners x 1024 / G sub-slabs, so that what a shard receives is already the
        # first level of its own partition (qm2_probe_subslabs_device); --flat-slabs: one slab per owner, split again
        # on the receiving side (qm2_probe_slabs_device, rounds 1-2)
        n_subs = 1 if args.flat_slabs else max(1, 1024 // G)
        xcd = 8 if n_subs > 1 else 1        # the router's output in one sub-region per XCD and region (qm2_route_xcd_device)
        if n_subs > 1:
            mean = max_bytes * 0.85 / (G * n_subs * xcd)
            cap = (int(mean * 1.12 + 8 * mean ** 0.5) + 256) & ~1
        else:
            cap = (int(max_bytes / G * 1.3) + 4096) & ~1
        # ONE stream for the router and all the logical shards: their kernels then run one after the other and the
        # HIP-event time of a launch is its own run time.  (With a stream per context, as in round 2, the eight shards'
        # launches queued behind each other's kernels and every tiny spill-list probe was charged ~1.4 ms of waiting:
        # "probe 11.35 ms per step" in profiles/r02_c3/bench_sharded_8_logical_20M.json was that, not work.)
        one = torch.cuda.Stream(device=dev)
        router = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=pyqm2.FLAG_KEYS_LATER | flags, span=(0, 64),
                               stream=one.cuda_stream)
        shards = []
        for own_lo, own_hi, lo, hi in spans:
            cnt = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
            c = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=flags, ext_keys=keys.data_ptr() + 8 * lo,
                              ext_counters=cnt.data_ptr(), span=(lo, hi), own=(own_lo, own_hi), stream=one.cuda_stream)
            if n_subs == 1:
                c.prepare(cap)
            shards.append((c, cnt))
        slabs = torch.empty(xcd * G * n_subs * cap, dtype=torch.int64, device=dev)
        fill = torch.zeros(xcd * G * n_subs, dtype=torch.int32, device=dev)
        call = torch.zeros(3, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

        def step(i):
            buf, nbytes, _ = batches[i % n_batches]
            fill.zero_()
            torch.cuda.synchronize()
            if n_subs > 1:
                router.route_xcd_device(buf.data_ptr(), nbytes, slabs.data_ptr(), cap, G * n_subs, fill.data_ptr(), call.data_ptr())
            else:
                router.route_device(buf.data_ptr(), nbytes, slabs.data_ptr(), cap, G, fill.data_ptr(), call.data_ptr())
            router.sync()
            for g, (c, _) in enumerate(shards):
                if n_subs > 1:
                    # owner g's part of the x-major output: 8 "peers" (the sender's XCD sub-regions) G * n_subs regions apart
                    c.probe_subslabs_device(slabs.data_ptr() + 8 * g * n_subs * cap, cap, xcd, n_subs, fill.data_ptr() + 4 * g * n_subs,
                                            kmers_bound=int(nbytes * 0.85 / G * 1.1), peer_stride=G * n_subs)
                else:
                    c.probe_slabs_device(slabs.data_ptr() + 8 * g * cap, cap, 1, fill.data_ptr() + 4 * g)
            for c, _ in shards:
                c.sync()

        for i in range(args.warmup):
            step(i)
        call.zero_()
        s0 = [c.finish(want_depth=False)[1] for c, _ in shards]
        p0 = [c.profile() for c, _ in shards]
        r0 = router.profile()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        s1 = [c.finish(want_depth=False)[1] for c, _ in shards]
        for a, (c, _) in zip(p0, shards):
            add_prof(a, c.profile())
        add_prof(r0, router.profile())
        n_bases, n_kmers, unfit = [int(x) for x in call]
        assert unfit == 0
        n_probes = sum(b["n_probes"] - a["n_probes"] for a, b in zip(s0, s1))
        n_hits = sum(b["n_hits"] - a["n_hits"] for a, b in zip(s0, s1))
        conserved = sum(sum64(torch, cnt) for _, cnt in shards) == sum(b["n_hits"] for b in s1)
        spilled = sum(c.profile()["spilled_kmers"] for c, _ in shards)
        retries = 0
        par = ("hash-range shards x%d, LOGICAL on one GPU (route + one partitioned probe per shard; the exchange is a no-op); %s"
               % (G, "10-bit route: %d sub-slabs per owner, the receivers split once" % n_subs if n_subs > 1 else "one slab per owner, the receivers split twice"))
        for c, _ in shards:
            c.close()
        router.close()
    kern_ms = sum(t_class.values())
    b_survey = n_bases + 8 * n_probes + 4 * n_hits
    # (the shards' contexts run on their own streams and overlap: summed HIP-event times count that overlap
    #  twice, so the path figure is taken over the wall time of the steps)
    gbs = b_survey / max(1, world) / elapsed / 1e9
    return {
        "metric": "input bases/sec counted vs hg38 k=30 dict", "value": round(n_bases / elapsed, 1), "unit": "bases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": args.workload + ": " + wl["desc"], "k": k, "hash_size": H, "dict_kmers": n_unique,
                   "reads_per_step_per_gpu": wl["reads"], "parallelism": par, "shards": G,
                   "inputs": "resident in HBM", "setup_s": round(t_setup, 1), "counts_conserved": bool(conserved),
                   "batches_cut_and_rerouted": retries, "spilled_kmers_rank0_or_all_logical": spilled},
        "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                     "traffic": None, "what": "whole path per GPU: SURVEY 8(d) bytes / wall time of the timed steps (route + exchange + partition + count)",
                     "kernel_ms_per_step": {c: round(v / args.steps, 3) for c, v in t_class.items() if v},
                     "sum_kernel_ms_per_step_overlapping": round(kern_ms / args.steps, 3)},
        "cpu_baseline": None,
    }


def main():
    # native libraries (RCCL prints a version banner) write to fd 1: keep stdout for the ONE JSON line
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("QM2_WORKLOAD", "c3"), choices=sorted(WORKLOADS),
                    help="c3 (default) = the configuration the metric is quoted on: full hg38-scale dictionary on one GPU")
    ap.add_argument("--batches", type=int, default=4, help="distinct read batches kept resident and cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-binary", action="store_true", help="CPU leg: the oracle port only")
    ap.add_argument("--no-streamed", action="store_true", help="skip the host-fed (PCIe-inclusive) regions")
    ap.add_argument("--no-bind", action="store_true", help="do not restrict this process to the CPUs of the GPU's NUMA node")
    ap.add_argument("--no-realistic", action="store_true", help="skip the c3r batch (half the reads from elsewhere, 5 % from 1000 repeat families)")
    ap.add_argument("--no-random-access", action="store_true", help="skip the random-access microbenchmark (roofline.random_access)")
    ap.add_argument("--no-geometry-check", action="store_true",
                    help="skip the partitioned-vs-direct comparison of all counters at the benchmarked batch geometry")
    ap.add_argument("--streamed-steps", type=int, default=4, help="steps of the ASCII host-fed regions; the packed ones take twice as many (the same seconds)")
    ap.add_argument("--feed-threads", type=int, default=14, help="host threads that fill (and pack into) the staging buffers in the streamed regions")
    ap.add_argument("--piece-mib", type=int, default=32, help="streamed regions: MiB of text a feeding thread packs at a time")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the -t 6 port sample")
    ap.add_argument("--ref-seconds", type=float, default=15.0, help="target counting time of the reference binary")
    ap.add_argument("--sweep-seconds", type=float, default=6.0, help="target CPU time of each other worker count")
    ap.add_argument("--cpu-threads", type=int, default=6, help="probe workers (reference recommends 6)")
    ap.add_argument("--cpu-sweep", default="0,12,24", help="other worker counts timed on smaller samples "
                    "(0 = the reference's serial mode); results in cpu_baseline.sweep_bases_per_s_by_workers")
    ap.add_argument("--scratch", default=os.environ.get("TMPDIR", "/tmp"), help="where the reference binary's .qm goes")
    ap.add_argument("--reads-per-step", type=int, default=0, help="override the workload's reads per step")
    ap.add_argument("--parallelism", default="replicas", choices=["replicas", "sharded"],
                    help="replicas (default; what the dictionary's size calls for: it always fits one GPU) or sharded = "
                         "BASELINE config 4: the table range-partitioned by home slot over the ranks, k-mers routed by "
                         "all-to-all (N = 1: --shards logical shards on the one GPU, the exchange a no-op)")
    ap.add_argument("--shards", type=int, default=8, help="--parallelism sharded on one GPU: logical shards")
    ap.add_argument("--flat-slabs", action="store_true", help="--parallelism sharded on one GPU: one slab per owner (rounds 1-2) instead of the 10-bit route")
    ap.add_argument("--mode", default="auto", choices=["auto", "direct", "partition"],
                    help="count pipeline: auto = chosen per batch from its size vs. H")
    args = ap.parse_args()

    # --gpus N is the number of ranks.  Under torch.distributed.run (the driver's form for N > 1) the environment says the
    # same; without it, N > 1 starts the ranks itself; anything else is a mistake that must not print a 1-GPU line.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, json_out))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: the launcher's rank count and --gpus disagree"
                         % (args.gpus, os.environ.get("WORLD_SIZE")))

    import numpy as np
    import torch
    import pyqm2

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("QM2_FORCE_DIST") == "1"    # (1-rank rehearsal of the N>1 code path)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("QM2_REHEARSE_ON_ONE_GPU") == "1":
            # N ranks of this very script on ONE card (tests/test_gpu_parity.py::test_bench_replicas_two_ranks_on_one_gpu):
            # RCCL refuses two ranks on one device, so the rehearsal's collectives go through gloo
            local = 0
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the count path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # the host side of the fed regions (pageable copies of the batches, pinned staging, the packing threads) on the socket
    # this rank's GPU hangs off (a placement hint; --no-bind or QM2_NO_BIND=1: wherever the scheduler puts them)
    cpus_near = pyqm2.bind_near_device(local) if not args.no_bind else None
    wl = dict(WORKLOADS[args.workload])
    if args.reads_per_step:
        wl["reads"] = args.reads_per_step
        import re as _re
        wl["desc"] = _re.sub(r"\d+M x", "%.0fM x" % (args.reads_per_step / 1e6), wl["desc"])
    k, H, n = 30, 1 << wl["lgH"], wl["genome"]

    # ---- untimed set-up: reference, dictionary, read batches, all made on the device -------
    t_setup = time.time()
    genome = torch.empty(n, dtype=torch.uint8, device=dev)
    n_blocks, dups, covered = genome_layout(n)
    pyqm2.tools.genome_generate(genome.data_ptr(), n, seed=22, n_blocks=n_blocks, dups=dups)
    keys = torch.empty(H, dtype=torch.int64, device=dev)
    scratch = torch.empty(H, dtype=torch.int64, device=dev)
    n_distinct, n_unique = pyqm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    torch.cuda.empty_cache()
    counters = torch.zeros(H, dtype=torch.int32, device=dev)

    n_batches = max(1, min(args.batches, args.steps + args.warmup))
    batches = []
    for b in range(n_batches):
        first = (rank * 1000 + b) * wl["reads"]          # every rank / batch draws its own reads
        off = pyqm2.tools.read_offsets(wl["reads"], first, wl["len_lo"], wl["len_hi"], 22)
        nbytes = int(off[-1])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d_off = None
        if wl["len_lo"] != wl["len_hi"]:
            d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        pyqm2.tools.reads_generate(genome.data_ptr(), n, buf.data_ptr(), d_off.data_ptr() if d_off is not None else None,
                                   wl["reads"], first, wl["len_lo"], wl["len_hi"], 22, sub_rate=wl["sub"],
                                   sample_from=covered)
        batches.append((buf, nbytes, nbytes - wl["reads"]))
    torch.cuda.synchronize()
    del genome
    torch.cuda.empty_cache()
    t_setup = time.time() - t_setup

    mode_flag = {"auto": 0, "direct": pyqm2.FLAG_DIRECT, "partition": pyqm2.FLAG_PARTITION}[args.mode]
    if args.parallelism == "sharded":
        del counters
        out = sharded_bench(args, wl, pyqm2, torch, np, dev, local, rank, world, use_dist, keys, batches, n_batches, H, k,
                            n_unique, mode_flag, t_setup)
        if rank == 0:
            json_out.write(json.dumps(out) + "\n")
            json_out.flush()
        if use_dist:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    # what the CLI derives from the head of its input file (qm2_count.c, estimate_kmers_per_kib): k-mers per KiB of sequence
    # lines, 5 % on top — it sizes the partition buffers (a 60 M-read batch then fits beside the table and four resident batches)
    kib_hint = min(1023, int(1024.0 * (wl["len_lo"] - k + 1) / (wl["len_lo"] + 1) * 1.05 + 8)) if wl["len_lo"] == wl["len_hi"] else 0
    ctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=pyqm2.FLAG_TIMING | mode_flag,
                        ext_keys=keys.data_ptr(), ext_counters=counters.data_ptr(), kmers_per_kib=kib_hint)
    ctx.prepare(max(b[1] for b in batches))

    def barrier():
        if use_dist:
            dist.barrier()

    def step(i):
        buf, nbytes, _ = batches[i % n_batches]
        ctx.submit_device(buf.data_ptr(), nbytes)

    for i in range(args.warmup):
        step(i)
    ctx.sync()
    _, st0 = ctx.finish(want_depth=False)
    prof0 = ctx.profile()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    # Every rank completes its own counters first: the steps above end with qm2_sync, which leaves the counts of the
    # private bucket interiors in the context's 2-byte array; only qm2_finish (or qm2_counters_device /
    # qm2_merge_counters) folds them into the 4-byte array that is about to be summed (include/qm2.h, ABI 4).
    # Timed on its own (a 10 ms streaming pass at hg38 scale), outside the steps and outside the reduce.
    tf0 = time.perf_counter()
    _, st1 = ctx.finish(want_depth=False)
    torch.cuda.synchronize()
    fold_ms = 1e3 * (time.perf_counter() - tf0)

    # The replicas' ONE exchange, once per job (not per step): RCCL sum of the u32 counters onto
    # rank 0, which then owns the job's counts.  Timed on its own and reported beside the steps.
    reduce_ms = None
    if use_dist:
        barrier()
        tr0 = time.perf_counter()
        step_elems = 1 << 28                     # 1 GiB pieces: no 2^31-element count reaches the collective
        for a in range(0, counters.numel(), step_elems):
            if dist.get_backend() == "gloo":     # (the one-GPU rehearsal: gloo has no reduce of device tensors)
                dist.all_reduce(counters[a:a + step_elems])
            else:
                dist.reduce(counters[a:a + step_elems], dst=0)
        torch.cuda.synchronize()
        barrier()
        reduce_ms = 1e3 * (time.perf_counter() - tr0)

    try:                                     # diagnostic builds only (-DQM2_STAMPS): phase stamps of bucket_count
        import ctypes as _C
        fn = pyqm2.lib().qm2_debug_stamps
        acc = (_C.c_uint64 * 32)()
        fn(acc)
        nl, npw = max(1, acc[8]), max(1, acc[9])
        lt = sum(acc[i] for i in (0, 1, 2, 3, 10))
        ptot = sum(acc[i] for i in (4, 5, 6, 7, 11))
        sys.stderr.write("STAMPS loaders=%d cycles/wave=%.0f flush=%.1f%% wait_regs=%.1f%% stage=%.1f%% fetch_issue=%.1f%% barrier=%.1f%% | "
                         "probers=%d cycles/wave=%.0f kmer_arrival_wait=%.1f%% kmer_load_issue=%.1f%% phase_a=%.1f%% phase_b=%.1f%% barrier=%.1f%%\n"
                         % (nl, lt / nl, 100.0 * acc[0] / max(1, lt), 100.0 * acc[10] / max(1, lt), 100.0 * acc[1] / max(1, lt),
                            100.0 * acc[2] / max(1, lt), 100.0 * acc[3] / max(1, lt),
                            npw, ptot / npw, 100.0 * acc[11] / max(1, ptot), 100.0 * acc[4] / max(1, ptot), 100.0 * acc[5] / max(1, ptot),
                            100.0 * acc[6] / max(1, ptot), 100.0 * acc[7] / max(1, ptot)))
        if acc[20]:
            tot = sum(acc[16:20])
            sys.stderr.write("STAMPS split_text waves=%d kcycles/wave: window=%.0f zero+barrier=%.0f kmers+copy_out=%.0f sort=%.0f total=%.0f\n"
                             % (acc[20], acc[16] / acc[20] / 1e3, acc[17] / acc[20] / 1e3, acc[18] / acc[20] / 1e3, acc[19] / acc[20] / 1e3,
                                tot / acc[20] / 1e3))
        if acc[27]:
            w = acc[27]
            sys.stderr.write("STAMPS sort_chunk (split_text) waves=%d kcycles/wave: hist_barrier=%.0f scan=%.0f reserve+barrier=%.0f scatter=%.0f atomics_return=%.0f dl+barrier=%.0f\n"
                             % (w, acc[21] / w / 1e3, acc[22] / w / 1e3, acc[23] / w / 1e3, acc[24] / w / 1e3, acc[25] / w / 1e3, acc[26] / w / 1e3))
    except AttributeError:
        pass

    elapsed = t1 - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    prof1 = ctx.profile()
    d = {a: st1[a] - st0[a] for a in st1}
    bases_rank = d["n_bases"]
    total_bases = bases_rank
    if use_dist:
        tb = torch.tensor([bases_rank], dtype=torch.int64, device=dev)
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    assert bases_rank == sum(batches[(args.warmup + i) % n_batches][2] for i in range(args.steps))
    # conservation check on the merged result: every hit of every rank is exactly one count
    hits_all = st1["n_hits"]
    if use_dist:
        th = torch.tensor([hits_all], dtype=torch.int64, device=dev)
        dist.all_reduce(th)
        hits_all = int(th.item())
    counts_conserved = None
    if rank == 0:
        piece = 1 << 28                      # (summed in pieces: torch widens the operand of an int64 sum)
        total = sum(int(counters[a:a + piece].sum(dtype=torch.int64)) for a in range(0, counters.numel(), piece))
        counts_conserved = bool(total == hits_all)
        if not counts_conserved:
            sys.stderr.write("COUNTS NOT CONSERVED: sum of the merged counters %d, hits of all ranks %d, this rank: %s, profile %s\n"
                             % (total, hits_all, st1, prof1))

    # ---- parity at the benchmarked geometry: batch 0 once more through the partitioned pipeline (the plan
    #      the steps above ran with: sub-regions, bucket capacities, buffers all sized for THIS batch) and once
    #      through the direct kernel, which shares nothing with it but the front end; all 2^lgH counters are
    #      compared through a position-weighted 64-bit checksum (a k-mer lost, doubled or counted on another
    #      slot changes it) plus the four statistics.  Every k-mer probed exactly once: QuicKmer.c:283-293. ----
    geometry = None
    if rank == 0 and not args.no_geometry_check and prof1["partitioned_batches"] - prof0["partitioned_batches"] > 0:
        gbuf, gbytes, gbases = batches[0]
        ctx.reset()
        ctx.submit_device(gbuf.data_ptr(), gbytes)
        _, gp = ctx.finish(want_depth=False)
        gprof = ctx.profile()
        cs_p = pyqm2.tools.checksum_u32(counters.data_ptr(), H)
        ctx.reset()
        dctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=pyqm2.FLAG_DIRECT, ext_keys=keys.data_ptr(),
                             ext_counters=counters.data_ptr())
        dctx.submit_device(gbuf.data_ptr(), gbytes)
        _, gd = dctx.finish(want_depth=False)
        dctx.close()
        cs_d = pyqm2.tools.checksum_u32(counters.data_ptr(), H)
        geometry = {"equal": bool(cs_p == cs_d and gp == gd and gprof["partitioned_batches"] == 1
                                  and gprof["fallback_batches"] == 0 and gp["n_bases"] == gbases),
                    "bytes": gbytes, "partitioned": {"checksum": "%016x" % cs_p[0], "sum": cs_p[1], "nonzero_slots": cs_p[2], **gp},
                    "direct": {"checksum": "%016x" % cs_d[0], "sum": cs_d[1], "nonzero_slots": cs_d[2], **gd}}
        assert geometry["equal"], geometry

    # ---- a less friendly batch beside the friendly one: the steps above read reads drawn from the dictionary's own
    #      genome (hit fraction 0.85); real WGS against a unique-k-mer dictionary hits about half and has repeat
    #      families.  c3r = the last resident batch with every second read replaced by random sequence (what an unrelated genome's
    #      reads are to this table: no hits) and 5 % of the reads replaced by copies of 1000 reads (every k-mer of those occurs
    #      ~3000 times per batch: the spill list's case, tools/heavy_hitters.py).  Same table, same context, same plan. ----
    realistic = None
    if rank == 0 and world == 1 and not args.no_realistic and n_batches >= 3 and wl["len_lo"] == wl["len_hi"] and part_like(prof1, prof0):
        rl_, m_ = wl["len_lo"], wl["reads"]
        rbuf, rbytes, rbases = batches[n_batches - 1]
        rows = rbuf.view(m_, rl_ + 1)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        piece = 2_000_000
        for a in range(1, m_, 2 * piece):                       # rows a, a + 2, ...: random reads
            idx = torch.arange(a, min(m_, a + 2 * piece), 2, device=dev)
            rows[idx, :rl_] = lut[torch.randint(0, 4, (idx.numel(), rl_), device=dev, generator=g, dtype=torch.int64)]
            del idx
        fam = rows[torch.arange(0, 2000, 2, device=dev)].clone()      # 1000 reads of the genome's own (even rows)
        n_rep = m_ // 20
        where = torch.randint(0, m_, (n_rep,), device=dev, generator=g)
        rows[where] = fam[torch.randint(0, 1000, (n_rep,), device=dev, generator=g)]
        del where, fam
        torch.cuda.synchronize()
        ctx.reset()
        ctx.submit_device(rbuf.data_ptr(), rbytes)                    # warm-up + this batch's own statistics
        _, rs = ctx.finish(want_depth=False)
        rp0 = ctx.profile()
        cs_p = pyqm2.tools.checksum_u32(counters.data_ptr(), H)
        r_steps = max(3, args.steps // 4)
        torch.cuda.synchronize()
        tA = time.perf_counter()
        for _ in range(r_steps):
            ctx.submit_device(rbuf.data_ptr(), rbytes)
        ctx.sync()
        torch.cuda.synchronize()
        r_sec = time.perf_counter() - tA
        rp1 = ctx.profile()
        ctx.reset()
        dctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=pyqm2.FLAG_DIRECT, ext_keys=keys.data_ptr(),
                             ext_counters=counters.data_ptr())
        dctx.submit_device(rbuf.data_ptr(), rbytes)
        _, rd = dctx.finish(want_depth=False)
        dctx.close()
        cs_d = pyqm2.tools.checksum_u32(counters.data_ptr(), H)
        ctx.reset()
        realistic = {"bases_per_s": round(r_steps * rs["n_bases"] / r_sec, 1), "ms_per_step": round(1e3 * r_sec / r_steps, 3), "steps": r_steps,
                     "what": "c3r: the same table and plan; every second read random sequence (an unrelated genome's reads: no hits), 5 %% of the reads "
                             "copies of 1000 reads (repeat families: ~%d copies of each of their k-mers per batch)" % (n_rep // 1000),
                     "hit_fraction": round(rs["n_hits"] / max(1, rs["n_kmers"]), 4), "probes_per_kmer": round(rs["n_probes"] / max(1, rs["n_kmers"]), 4),
                     "spilled_kmers_per_step": (rp1["spilled_kmers"] - rp0["spilled_kmers"]) // r_steps,
                     "fallback_batches": rp1["fallback_batches"] - rp0["fallback_batches"],
                     "kernel_ms_per_step": {c: round((rp1["class_ms"][c] - rp0["class_ms"][c]) / r_steps, 3) for c in rp1["class_ms"]
                                            if rp1["class_ms"][c] - rp0["class_ms"][c] > 0.05 * r_steps},
                     "partitioned_equals_direct": bool(cs_p == cs_d and rs == rd),
                     "checksum": "%016x" % cs_p[0]}
        assert realistic["partitioned_equals_direct"], (realistic, cs_d, rd)

    # ---- roofline: per-kernel HIP-event times (recorded on the stream the kernels run on) -----
    kern_ms = prof1["kernel_ms"] - prof0["kernel_ms"]
    cls_ms = {c: prof1["class_ms"][c] - prof0["class_ms"][c] for c in prof1["class_ms"]}
    cls_n = {c: prof1["class_launches"][c] - prof0["class_launches"][c] for c in prof1["class_launches"]}
    part_batches = prof1["partitioned_batches"] - prof0["partitioned_batches"]
    b_survey = d["n_bases"] + 8 * d["n_probes"] + 4 * d["n_hits"]        # SURVEY 8(d), whole timed region
    levels_kmer = (cls_n["split_kmer"] // max(1, part_batches)) if part_batches else 0
    # bytes each kernel class must move over the timed region by ITS OWN model
    # (the finest level and the table pass hold bucket records, DESIGN.md section 3: 6 bytes from 2^30 slots on, else 8)
    # (qm2_device.hip.h rec_ext_bits: slot bits 12 + 2 code bits + the hash bits above log2 H (+ 2 per base beyond 30) fit 16)
    rec = 6 if 12 + 2 + (32 - wl["lgH"]) + (2 * (k - 30) if k > 30 else 0) <= 16 else 8
    alg = {
        "count_direct": b_survey if part_batches == 0 else 0,
        # text in, k-mers out (records, when the text level is the finest one)
        "split_text": d["n_bases"] + (8 if levels_kmer else rec) * d["n_kmers"],
        # k-mers in + out per level, the last one writing records
        "split_kmer": (16 * (levels_kmer - 1) + 8 + rec) * d["n_kmers"] if levels_kmer else 0,
        # records in; per launch the table image of every slot + its 2-byte interior counter read and written
        "bucket_count": (rec * d["n_kmers"] + args.steps * (rec * H + 4 * H)) if part_batches else 0,
        "probe": 0, "encode": 0, "lines": 0,
    }
    # HBM bytes per launch measured by PMC counters: written by tools/summarize_profile.py from the rocprofv3
    # passes named in the file (profiles/<round>_<workload>/); absent = null
    traffic_step, traffic_src, traffic_k, valu_k, lds_k, traffic_stale = None, None, {}, {}, {}, None
    tr_file = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tr_file):
        try:
            tj = json.load(open(tr_file))
            if tj.get("kernel_source_sha16") != kernel_source_sha16():
                # measured on other kernels than the ones that just ran: not quoted
                traffic_stale = "profiles/traffic_%s.json was measured on kernel sources %s, this tree is %s: not quoted" % (
                    args.workload, tj.get("kernel_source_sha16"), kernel_source_sha16())
            else:
                traffic_k = {c: v.get("hbm_bytes_per_launch") for c, v in tj.get("per_kernel", {}).items()}
                valu_k = {c: v.get("valu_wave_insts_per_launch") for c, v in tj.get("per_kernel", {}).items()}
                lds_k = {c: {q: v.get(q) for q in ("lds_insts_per_launch", "lds_bank_conflict_cycles_per_launch", "lds_active_cycles_per_launch")}
                         for c, v in tj.get("per_kernel", {}).items()}
                traffic_step = tj.get("hbm_bytes_per_step")
                traffic_src = tj.get("source")
        except Exception:
            pass
    dom = max(cls_ms, key=lambda c: cls_ms[c])
    per_kernel = {}
    for c in cls_ms:
        if not cls_n[c]:
            continue
        ms = cls_ms[c] / cls_n[c]
        per_kernel[c] = {"launches": cls_n[c], "avg_launch_ms": round(ms, 4), "ms_per_step": round(cls_ms[c] / args.steps, 4),
                         "kernel_model_bytes_per_launch": alg[c] // cls_n[c],
                         "kernel_model_GBs": round(alg[c] / cls_n[c] / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                         "traffic": traffic_k.get(c)}
    pipe_gbs = b_survey / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    dom_ms = cls_ms[dom] / max(1, cls_n[dom])
    dom_survey = b_survey / args.steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(pipe_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(pipe_gbs / HBM_PEAK_GBS, 5), "traffic": traffic_step, "traffic_source": traffic_src or traffic_stale,
                "what": "whole path: SURVEY 8(d) bytes (1 B/base + 8 B/slot inspected + 4 B/hit) of a step / summed kernel time of a step",
                "algorithmic_bytes_per_step": b_survey // args.steps,
                "bytes_per_base": round(b_survey / max(1, d["n_bases"]), 3),
                "sum_kernel_ms_per_step": round(kern_ms / args.steps, 4),
                "mode": "partitioned" if part_batches else "direct",
                "dominant_kernel": {"kernel": "qm2::%s_kernel" % dom, "avg_launch_ms": round(dom_ms, 4),
                                    "survey_bytes_over_its_time_GBs": round(dom_survey, 1),
                                    "survey_frac": round(dom_survey / HBM_PEAK_GBS, 5),
                                    "kernel_model_frac": round((per_kernel[dom]["kernel_model_GBs"] or 0) / HBM_PEAK_GBS, 5)},
                "kernels": per_kernel,
                "kernel_only_bases_per_s": round(d["n_bases"] / (kern_ms * 1e-3), 1) if kern_ms > 0 else None}

    ctx.close()
    ctx = None
    del counters
    torch.cuda.empty_cache()

    # ---- the instruction side (what the byte fraction above is bound BY: the three kernels' time x CUs is constant, EXPERIMENTS.md 4.1.3):
    #      VALU wave-instructions per launch (SQ_INSTS_VALU, a PMC pass of tools/profile.sh on these very kernel sources)
    #      against what the chip can issue in the launch's measured time ---------------------------------------------
    if any(valu_k.get(c) for c in per_kernel):
        clk_hz = 2.4e9                                                      # MI355X max clock (MI355X_MICROARCH.md); under load the chip holds less, so the fraction reads low
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        issue = n_cu * VALU_WAVE_INSTS_PER_CU_CYCLE * clk_hz               # wave64 VALU instructions per second, whole chip
        vk, tot_i, tot_s = {}, 0.0, 0.0
        for c, e in per_kernel.items():
            if not valu_k.get(c):
                continue
            sec = e["avg_launch_ms"] * 1e-3
            kmers_launch = d["n_kmers"] / max(1, cls_n[c])
            vk[c] = {"valu_wave_insts_per_launch": valu_k[c], "lane_ops_per_kmer": round(64.0 * valu_k[c] / max(1.0, kmers_launch), 1),
                     "issue_frac": round(valu_k[c] / (sec * issue), 4), **{q: v for q, v in (lds_k.get(c) or {}).items() if v is not None}}
            if vk[c].get("lds_active_cycles_per_launch"):
                vk[c]["lds_conflict_share_of_active_cycles"] = round(vk[c]["lds_bank_conflict_cycles_per_launch"] / vk[c]["lds_active_cycles_per_launch"], 3)
            tot_i += valu_k[c] * cls_n[c] / args.steps
            tot_s += sec * cls_n[c] / args.steps
        roofline["valu"] = {"bound": "valu issue", "unit": "wave64 VALU instructions/s", "peak": issue, "peak_is": "%d CUs x 4 SIMDs x 1 wave64 instruction per 4 cycles x %.1f GHz (the max clock)" % (n_cu, clk_hz / 1e9),
                            "achieved": round(tot_i / tot_s, 1) if tot_s else None, "frac": round(tot_i / tot_s / issue, 4) if tot_s else None,
                            "lane_ops_per_kmer_whole_path": round(64.0 * tot_i / max(1.0, d["n_kmers"] / args.steps), 1),
                            "kernels": vk, "source": traffic_src}
    else:
        roofline["valu"] = None

    # ---- north_star's other denominator: the HBM RANDOM-ACCESS roofline = what this device does with uniformly
    #      random 8-byte loads + 4-byte atomic adds over arrays of this table's size (the direct kernel's access pattern,
    #      one of each per k-mer at least), measured here and now; and B_sector, the bytes that pattern moves in 64-byte sectors ----
    if rank == 0 and not args.no_random_access:
        try:
            n_ops = 1 << 29
            ms = {m: pyqm2.tools.microbench(H, n_ops, m, repeats=2) for m in (0, 1, 2)}
            ceiling = n_ops / (ms[2] * 1e-3)
            kmers_per_s = d["n_kmers"] / elapsed / max(1, world)
            crossings = (d["n_probes"] - d["n_kmers"]) / 8.0            # a walk's next slot lies in the next 64-byte line one time in eight
            b_sector = 64.0 * (d["n_kmers"] + crossings) + 128.0 * d["n_hits"]
            roofline["random_access"] = {
                "what": "uniformly random 8-byte load + 4-byte atomic add per operation over arrays of the table's size (qm2t_microbench, 2^29 operations, 16 in flight per lane): "
                        "the ceiling of one sector gather + one memory-side atomic per k-mer, which the partitioned pipeline replaces by streams",
                "slots": H, "load8_Gops_per_s": round(n_ops / ms[0] / 1e6, 2), "atomic4_Gops_per_s": round(n_ops / ms[1] / 1e6, 2),
                "load8_atomic4_Gops_per_s": round(ceiling / 1e9, 2),
                "kmers_per_s": round(kmers_per_s, 1), "kmers_per_s_over_the_ceiling": round(kmers_per_s / ceiling, 2),
                "B_sector_bytes_per_step": int(b_sector / args.steps),
                "B_sector_GBs": round(b_sector / max(1, world) / elapsed / 1e9, 1),
                "B_sector_frac_of_peak": round(b_sector / max(1, world) / elapsed / 1e9 / HBM_PEAK_GBS, 4)}
        except Exception as e:                                          # (no room for the microbenchmark's own 48 GiB: say so)
            roofline["random_access"] = {"error": str(e)}
        torch.cuda.empty_cache()

    # ---- SURVEY 8(d)'s wall-clock metric: the batches fed from HOST memory through the pinned staging
    #      buffers, first commit -> return of qm2_finish, allocations made beforehand ---------------------
    streamed = None
    if not args.no_streamed and wl["len_lo"] == wl["len_hi"]:
        # N > 1: every rank feeds its own GPU from host memory AT THE SAME TIME (one barrier in front of the timed
        # region); the aggregate is all ranks' bases over the slowest rank's time — what N x PCIe + the host's
        # memory system give together, which a single-rank figure times N would only guess
        rl = wl["len_lo"]
        nb = 2
        hosts = [batches[i % n_batches][0].cpu().numpy() for i in range(nb)]
        step_bytes = max(b[1] for b in batches)
        batches = batches[:1]                                         # (batch 0 stays for the FASTQ text and the CPU legs)
        torch.cuda.empty_cache()

        def run(sctx, feed, steps):
            feed(0)                                                   # warm-up: page in, first-touch
            sctx.finish(want_depth=False)
            a0 = sctx.profile()["allocations"]
            _, s0 = sctx.finish(want_depth=False)
            barrier()
            tA = time.perf_counter()
            moved = 0
            for i in range(steps):
                moved += feed(i)
            _, s1 = sctx.finish(want_depth=False)
            sec = time.perf_counter() - tA
            nbases, allocs = s1["n_bases"] - s0["n_bases"], sctx.profile()["allocations"] - a0
            per_rank = [round(moved / sec / 1e9, 2)]
            if use_dist:                                              # sums, the slowest rank's time, every rank's own rate
                tot = torch.tensor([nbases, moved, allocs], dtype=torch.int64, device=dev)
                dist.all_reduce(tot)
                tmax = torch.tensor([sec], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                rates = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
                dist.all_gather(rates, torch.tensor([moved / sec / 1e9], dtype=torch.float64, device=dev))
                nbases, moved, allocs, sec = int(tot[0]), int(tot[1]), int(tot[2]), float(tmax.item())
                per_rank = [round(float(r.item()), 2) for r in rates]
            return nbases, moved, sec, allocs, per_rank

        # (1) PACKED (ABI 6, include/qm2.h "packed sequence"): the threads that fill the pinned buffers pack what they copy —
        # the reset test and the 2-bit code of every character, QuicKmer.c:403-411, nothing more — so 0.375 bytes per base
        # cross the link; this is what qm2_submit and the CLI's reader threads do by default.  A context of its own: its
        # device batches hold blocks only (QM2_PREPARE_PACKED: 3/8 of the bytes), which leaves room for batches as large as
        # the resident steps' (the table stream is a fixed cost per batch).
        pctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=mode_flag, ext_keys=keys.data_ptr(), kmers_per_kib=kib_hint,
                             n_staging=max(14, args.feed_threads + 4), batch_bytes=(step_bytes + 63) & ~63)
        pctx.prepare(host=True, packed=True)

        def feed_packed(i):
            h = hosts[i % nb]
            pctx.submit_threads(None, args.feed_threads, addr=h.ctypes.data, n=h.size)     # ONE call of the C-ABI per step (qm2_submit_threads)
            return h.size * 24 // 64                                                        # (the blocks' bytes, to a piece's rounding; checked against the profile below)

        # ONE JOB of BASELINE.md's size per rank: a 30x human genome = 600 M reads of 150 bases (tutorial.md:146-162) -> 10 steps of the C3 workload;
        # never fewer than twice the ASCII region's steps (chr22-scale steps are small)
        job_reads = int(os.environ.get("QM2_BENCH_JOB_READS", "600000000"))
        packed_steps = max(2 * args.streamed_steps, min(40, -(-job_reads // max(1, wl["reads"]))))
        nbases, moved, sec, allocs, per_rank = run(pctx, feed_packed, packed_steps)
        assert nbases == world * packed_steps * wl["reads"] * rl
        staged_total = pctx.profile()["staged_bytes"]          # (warm-up step included: packed_steps + 1 steps of the same size)
        assert abs(staged_total / (packed_steps + 1) * packed_steps - moved / world) < 1e-3 * moved, (staged_total, moved)
        streamed = {"streamed_bases_per_s": round(nbases / sec, 1), "h2d_GBps": round(moved / sec / 1e9, 2),
                    "h2d_GBps_per_rank": per_rank, "ranks_feeding_concurrently": world,
                    "bytes_per_base_over_the_link": round(moved / max(1, nbases), 4),
                    "steps": packed_steps, "seconds": round(sec, 3), "allocations_inside": allocs,
                    "device_batch_positions": (step_bytes + 63) & ~63,
                    "pack_isa": pyqm2.lib().qm2_pack_isa().decode(), "cpus_of_the_gpus_numa_node_bound_to": cpus_near,
                    "feed": "qm2_submit_threads(ctx, lines, n, %d): one call per step on a 9 GB buffer of sequence lines in pageable memory; the library's threads pack "
                            "32 MiB pieces into the pinned staging buffers (qm2_pack_lines: 2-bit codes + reset flags, what the CLI's reader threads do behind pread), "
                            "the calling thread commits them as they get ready; timed from the first call to the return of qm2_finish; the region is ONE JOB of "
                            "BASELINE.md's size per rank (a 30x human genome, 600 M reads of 150 bases) where the workload's steps allow" % args.feed_threads,
                    "job_reads": packed_steps * wl["reads"]}
        # ... and what the kernels make of packed input that is already in HBM (the ceiling of the figure above; `value` stays
        # the ASCII-resident rate: there the device does the whole of QuicKmer.c:403-420)
        if rank == 0:
            blocks, pk, _ = pyqm2.pack_lines(memoryview(hosts[0]))
            d_blocks = torch.from_numpy(blocks.view(np.int64)).to(dev)
            del blocks
            for _ in range(2):
                pctx.submit_packed_device(d_blocks.data_ptr(), pk)
            pctx.sync()
            _, r0 = pctx.finish(want_depth=False)
            pr0 = pctx.profile()
            tA = time.perf_counter()
            for _ in range(args.streamed_steps):
                pctx.submit_packed_device(d_blocks.data_ptr(), pk)
            pctx.sync()
            sec_r = time.perf_counter() - tA
            _, r1 = pctx.finish(want_depth=False)
            pr1 = pctx.profile()
            assert r1["n_bases"] - r0["n_bases"] == args.streamed_steps * wl["reads"] * rl
            streamed["packed_resident_bases_per_s"] = round((r1["n_bases"] - r0["n_bases"]) / sec_r, 1)
            streamed["packed_resident"] = {"partitioned_batches": pr1["partitioned_batches"] - pr0["partitioned_batches"],
                                           "direct_batches": pr1["direct_batches"] - pr0["direct_batches"]}
            del d_blocks
        pctx.close()
        torch.cuda.empty_cache()

        # (2) the same lines as ASCII, 1 byte per base (QM2_FLAG_SUBMIT_ASCII / QM2_READER=lines; rounds 1-4's streamed figure),
        # and (3) FASTQ text, packed by the feeding threads and as raw text: one context with the library's default batches
        sctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=mode_flag, ext_keys=keys.data_ptr(),
                             n_staging=max(14, args.feed_threads + 4))
        sctx.prepare(host=True, raw=True)

        def feed_lines(i):
            h = hosts[i % nb]
            sctx.feed_host(h.ctypes.data, h.size, rl + 1, threads=args.feed_threads)
            return h.size

        nbases, moved, sec, allocs, per_rank = run(sctx, feed_lines, args.streamed_steps)
        streamed.update({"streamed_ascii_bases_per_s": round(nbases / sec, 1), "ascii_h2d_GBps": round(moved / sec / 1e9, 2),
                         "ascii_h2d_GBps_per_rank": per_rank, "ascii_allocations_inside": allocs})
        # the same reads as raw FASTQ text ("@r" / seq / "+" / qual: 2 x 151 + 5 bytes a record), line discipline on the device
        del hosts
        m = min(wl["reads"], 20_000_000)
        src = batches[0][0][:m * (rl + 1)].view(m, rl + 1)
        rec = torch.empty((m, 2 * (rl + 1) + 5), dtype=torch.uint8, device=dev)
        rec[:, 0:3] = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device=dev)
        rec[:, 3:3 + rl + 1] = src
        rec[:, rl + 4:rl + 6] = torch.tensor(list(b"+\n"), dtype=torch.uint8, device=dev)
        rec[:, rl + 6:2 * rl + 6] = ord("I")
        rec[:, 2 * rl + 6] = 10
        fq = rec.cpu().numpy().reshape(-1)
        del rec, src
        torch.cuda.empty_cache()

        def feed_fastq(i):
            sctx.feed_host(fq.ctypes.data, fq.size, 2 * (rl + 1) + 5, raw=(True, 1), threads=args.feed_threads)
            return fq.size

        # packed by the feeding threads, line discipline included (qm2_pack_reads: headers, '+' and quality lines never
        # leave the host) ...
        def feed_fastq_packed(i):
            return sctx.feed_host_packed(fq.ctypes.data, fq.size, 2 * (rl + 1) + 5, reads=(True, 1), threads=args.feed_threads, piece_text=args.piece_mib << 20)

        raw_steps = max(1, args.streamed_steps * 3 // 2)      # (6 x 6.1 GB by default: the drain of the last batch is then < 10 % of the region)
        fq_steps = max(2 * raw_steps, min(40, -(-job_reads // max(1, m))))        # (the same job as FASTQ text: 30 steps of 20 M reads)
        nbases, moved, sec, allocs, per_rank = run(sctx, feed_fastq_packed, fq_steps)
        assert nbases == world * fq_steps * m * rl
        streamed.update({"streamed_fastq_bases_per_s": round(nbases / sec, 1), "fastq_h2d_GBps": round(moved / sec / 1e9, 2),
                         "fastq_text_GBps": round(world * fq_steps * fq.size / sec / 1e9, 2), "fastq_h2d_GBps_per_rank": per_rank,
                         "fastq_reads_per_step": m, "fastq_steps": fq_steps, "fastq_allocations_inside": allocs})
        # ... and as raw text over the link, line discipline on the device (QM2_READER=raw; rounds 2-4's figure)
        nbases, moved, sec, allocs, per_rank = run(sctx, feed_fastq, raw_steps)
        streamed.update({"streamed_raw_fastq_bases_per_s": round(nbases / sec, 1), "raw_fastq_h2d_GBps": round(moved / sec / 1e9, 2),
                         "raw_fastq_h2d_GBps_per_rank": per_rank,
                         "raw_fastq_reads_per_step": m, "raw_fastq_steps": raw_steps, "raw_fastq_allocations_inside": allocs,
                         "raw_verdict": sctx.raw_verdict()})
        assert nbases == world * raw_steps * m * rl and sctx.raw_verdict() == 0
        sctx.close()
        del fq
    batches = batches[:1]
    torch.cuda.empty_cache()

    # ---- CPU baselines (rank 0, N = 1): reference binary, oracle port, worker sweep -----------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_legs(args, wl, pyqm2, torch, keys, batches[0][0], batches[0][1], H, k, local, n_unique)

    # what a rank holds of HOST memory at its peak (the text buffers of the host-fed regions, the pinned staging): an N-rank run takes N of these
    import resource
    rss_gib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20
    if use_dist:
        tr = torch.tensor([rss_gib], dtype=torch.float64, device=dev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        rss_gib = float(tr.item())

    if rank == 0:
        out = {
            "metric": "input bases/sec counted vs hg38 k=30 dict",
            "value": round(total_bases / elapsed, 1),
            "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "value_is": "the count path with the read batches (ASCII sequence lines) already resident in HBM when the timed region starts "
                        "(the bench contract); SURVEY 8(d)'s host-fed wall-clock metric (first commit -> return of qm2_finish, PCIe "
                        "included, the host threads packing what they feed) is streamed_bases_per_s in this same line, FASTQ text: "
                        "streamed.streamed_fastq_bases_per_s; the same without the packing: streamed.streamed_ascii_bases_per_s / "
                        "streamed_raw_fastq_bases_per_s",
            "config": {"workload": args.workload + ": " + wl["desc"], "k": k, "hash_size": H,
                       "dict_kmers": n_unique, "fill": round(n_unique / H, 4),
                       "partitioned_equals_direct_at_bench_batch": geometry["equal"] if geometry else None,
                       "geometry_check": geometry,
                       "reads_per_step_per_gpu": wl["reads"], "bases_per_step_per_gpu": batches[0][2],
                       "kmers_per_base": round(d["n_kmers"] / max(1, d["n_bases"]), 4),
                       "probes_per_kmer": round(d["n_probes"] / max(1, d["n_kmers"]), 4),
                       "hit_fraction": round(d["n_hits"] / max(1, d["n_kmers"]), 4),
                       "parallelism": "replicas x%d (dictionary replicated, reads partitioned, one RCCL reduce per job)" % world
                       if world > 1 else "1 GPU", "inputs": "resident in HBM (value); host-fed rates under `streamed`",
                       "setup_s": round(t_setup, 1), "host_peak_rss_gib_per_rank": round(rss_gib, 1),
                       "final_reduce_ms": round(reduce_ms, 2) if reduce_ms is not None else None,
                       # the same steps with the job's per-job costs (the fold of the 2-byte interiors + the one exchange) charged to them in full (a 30x hg38 job is 10 such steps
                       # per GPU at N = 1 and fewer at N > 1: the exchange is per job, `value` is per step)
                       "bases_per_s_with_the_fold_and_the_reduce_charged_to_these_steps":
                           round(total_bases / (elapsed + (reduce_ms + fold_ms) / 1e3), 1) if reduce_ms is not None else None,
                       "fold_ms": round(fold_ms, 2),
                       "counts_conserved": counts_conserved, "spilled_kmers": prof1["spilled_kmers"] - prof0["spilled_kmers"]},
            "kernel_only_bases_per_s": roofline["kernel_only_bases_per_s"],
            "realistic_bases_per_s": realistic["bases_per_s"] if realistic else None,
            "realistic": realistic,
            "streamed_bases_per_s": streamed["streamed_bases_per_s"] if streamed else None,
            "h2d_GBps": streamed["h2d_GBps"] if streamed else None,
            "streamed": streamed,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
