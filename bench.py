#!/usr/bin/env python3
"""bench.py — input bases/sec counted by the MI355X `count` hot path.

One "step" = one pass of the hot path (fused encode + probe + increment kernel,
through the C-ABI qm2_submit_device) over one batch of synthetic reads that is
already resident in HBM.  Workloads (BASELINE.json `configs`):

  c3  (default; BASELINE.json's metric is quoted on the hg38 k=30 dictionary, which fits one
      GPU) hg38-like dictionary: 3.1 Gb synthetic reference, k=30, H=2^32, 2.27 G unique
      k-mers (32 GiB keys + 16 GiB u32 counters), batches of 40 M x 150 bp reads
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

WORKLOADS = {
    "c2": dict(genome=50_818_468, lgH=28, reads=10_000_000, len_lo=150, len_hi=150, sub=0.005,
               desc="chr22-like 50.8 Mb synthetic reference, k=30, H=2^28; 10M x 150 bp reads per step"),
    "c3": dict(genome=3_100_000_000, lgH=32, reads=40_000_000, len_lo=150, len_hi=150, sub=0.005,
               desc="hg38-like 3.1 Gb synthetic reference, k=30, H=2^32; 40M x 150 bp reads per step"),
    "c5": dict(genome=3_100_000_000, lgH=32, reads=240_000, len_lo=15_000, len_hi=20_000, sub=0.001,
               desc="hg38-like dictionary (H=2^32), HiFi-like reads U(15k,20k) bp, 240k reads per step"),
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=6, help="probe workers (reference recommends 6)")
    ap.add_argument("--cpu-sweep", default="", help="comma list of extra worker counts to time on a smaller "
                    "sample (e.g. 0,12,24; 0 = the reference's serial mode); results go to cpu_baseline.sweep")
    ap.add_argument("--reads-per-step", type=int, default=0, help="override the workload's reads per step")
    ap.add_argument("--mode", default="auto", choices=["auto", "direct", "partition"],
                    help="count pipeline: auto = chosen per batch from its size vs. H")
    args = ap.parse_args()

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
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the count path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
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
    t_setup = time.time() - t_setup

    mode_flag = {"auto": 0, "direct": pyqm2.FLAG_DIRECT, "partition": pyqm2.FLAG_PARTITION}[args.mode]
    ctx = pyqm2.Counter(None, hash_size=H, k=k, device=local, flags=pyqm2.FLAG_TIMING | mode_flag,
                        ext_keys=keys.data_ptr(), ext_counters=counters.data_ptr())

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

    # The replicas' ONE exchange, once per job (not per step): RCCL sum of the u32 counters onto
    # rank 0, which then owns the job's counts.  Timed on its own and reported beside the steps.
    reduce_ms = None
    if use_dist:
        tr0 = time.perf_counter()
        step_elems = 1 << 28                     # 1 GiB pieces: no 2^31-element count reaches the collective
        for a in range(0, counters.numel(), step_elems):
            dist.reduce(counters[a:a + step_elems], dst=0)
        torch.cuda.synchronize()
        barrier()
        reduce_ms = 1e3 * (time.perf_counter() - tr0)

    try:                                     # diagnostic builds only (-DQM2_STAMPS): phase stamps of bucket_count
        import ctypes as _C
        fn = pyqm2.lib().qm2_debug_stamps
        acc = (_C.c_uint64 * 16)()
        fn(acc)
        nl, npw = max(1, acc[8]), max(1, acc[9])
        lt = sum(acc[i] for i in (0, 1, 2, 3, 10))
        ptot = sum(acc[i] for i in (4, 5, 6, 7))
        sys.stderr.write("STAMPS loaders=%d cycles/wave=%.0f flush=%.1f%% wait_regs=%.1f%% stage=%.1f%% fetch_issue=%.1f%% barrier=%.1f%% | "
                         "probers=%d cycles/wave=%.0f kmer_loads=%.1f%% phase_a=%.1f%% phase_b=%.1f%% barrier=%.1f%%\n"
                         % (nl, lt / nl, 100.0 * acc[0] / max(1, lt), 100.0 * acc[10] / max(1, lt), 100.0 * acc[1] / max(1, lt),
                            100.0 * acc[2] / max(1, lt), 100.0 * acc[3] / max(1, lt),
                            npw, ptot / npw, 100.0 * acc[4] / max(1, ptot), 100.0 * acc[5] / max(1, ptot),
                            100.0 * acc[6] / max(1, ptot), 100.0 * acc[7] / max(1, ptot)))
    except AttributeError:
        pass

    elapsed = t1 - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    _, st1 = ctx.finish(want_depth=False)
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
        counts_conserved = bool(int(counters.sum(dtype=torch.int64)) == hits_all)

    # ---- roofline: per-kernel HIP-event times (recorded on the stream the kernels run on) -----
    batches_timed = prof1["n_launches"] - prof0["n_launches"]
    kern_ms = prof1["kernel_ms"] - prof0["kernel_ms"]
    cls_ms = {c: prof1["class_ms"][c] - prof0["class_ms"][c] for c in prof1["class_ms"]}
    cls_n = {c: prof1["class_launches"][c] - prof0["class_launches"][c] for c in prof1["class_launches"]}
    part_batches = prof1["partitioned_batches"] - prof0["partitioned_batches"]
    b_survey = d["n_bases"] + 8 * d["n_probes"] + 4 * d["n_hits"]        # SURVEY 8(d), whole timed region
    levels_kmer = (cls_n["split_kmer"] // max(1, part_batches)) if part_batches else 0
    # algorithmic bytes each kernel class must move over the timed region
    alg = {
        "count_direct": b_survey if part_batches == 0 else 0,
        "split_text": d["n_bases"] + 8 * d["n_kmers"],                     # text in, k-mers out
        "split_kmer": 16 * d["n_kmers"] * levels_kmer,                     # k-mers in + out, per level
        "bucket_count": (8 * d["n_kmers"] + args.steps * (8 * H + 8 * n_unique)) if part_batches else 0,
        "probe": 0, "encode": 0,
    }
    dom = max(cls_ms, key=lambda c: cls_ms[c])
    dom_ms = cls_ms[dom]
    dom_n = max(1, cls_n[dom])
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tr_file = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tr_file):
        try:
            tj = json.load(open(tr_file))
            traffic = tj.get("per_kernel", {}).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    pipe_gbs = b_survey / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "qm2::%s_kernel" % dom, "launches": cls_n[dom],
                "avg_launch_ms": round(dom_ms / dom_n, 4),
                "algorithmic_bytes_per_launch": alg[dom] // dom_n,
                "pipeline": {"mode": "partitioned" if part_batches else "direct",
                             "kernel_ms_per_step": {c: round(cls_ms[c] / args.steps, 4) for c in cls_ms if cls_n[c]},
                             "sum_kernel_ms_per_step": round(kern_ms / args.steps, 4),
                             "survey_bytes_per_base": round(b_survey / max(1, d["n_bases"]), 3),
                             "survey_algorithmic_GBs": round(pipe_gbs, 2),
                             "survey_frac_of_peak": round(pipe_gbs / HBM_PEAK_GBS, 5),
                             "kernel_only_bases_per_s": round(d["n_bases"] / (kern_ms * 1e-3), 1) if kern_ms > 0 else None}}

    # ---- CPU baseline: the oracle's reference-model threaded counter on a bounded sample ------
    cpu = None
    # the measured context and everything but batch 0 are released before the CPU leg opens its own
    ctx.close()
    ctx = None
    del counters
    batches = batches[:1]
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        hk = keys.cpu().numpy().view(np.uint64)
        buf, nbytes, _ = batches[0]
        line = wl["len_lo"] + 1 if wl["len_lo"] == wl["len_hi"] else None
        # calibrate on a small slice, then size the sample for ~cpu_seconds
        probe_bytes = min(nbytes, 400_000 * 151)
        host = buf[:probe_bytes].cpu().numpy().tobytes()
        host = host[:host.rfind(b"\n") + 1]
        tA = time.perf_counter()
        ol.count_lines(hk, k, host, threads=args.cpu_threads)
        rate = len(host) / max(1e-6, time.perf_counter() - tA)
        want = int(min(nbytes, max(len(host), rate * args.cpu_seconds)))
        host = buf[:want].cpu().numpy().tobytes()
        host = host[:host.rfind(b"\n") + 1]
        tA = time.perf_counter()
        cdepth, cst = ol.count_lines(hk, k, host, threads=args.cpu_threads)
        cpu_s = time.perf_counter() - tA
        cdepth = ol.remove_drain_padding(cdepth, hk, cst["n_kmers"])
        # the same sample through the GPU path must give the same counters (bit-exact)
        chk = pyqm2.Counter(None, hash_size=H, k=k, device=local, ext_keys=keys.data_ptr())
        chk.submit(host)
        gdepth, gst = chk.finish()
        chk.close()
        same = bool(np.array_equal(gdepth, cdepth) and gst["n_kmers"] == cst["n_kmers"] and gst["n_hits"] >= 0)
        cpu = {"value": round(cst["n_bases"] / cpu_s, 1), "unit": "bases/s", "cores": args.cpu_threads + 1,
               "kind": "port", "host_cores": os.cpu_count(),
               "sample": "first %d reads (%d bases) of batch 0, oracle threaded model: 1 producer + %d probe workers, %.1f s"
                         % (host.count(b"\n"), cst["n_bases"], args.cpu_threads, cpu_s),
               "gpu_counts_identical_on_sample": same}
        if args.cpu_sweep:
            sweep = {}
            sub = host[:max(1, len(host) // 4)]
            sub = sub[:sub.rfind(b"\n") + 1]
            for thr in [int(x) for x in args.cpu_sweep.split(",") if x != ""]:
                tA = time.perf_counter()
                _, sst = ol.count_lines(hk, k, sub if thr else sub[:max(1, len(sub) // 6)][:sub[:max(1, len(sub) // 6)].rfind(b"\n") + 1], threads=thr)
                sweep[str(thr)] = round(sst["n_bases"] / (time.perf_counter() - tA), 1)
            cpu["sweep_bases_per_s_by_workers"] = sweep
        del hk

    if rank == 0:
        out = {
            "metric": "input bases/sec counted vs hg38 k=30 dict",
            "value": round(total_bases / elapsed, 1),
            "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": args.workload + ": " + wl["desc"], "k": k, "hash_size": H,
                       "dict_kmers": n_unique, "fill": round(n_unique / H, 4),
                       "reads_per_step_per_gpu": wl["reads"], "bases_per_step_per_gpu": batches[0][2],
                       "kmers_per_base": round(d["n_kmers"] / max(1, d["n_bases"]), 4),
                       "probes_per_kmer": round(d["n_probes"] / max(1, d["n_kmers"]), 4),
                       "hit_fraction": round(d["n_hits"] / max(1, d["n_kmers"]), 4),
                       "parallelism": "replicas x%d (dictionary replicated, reads partitioned, one RCCL reduce per job)" % world
                       if world > 1 else "1 GPU", "inputs": "resident in HBM", "setup_s": round(t_setup, 1),
                       "final_reduce_ms": round(reduce_ms, 2) if reduce_ms is not None else None,
                       "counts_conserved": counts_conserved},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
