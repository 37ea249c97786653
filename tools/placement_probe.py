#!/usr/bin/env python3
"""Diagnostic: does the speed of split_text depend on WHERE the partition buffers land?

bench.py's split_text time differs between processes (21.4 / 25 / 28 ms at C3, 50 M reads per launch) and is
constant inside a process.  This script keeps one process, one dictionary and one read batch, and re-creates the
counting context several times with differently sized dummy allocations in between, so that the partition buffers
move.  Prints the per-kernel times of every trial.

    python tools/placement_probe.py [trials] [reads]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
    global T0
    T0 = time.time()
    import numpy as np
    import torch
    import bench
    import pyqm2
    dev = torch.device("cuda", 0)
    wl = dict(bench.WORKLOADS["c3"])
    wl["reads"] = reads
    k, H, n = 30, 1 << wl["lgH"], wl["genome"]
    genome = torch.empty(n, dtype=torch.uint8, device=dev)
    n_blocks, dups, covered = bench.genome_layout(n)
    pyqm2.tools.genome_generate(genome.data_ptr(), n, seed=22, n_blocks=n_blocks, dups=dups)
    keys = torch.empty(H, dtype=torch.int64, device=dev)
    scratch = torch.empty(H, dtype=torch.int64, device=dev)
    pyqm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
    del scratch
    torch.cuda.empty_cache()
    counters = torch.zeros(H, dtype=torch.int32, device=dev)
    n_batches = int(os.environ.get("PROBE_BATCHES", "4"))
    batches = []
    for b in range(n_batches):
        first = b * reads
        off = pyqm2.tools.read_offsets(reads, first, wl["len_lo"], wl["len_hi"], 22)
        nbytes = int(off[-1])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        pyqm2.tools.reads_generate(genome.data_ptr(), n, buf.data_ptr(), None, reads, first, wl["len_lo"], wl["len_hi"], 22,
                                   sub_rate=wl["sub"], sample_from=covered)
        batches.append(buf)
        print("batch %d at 0x%x (%d B)" % (b, buf.data_ptr(), nbytes), flush=True)
    torch.cuda.synchronize()
    del genome
    torch.cuda.empty_cache()
    dummies = []
    sizes = [0, 1 << 20, (1 << 30) + 4096, 3 * (1 << 29) + 12345 * 256, 1 << 21, (5 << 30) + (1 << 16)]
    for t in range(trials):
        sz = sizes[t % len(sizes)]
        if sz:
            dummies.append(torch.empty(sz, dtype=torch.uint8, device=dev))
        ctx = pyqm2.Counter(None, hash_size=H, k=k, device=0, flags=pyqm2.FLAG_TIMING, ext_keys=keys.data_ptr(),
                            ext_counters=counters.data_ptr())
        ctx.prepare(nbytes)
        ctx.submit_device(batches[0].data_ptr(), nbytes)
        ctx.sync()
        for r in range(int(os.environ.get("PROBE_SERIES", "0"))):
            p0 = ctx.profile()
            for i in range(5):
                ctx.submit_device(batches[i % n_batches].data_ptr(), nbytes)
            ctx.sync()
            p1 = ctx.profile()
            print("   t=%.1f s series %d: %s" % (time.time() - T0, r, {c[:7]: round((p1["class_ms"][c] - p0["class_ms"][c]) / 5, 2)
                                                                        for c in p1["class_ms"] if p1["class_ms"][c] - p0["class_ms"][c] > 3}), flush=True)
        per = []
        for b in range(n_batches):
            p0 = ctx.profile()
            for _ in range(2):
                ctx.submit_device(batches[b].data_ptr(), nbytes)
            ctx.sync()
            p1 = ctx.profile()
            per.append({c[:7]: round((p1["class_ms"][c] - p0["class_ms"][c]) / 2, 2) for c in p1["class_ms"]
                        if p1["class_ms"][c] - p0["class_ms"][c] > 3})
        print("trial %d dummy +%d B: %s" % (t, sz, per), flush=True)
        # bench.py's pattern: N steps queued back to back, one sync at the end
        for depth in (5, 1, 5):
            p0 = ctx.profile()
            for i in range(5):
                ctx.submit_device(batches[i % n_batches].data_ptr(), nbytes)
                if depth == 1:
                    ctx.sync()
            ctx.sync()
            p1 = ctx.profile()
            print("   queue depth %d: %s" % (depth, {c[:7]: round((p1["class_ms"][c] - p0["class_ms"][c]) / 5, 2) for c in p1["class_ms"]
                                                    if p1["class_ms"][c] - p0["class_ms"][c] > 3}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
