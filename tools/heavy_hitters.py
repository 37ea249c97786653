#!/usr/bin/env python3
"""How does the partitioned pipeline take reads with heavy hitters (repeat families, poly-A) — the k-mers that real
WGS data has and bench.py's uniform reads do not?  One C3 batch (50 M x 150 bp); a fraction of its reads is replaced by
copies of a few reads (every k-mer of such a read then occurs thousands of times in the batch and overflows its
finest bucket into the spill list) and by poly-A lines.  Prints per-kernel times per variant.

    python tools/heavy_hitters.py [reads]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd"))


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    import torch
    import bench
    import pyqm2
    dev = torch.device("cuda", 0)
    wl = dict(bench.WORKLOADS["c3"])
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
    off = pyqm2.tools.read_offsets(reads, 0, 150, 150, 22)
    nbytes = int(off[-1])
    base = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pyqm2.tools.reads_generate(genome.data_ptr(), n, base.data_ptr(), None, reads, 0, 150, 150, 22, sub_rate=wl["sub"], sample_from=covered)
    del genome
    torch.cuda.empty_cache()
    g = torch.Generator(device=dev)
    g.manual_seed(5)

    def variant(frac_rep, families, frac_poly, base_char="A", frac_random=0.0):
        buf = base.clone().view(reads, 151)
        if frac_random:                                   # reads of an unrelated genome: no hits, every walk ends on an empty slot
            m = int(reads * frac_random)
            where = torch.randint(0, reads, (m,), device=dev, generator=g)
            acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
            for a in range(0, m, 5_000_000):
                w = where[a:a + 5_000_000]
                seq = acgt[torch.randint(0, 4, (len(w), 151), device=dev, generator=g)]
                seq[:, 150] = 10
                buf[w] = seq
        if frac_rep:
            m = int(reads * frac_rep)
            where = torch.randint(0, reads, (m,), device=dev, generator=g)
            what = torch.randint(0, families, (m,), device=dev, generator=g)
            buf[where] = base.view(reads, 151)[what]
        if frac_poly:
            m = int(reads * frac_poly)
            where = torch.randint(0, reads, (m,), device=dev, generator=g)
            line = torch.full((151,), ord(base_char), dtype=torch.uint8, device=dev)
            line[150] = 10
            buf[where] = line
        return buf.view(-1)

    # (HH_KPK: the k-mers-per-KiB hint bench.py and the CLI give, 0.8 x 1024 for 150-base reads; 0 = none: buffers for one k-mer per byte)
    ctx = pyqm2.Counter(None, hash_size=H, k=k, device=0, flags=pyqm2.FLAG_TIMING, ext_keys=keys.data_ptr(), ext_counters=counters.data_ptr(),
                        kmers_per_kib=int(os.environ.get("HH_KPK", "824")))
    ctx.prepare(nbytes)
    only = os.environ.get("HH_ONLY")
    for name, args in (("uniform reads", (0, 0, 0)), ("5 % of the reads from 1000 families", (0.05, 1000, 0)),
                       ("5 % from 100 000 families", (0.05, 100_000, 0)), ("20 % from 10 000 families", (0.20, 10_000, 0)),
                       ("0.5 % poly-A", (0, 0, 0.005)), ("0.5 % poly-C (not in the dictionary)", (0, 0, 0.005, "C")), ("5 % from 1000 families + 0.5 % poly-A", (0.05, 1000, 0.005)),
                       ("half the reads random sequence", (0, 0, 0, "A", 0.5)), ("half random + 5 % from 1000 families", (0.05, 1000, 0, "A", 0.5))):
        if only and only not in name:
            continue
        buf = variant(*args)
        _, s0 = ctx.finish(want_depth=False)
        ctx.submit_device(buf.data_ptr(), nbytes)
        ctx.sync()
        _, s1 = ctx.finish(want_depth=False)
        per = {a: s1[a] - s0[a] for a in s1}
        p0 = ctx.profile()
        stamps = None
        try:                                                 # a -DQM2_STAMPS build (QM2_LIB=...): bucket_count's phases for this variant
            import ctypes as C
            stamps = pyqm2.lib().qm2_debug_stamps
            stamps((C.c_uint64 * 32)())                      # (reading clears)
        except AttributeError:
            pass
        for _ in range(3):
            ctx.submit_device(buf.data_ptr(), nbytes)
        ctx.sync()
        p1 = ctx.profile()
        if stamps:
            acc = (C.c_uint64 * 32)()
            stamps(acc)
            nl, npw = max(1, acc[8]), max(1, acc[9])
            lt, pt = sum(acc[i] for i in (0, 1, 2, 3, 10)), sum(acc[i] for i in (4, 5, 6, 7, 11))
            print("    bucket_count per wave (cycles / 3 launches): loaders %.0f = flush %.0f wait_regs %.0f stage %.0f fetch_issue %.0f barrier %.0f | probers %.0f = "
                  "kmer_arrival_wait %.0f kmer_load_issue %.0f phase_a %.0f phase_b %.0f barrier %.0f" % (
                      lt / nl / 3, acc[0] / nl / 3, acc[10] / nl / 3, acc[1] / nl / 3, acc[2] / nl / 3, acc[3] / nl / 3,
                      pt / npw / 3, acc[11] / npw / 3, acc[4] / npw / 3, acc[5] / npw / 3, acc[6] / npw / 3, acc[7] / npw / 3), flush=True)
        ms = {c: round((p1["class_ms"][c] - p0["class_ms"][c]) / 3, 2) for c in p1["class_ms"] if p1["class_ms"][c] - p0["class_ms"][c] > 0.3}
        tot = (p1["kernel_ms"] - p0["kernel_ms"]) / 3
        print("%-44s %6.1f ms per batch = %5.1f G bases/s  %s  fallbacks %d; slots inspected per k-mer %.3f" % (
              name, tot, reads * 150 / tot / 1e6, ms, p1["fallback_batches"] - p0["fallback_batches"], per["n_probes"] / max(1, per["n_kmers"])), flush=True)
        del buf
    ctx.close()


if __name__ == "__main__":
    main()
