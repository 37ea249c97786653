import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "quick-mer2_amd"))
import pyqm2 as qm2
n, H, k = 6_000_000, 1 << 23, 30
genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
qm2.tools.genome_generate(genome.data_ptr(), n, seed=5, n_blocks=[], dups=[])
keys = torch.zeros(H, dtype=torch.int64, device="cuda")
scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
nd, nu = qm2.tools.dict_build(genome.data_ptr(), n, k, keys.data_ptr(), H, scratch.data_ptr())
n_reads, rl = 400_000, 150
reads = torch.zeros(n_reads * (rl + 1), dtype=torch.uint8, device="cuda")
qm2.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, rl, rl, 5)
torch.cuda.synchronize()
out = {}
for name, flags in (("direct", qm2.FLAG_DIRECT), ("partition", qm2.FLAG_PARTITION)):
    with qm2.Counter(None, hash_size=H, k=k, ext_keys=keys.data_ptr(), flags=flags) as c:
        for _ in range(3):
            c.submit_device(reads.data_ptr(), reads.numel())
        d, st = c.finish()
        prof = c.profile()
    out[name] = (d, st, prof)
bad = np.flatnonzero(out["partition"][0] != out["direct"][0])
print("OLD KERNEL, SKEWED: slots that differ from the direct kernel:", len(bad), "in buckets", sorted(set((bad >> 12).tolist()))[:12], "partitioned batches", out["partition"][2]["partitioned_batches"], "dict", nu)
