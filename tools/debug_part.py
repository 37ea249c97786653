import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quick-mer2_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, pyqm2 as q
def run(lgH, n, n_reads, bits):
    H = 1 << lgH
    genome = torch.zeros(n, dtype=torch.uint8, device="cuda")
    q.tools.genome_generate(genome.data_ptr(), n, seed=3)
    keys = torch.zeros(H, dtype=torch.int64, device="cuda"); scratch = torch.zeros(H, dtype=torch.int64, device="cuda")
    nd, nu = q.tools.dict_build(genome.data_ptr(), n, 30, keys.data_ptr(), H, scratch.data_ptr()); del scratch
    reads = torch.zeros(n_reads * 151, dtype=torch.uint8, device="cuda")
    q.tools.reads_generate(genome.data_ptr(), n, reads.data_ptr(), None, n_reads, 0, 150, 150, 5)
    torch.cuda.synchronize()
    with q.Counter(None, hash_size=H, k=30, ext_keys=keys.data_ptr(), flags=q.FLAG_DIRECT) as c:
        c.submit_device(reads.data_ptr(), reads.numel()); dd, sd = c.finish()
    with q.Counter(None, hash_size=H, k=30, ext_keys=keys.data_ptr(), flags=q.FLAG_PARTITION, part_max_bits=bits) as c:
        c.submit_device(reads.data_ptr(), reads.numel()); dp, sp = c.finish()
    diff = np.flatnonzero(dd != dp)
    print(json.dumps({"lgH": lgH, "bits": bits, "fill": nu / H, "stats_eq": sd == sp, "n_diff": int(len(diff)),
                      "sum_direct": int(dd.astype(np.int64).sum()), "sum_part": int(dp.astype(np.int64).sum()), "hits": sd["n_hits"]}))
    if len(diff):
        pos = diff % 4096
        print(" first diffs", diff[:10].tolist(), "pos-in-bucket", pos[:20].tolist())
        print(" direct", dd[diff[:10]].tolist(), "part", dp[diff[:10]].tolist())
        print(" hist pos<256:", int((pos < 256).sum()), "pos>=3840:", int((pos >= 3840).sum()), "interior:", int(((pos >= 256) & (pos < 3840)).sum()))
for lgH, n, nr, bits in ((17, 60000, 20000, 8), (17, 60000, 20000, 3), (17, 60000, 20000, 2), (22, 2_000_000, 300_000, 8), (22, 2_000_000, 300_000, 5), (24, 6_000_000, 1_000_000, 8), (24, 6_000_000, 1_000_000, 6), (28, 50_000_000, 2_000_000, 8)):
    run(lgH, n, nr, bits)
