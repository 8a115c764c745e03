import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
for k, parts in ((31, 4096), (63, 8192)):
    c = gkc.Counter(0); n = 20_000_000
    c.configure(k, 10, parts, bench.repart_for_bench(10, parts))
    db, do = c.synth_reads_device(2, n, 150, n * 5, 10000)
    c.begin_pass(0); c.push_reads_device(db, do, n, n * 150); c.finish_pass()
    tot = np.zeros(64)
    for p in range(0, parts, parts // 32):
        lo, hi, ab = c.partition(0, p)
        top = ((lo >> np.uint64(2 * k - 6)) if k <= 31 else (hi >> np.uint64(2 * k - 64 - 6))).astype(np.int64) & 63
        w = ab.astype(np.float64)          # keys weighted by abundance = k-mer occurrences routed (before dedupe); distinct would be unweighted
        tot += np.bincount(top, minlength=64)
    tot /= tot.sum() / 64
    print("k", k, "relative density of distinct k-mers by top 3 bases (64 groups):")
    print(np.round(tot, 2).tolist())
    c.close()
