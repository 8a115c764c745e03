"""Developer probe (GPU box): where do the overfull key-range sub-buckets of a partition come from? Counts 2e7 reads at the benchmark's
partition size, fetches one partition and looks inside its 13-bit bins: share of the keys under the most frequent 20-bit prefix of every
bin, and whether that prefix is one of the partition's own minimizers (as it appears at the start of a canonical k-mer)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
n, k, m = 20_000_000, 31, 10
parts = 820                                      # ~2.9e6 k-mers per partition, as in the 1e8-read benchmark
rep = bench.repart_for_bench(m, parts)
c = gkc.Counter(0)
c.configure(k, m, parts, rep)
db, do = c.synth_reads_device(2, n, 150, n * 5, 10000)
c.begin_pass(0); c.push_reads_device(db, do, n, n * 150); c.finish_pass()
def rc_m(x):
    r = 0
    for i in range(m):
        r = (r << 2) | (((x >> (2 * i)) & 3) ^ 2)
    return r
for p in (5, 400):
    lo, hi, ab = c.partition(0, p)
    keys = lo.astype(np.uint64); w = ab.astype(np.int64)
    tot = int(w.sum())
    b13 = (keys >> np.uint64(2 * k - 13)).astype(np.int64)
    p20 = (keys >> np.uint64(2 * k - 20)).astype(np.int64)
    bins = np.bincount(b13, weights=w, minlength=8192)
    mins = np.nonzero(rep == p)[0]
    hot = set(int(x) for x in mins) | set(rc_m(int(x)) for x in mins)
    rows = []
    for lo_e, hi_e in ((0, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 1 << 30)):
        sel = np.nonzero((bins > lo_e) & (bins <= hi_e))[0]
        kk = 0; top = 0; topm = 0
        for b in sel[:400]:
            msk = b13 == b
            pref = p20[msk]; ww = w[msk]
            u, inv = np.unique(pref, return_inverse=True)
            s = np.bincount(inv, weights=ww)
            j = int(np.argmax(s)); kk += int(ww.sum()); top += int(s[j]); topm += int(s[j]) if int(u[j]) in hot else 0
        rows.append((hi_e, len(sel), int(bins[sel].sum()) * 100.0 / tot, 100.0 * top / max(kk, 1), 100.0 * topm / max(kk, 1)))
    print("partition %d: %d keys, %d minimizers" % (p, tot, len(mins)))
    for r in rows:
        print("  bins <= %-10d: %5d bins, %5.1f %% of keys; top 20-bit prefix holds %5.1f %% of a bin's keys (%5.1f %% when it is one of the partition's minimizers)" % r)
