"""Developer probe: one bench-like step with every library timer printed (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
profile = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # 1 = GKC_SYNTH_SKEWED
freq_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0         # 1 = minimizers in frequency order (table from the first 10^6 reads, as bench.py's freq_order block)
c = gkc.Counter(0)
rep = bench.repart_for_bench(10, parts)
db, do = c.synth_reads_device(2, n, 150, n * 5, 10000, profile=profile)
freq = None
if freq_mode:
    m = 10; n_s = min(n, 1_000_000)
    hb = c.device_to_host(db, n_s * 150); ho = np.arange(n_s + 1, dtype=np.uint64) * np.uint64(150)
    cnts = c.count_mmers(m, hb, ho)
    idx = np.nonzero(cnts)[0]; order_ = idx[np.lexsort((idx, cnts[idx]))]
    freq = np.full(4 ** m, 4 ** m, dtype=np.uint32); freq[order_] = np.arange(len(order_), dtype=np.uint32); freq[-1] = 4 ** m - 1
c.configure(k, 10, parts, rep, freq_order=freq)
names = ["scan_count", "scan_emit", "scan_refine", "dedupe_bin", "dedupe_sort", "expand_count", "expand_scatter", "bucket_sort", "bucket_sort_big", "bucket_sort_wg", "bucket_sort_deep", "split_levels", "compact",
         "total_stage_a", "total_stage_b"]
for it in range(3):
    base = {x: c.timing(x) for x in names}
    t0 = time.perf_counter()
    c.begin_pass(0); t1 = time.perf_counter()
    c.push_reads_device(db, do, n, n * 150); t2 = time.perf_counter()
    c.finish_pass(); t3 = time.perf_counter()
    print("iter", it, "begin %.1f ms push %.1f ms finish %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3),
          {x: (round(c.timing(x)[0] - base[x][0], 2), c.timing(x)[1] - base[x][1]) for x in names})
print(c.stats())
